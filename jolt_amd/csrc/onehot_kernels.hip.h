// jolt_amd/csrc/onehot_kernels.hip.h -- gathers over hot-index columns (LazyFoldedRa, crates/jolt-kernels/src/optimized/lazy_ra.rs).
// All of them are lookups + additions: the eq weights of the bound bits are pre-scaled into the branch tables, exactly as the
// reference does (lazy_ra.rs:17-24), so the per-cycle work has no multiplication and reads 1 byte instead of 32 per entry.
#pragma once
#include "onehot.hpp"
#include "sumcheck_kernels.hip.h"

namespace jolt {

// value(p, j) at branch width `width` (lazy_ra.rs:184-209 `gather`):
//   sum_{off < width} branch[off * K + index(p, j * width + off)]        (cold cycles contribute nothing)
// idx = the column's first entry; wide = 1 for 16-bit indices
__device__ __forceinline__ Fr onehot_gather(const Fr* __restrict__ branch, const uint8_t* __restrict__ idx, uint32_t width, uint32_t K, size_t j, uint32_t wide = 0) {
    Fr sum = Fr::zero();
    for (uint32_t off = 0; off < width; ++off) {
        const uint32_t k = hot_load(idx, j * width + off, wide);
        if (k != kColdIdx) sum = add(sum, ld_fr(branch + (size_t)off * K + k));
    }
    return sum;
}

// double_branches (lazy_ra.rs:211-231): next = [(1 - c) * table ; c * table], for all polynomials at once.
// in/out: [poly][width * K] resp. [poly][2 * width * K]
static __global__ __launch_bounds__(kBlock) void k_onehot_double_branches(const Fr* __restrict__ in, Fr* __restrict__ out, size_t per_poly_in, size_t n_polys, Fr c,
                                                                         int shifted) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= per_poly_in * n_polys) return;
    size_t p = i / per_poly_in, e = i - p * per_poly_in;
    Fr v = ld_fr(in + i);
    Fr hi;
    if (shifted) {
        uint32_t chi[4] = {c.l[4], c.l[5], c.l[6], c.l[7]};
        hi = mul_shifted(v, chi);
    } else {
        hi = mul(v, c);
    }
    // (1 - c) * v = v - c * v: the same field element as the reference's one_minus * value (exact arithmetic)
    st_fr(out + p * 2 * per_poly_in + e, sub(v, hi));
    st_fr(out + p * 2 * per_poly_in + per_poly_in + e, hi);
}

// materialize (lazy_ra.rs:233-268): dense[p][j] = gather(branch_p, width, j) for j < cycles / width; blockIdx.y = polynomial
struct OneHotDense {
    Fr* out[kMaxBatchTables];
};
static __global__ __launch_bounds__(kBlock) void k_onehot_materialize(const Fr* __restrict__ branch, size_t per_poly, const uint8_t* __restrict__ idx, size_t cycles,
                                                                     uint32_t width, uint32_t K, size_t first_poly, OneHotDense o, uint32_t wide = 0) {
    const size_t p = blockIdx.y;
    size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= cycles / width) return;
    st_fr(o.out[p] + j, onehot_gather(branch + (first_poly + p) * per_poly, hot_col(idx, (first_poly + p) * cycles, wide), width, K, j, wide));
}

// Pushforward tables (optimized/booleanity.rs:24-31): G_p[k] = sum_j w[j] * [index(p, j) == k].  One block accumulates a
// slice of the cycles into K buckets in LDS (one owner lane per bucket and pass: no atomics on 256-bit values), partial
// tables are summed by k_onehot_pushforward_reduce.  blockIdx.y = polynomial.
static __global__ __launch_bounds__(kBlock) void k_onehot_pushforward(const uint8_t* __restrict__ idx, uint32_t wide, const Fr* __restrict__ w, size_t cycles, uint32_t K,
                                                                     Fr* __restrict__ partials /* [poly][block][K] */) {
    extern __shared__ unsigned char smem_raw[];
    Fr* buckets = reinterpret_cast<Fr*>(smem_raw);  // K entries
    const size_t p = blockIdx.y;
    for (uint32_t k = threadIdx.x; k < K; k += kBlock) buckets[k] = Fr::zero();
    __syncthreads();
    const uint8_t* col = hot_col(idx, p * cycles, wide);
    const size_t per_block = (cycles + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per_block, hi = lo + per_block < cycles ? lo + per_block : cycles;
    // lane k owns bucket k: every lane scans the block's slice in chunks of kBlock cycles staged through LDS indices
    __shared__ uint32_t s_idx[kBlock];
    for (size_t base = lo; base < hi; base += kBlock) {
        size_t j = base + threadIdx.x;
        s_idx[threadIdx.x] = j < hi ? hot_load(col, j, wide) : kColdIdx;
        __syncthreads();
        const size_t n = hi - base < (size_t)kBlock ? hi - base : (size_t)kBlock;
        for (uint32_t k = threadIdx.x; k < K; k += kBlock) {
            Fr acc = buckets[k];
            for (size_t t = 0; t < n; ++t)
                if (s_idx[t] == (uint32_t)k) acc = add(acc, ld_fr(w + base + t));
            buckets[k] = acc;
        }
        __syncthreads();
    }
    for (uint32_t k = threadIdx.x; k < K; k += kBlock) st_fr(partials + ((size_t)p * gridDim.x + blockIdx.x) * K + k, buckets[k]);
}
// The same partial tables for small K (<= 32: every RA column of the reference, K = 16), one wavefront per (polynomial, slice of
// cycles): each LANE keeps its own K buckets in LDS ([bucket][half][lane] -> conflict-free 16-byte accesses) and walks the slice
// with coalesced index / weight loads, so all 64 lanes add weights all the time -- the kernel above has K lanes working per
// workgroup and re-scans the slice once per bucket (36 columns at T = 2^20: 6.0 ms there, see DESIGN.md 3.4b for this one).
// blockIdx.x = polynomial (fastest: the columns of one slice run together and share its weights in L2), blockIdx.y = slice.
static __global__ __launch_bounds__(64) void k_onehot_pushforward_lanes(const uint8_t* __restrict__ idx, const Fr* __restrict__ w, size_t cycles, uint32_t K,
                                                                        size_t per_block, Fr* __restrict__ partials /* [poly][slice][K] */) {
    extern __shared__ uint4 push_sh[];  // K * 2 * 64 uint4
    const uint32_t lane = threadIdx.x;
    const size_t p = blockIdx.x, slice = blockIdx.y, nslices = gridDim.y;
    for (uint32_t e = lane; e < K * 2 * 64; e += 64) push_sh[e] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const uint8_t* col = idx + p * cycles;
    const size_t lo = slice * per_block, hi = lo + per_block < cycles ? lo + per_block : cycles;
    // four cycles per lane in flight: the index byte and the 32-byte weight of a cycle are global loads, the bucket update a dependent LDS read-modify-write;
    // at 5 wavefronts per CU (the buckets' 32 KiB each) nothing else hides the loads
    constexpr int kFlight = 4;
    for (size_t j0 = lo + lane; j0 < hi; j0 += 64 * kFlight) {
        uint8_t kk[kFlight];
        Fr xx[kFlight];
#pragma unroll
        for (int q = 0; q < kFlight; ++q) {
            const size_t j = j0 + 64 * (size_t)q;
            kk[q] = j < hi ? col[j] : kOneHotCold;
        }
#pragma unroll
        for (int q = 0; q < kFlight; ++q)
            if (kk[q] != kOneHotCold) xx[q] = ld_fr(w + j0 + 64 * (size_t)q);
#pragma unroll
        for (int q = 0; q < kFlight; ++q) {
            if (kk[q] == kOneHotCold) continue;
            uint4* b0 = push_sh + ((uint32_t)kk[q] * 2) * 64 + lane;
            uint4* b1 = b0 + 64;
            uint4 a0 = *b0, a1 = *b1;
            Fr acc;
            acc.l[0] = a0.x; acc.l[1] = a0.y; acc.l[2] = a0.z; acc.l[3] = a0.w; acc.l[4] = a1.x; acc.l[5] = a1.y; acc.l[6] = a1.z; acc.l[7] = a1.w;
            acc = add(acc, xx[q]);
            *b0 = make_uint4(acc.l[0], acc.l[1], acc.l[2], acc.l[3]);
            *b1 = make_uint4(acc.l[4], acc.l[5], acc.l[6], acc.l[7]);
        }
    }
    __syncthreads();
    auto load_lds = [&](uint32_t k, uint32_t l) {
        const uint4 a0 = push_sh[(k * 2) * 64 + l], a1 = push_sh[(k * 2 + 1) * 64 + l];
        Fr s;
        s.l[0] = a0.x; s.l[1] = a0.y; s.l[2] = a0.z; s.l[3] = a0.w; s.l[4] = a1.x; s.l[5] = a1.y; s.l[6] = a1.z; s.l[7] = a1.w;
        return s;
    };
    if (64 % K == 0) {
        // g = 64 / K lanes per bucket: lane (b, q) adds K of the 64 per-lane values of bucket b, the g parts meet in log2 g shuffles
        const uint32_t g = 64 / K, b = lane / g, q = lane % g;
        Fr s = Fr::zero();
        for (uint32_t i = 0; i < K; ++i) s = add(s, load_lds(b, q * K + i));
        for (uint32_t off = g >> 1; off >= 1; off >>= 1) {
            Fr o;
#pragma unroll
            for (int t = 0; t < 8; ++t) o.l[t] = __shfl_xor(s.l[t], off, 64);
            s = add(s, o);
        }
        if (q == 0) st_fr(partials + (p * nslices + slice) * K + b, s);
    } else {
        for (uint32_t k = 0; k < K; ++k) {
            Fr s = load_lds(k, lane);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                Fr o;
#pragma unroll
                for (int t = 0; t < 8; ++t) o.l[t] = __shfl_xor(s.l[t], off, 64);
                s = add(s, o);
            }
            if (lane == 0) st_fr(partials + (p * nslices + slice) * K + k, s);
        }
    }
}
static __global__ __launch_bounds__(kBlock) void k_onehot_pushforward_reduce(const Fr* __restrict__ partials, int nblocks, uint32_t K, Fr* __restrict__ out) {
    const size_t p = blockIdx.y;
    uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= K) return;
    Fr s = Fr::zero();
    for (int b = 0; b < nblocks; ++b) s = add(s, ld_fr(partials + ((size_t)p * nblocks + b) * K + k));
    st_fr(out + p * K + k, s);
}

// The same sums for K <= kBlock with the partial tables of a polynomial dealt to kBlock / K lanes per bucket (round 6, second session): the kernel above walks its
// nblocks partials as ONE dependent chain per (polynomial, bucket) -- 1024 loads and additions at T = 2^22, 0.37 ms for 576 busy threads; here a workgroup per polynomial
// keeps kBlock / K chains of nblocks K / kBlock links per bucket and folds them through LDS.  blockIdx.x = polynomial.
static __global__ __launch_bounds__(kBlock) void k_onehot_pushforward_reduce_split(const Fr* __restrict__ partials, int nblocks, uint32_t K, Fr* __restrict__ out) {
    __shared__ Fr part[kBlock];
    const size_t p = blockIdx.x;
    const uint32_t G = kBlock / K, k = threadIdx.x % K, g = threadIdx.x / K;  // G >= 1 chains per bucket; threads beyond G * K idle
    Fr s = Fr::zero();
    if (g < G)
        for (int b = (int)g; b < nblocks; b += (int)G) s = add(s, ld_fr(partials + ((size_t)p * nblocks + b) * K + k));
    part[threadIdx.x] = s;
    __syncthreads();
    if (g == 0) {
        for (uint32_t h = 1; h < G; ++h) s = add(s, part[h * K + k]);
        st_fr(out + p * K + k, s);
    }
}

// Round sums of eq(w, j) * sum_v c_v * prod_{i<F} ra_{vF+i}(j) while the selector columns are still index-encoded: the same
// sums as k_split_eq_uniform<F> over dense tables, with every (lo, hi) pair gathered (lazy_ra.rs:116-149 lo_hi_all).
struct LazyArgs {
    const uint8_t* idx;   // [poly][cycles0]
    uint32_t wide;        // 16-bit indices (K > 255)
    const Fr* branch;     // [poly][width * K]
    size_t cycles0;       // unbound cycle count (row stride of idx)
    uint32_t width, K;
    Fr coeff[kMaxGroups];
    uint32_t coeff_one[kMaxGroups];
    int V;
};
template <int F>
static __global__ __launch_bounds__(kBlock) void k_split_eq_uniform_lazy(LazyArgs a, const Fr* __restrict__ e_out, const Fr* __restrict__ e_in, int in_bits,
                                                                         size_t rows, Fr* __restrict__ partials, uint32_t ticket, uint32_t slot, RoundDone rd) {
    Fr acc[F];
#pragma unroll
    for (int t = 0; t < F; ++t) acc[t] = Fr::zero();
    const size_t items = rows * (size_t)a.V;
    const size_t mask = ((size_t)1 << in_bits) - 1;
    const size_t per_poly = (size_t)a.width * a.K;
    size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < items; i += stride) {
        const uint32_t v = (uint32_t)(i / rows);
        const size_t row = i - (size_t)v * rows;
        Fr lo[F], hi[F];
#pragma unroll
        for (int k = 0; k < F; ++k) {
            const size_t p = (size_t)v * F + k;
            lo[k] = onehot_gather(a.branch + p * per_poly, hot_col(a.idx, p * a.cycles0, a.wide), a.width, a.K, 2 * row, a.wide);
            hi[k] = onehot_gather(a.branch + p * per_poly, hot_col(a.idx, p * a.cycles0, a.wide), a.width, a.K, 2 * row + 1, a.wide);
        }
        Fr w = mul(ld_fr(e_out + (row >> in_bits)), ld_fr(e_in + (row & mask)));
        if (!a.coeff_one[v]) w = mul(w, a.coeff[v]);
        lo[0] = mul(lo[0], w);
        hi[0] = mul(hi[0], w);
        Fr q[F];
        uniform_item<F>(lo, hi, q);
#pragma unroll
        for (int t = 0; t < F; ++t) acc[t] = add(acc[t], q[t]);
    }
    block_reduce_store<F>(acc, partials);
    finish_member(partials, F, ticket, slot, rd);
}

// The same sums with the branch tables of ONE product group staged in LDS (F * width * (K + 1) entries; entry K of every
// block is zero, so a cold cycle is a lookup like any other and the gather has no branch).  The branch tables outgrow the
// 32 KiB L1 after the first bind and every 32-byte lookup then drags a 128-byte line out of L2: at T = 2^20 the 33.5 M
// lookups of a round of the 32-column member cost ~270 us regardless of how few pairs were left.
// blockIdx.x = v * blocks_per_v + block within the group; the 2 * width index bytes of a pair are one aligned load.
template <int F>
static __global__ __launch_bounds__(kBlock) void k_split_eq_uniform_lazy_lds(LazyArgs a, uint32_t blocks_per_v, const Fr* __restrict__ e_out,
                                                                             const Fr* __restrict__ e_in, int in_bits, size_t rows, Fr* __restrict__ partials,
                                                                             uint32_t ticket, uint32_t slot, RoundDone rd) {
    extern __shared__ uint4 lazy_lds_raw[];
    Fr* tab = reinterpret_cast<Fr*>(lazy_lds_raw);
    const uint32_t v = blockIdx.x / blocks_per_v, bx = blockIdx.x - v * blocks_per_v;
    const uint32_t KP = a.K + 1, per_poly_lds = a.width * KP, per_poly = a.width * a.K;
    for (uint32_t e = threadIdx.x; e < F * per_poly_lds; e += kBlock) {
        uint32_t k = e / per_poly_lds, r = e - k * per_poly_lds, off = r / KP, entry = r - off * KP;
        tab[e] = entry < a.K ? ld_fr(a.branch + ((size_t)v * F + k) * per_poly + (size_t)off * a.K + entry) : Fr::zero();
    }
    __syncthreads();
    Fr acc[F];
#pragma unroll
    for (int t = 0; t < F; ++t) acc[t] = Fr::zero();
    const size_t mask = ((size_t)1 << in_bits) - 1;
    const uint32_t width = a.width;
    for (size_t row = (size_t)bx * kBlock + threadIdx.x; row < rows; row += (size_t)blocks_per_v * kBlock) {
        Fr lo[F], hi[F];
#pragma unroll
        for (int k = 0; k < F; ++k) {
            const uint8_t* col = a.idx + ((size_t)v * F + k) * a.cycles0 + 2 * row * width;  // lo: bytes [0, width), hi: [width, 2 width)
            uint64_t b_lo, b_hi;  // the index bytes, little-endian
            if (width == 8) { uint4 q = *reinterpret_cast<const uint4*>(col); b_lo = (uint64_t)q.x | ((uint64_t)q.y << 32); b_hi = (uint64_t)q.z | ((uint64_t)q.w << 32); }
            else if (width == 4) { uint2 q = *reinterpret_cast<const uint2*>(col); b_lo = q.x; b_hi = q.y; }
            else if (width == 2) { uint32_t q = *reinterpret_cast<const uint32_t*>(col); b_lo = q & 0xFFFFu; b_hi = q >> 16; }
            else { uint32_t q = *reinterpret_cast<const uint16_t*>(col); b_lo = q & 0xFFu; b_hi = q >> 8; }
            const Fr* tk = tab + (uint32_t)k * per_poly_lds;
            Fr s0 = Fr::zero(), s1 = Fr::zero();
            for (uint32_t off = 0; off < width; ++off) {
                uint32_t i0 = (uint32_t)(b_lo >> (8 * off)) & 0xFFu, i1 = (uint32_t)(b_hi >> (8 * off)) & 0xFFu;
                i0 = i0 == kOneHotCold ? a.K : i0;
                i1 = i1 == kOneHotCold ? a.K : i1;
                s0 = add(s0, tk[off * KP + i0]);
                s1 = add(s1, tk[off * KP + i1]);
            }
            lo[k] = s0;
            hi[k] = s1;
        }
        Fr w = mul(ld_fr(e_out + (row >> in_bits)), ld_fr(e_in + (row & mask)));
        if (!a.coeff_one[v]) w = mul(w, a.coeff[v]);
        lo[0] = mul(lo[0], w);
        hi[0] = mul(hi[0], w);
        Fr q[F];
        uniform_item<F>(lo, hi, q);
#pragma unroll
        for (int t = 0; t < F; ++t) acc[t] = add(acc[t], q[t]);
    }
    block_reduce_store<F>(acc, partials);
    finish_member(partials, F, ticket, slot, rd);
}

// Booleanity cycle phase (crates/jolt-kernels/src/optimized/booleanity.rs:574-633): inner quadratic
//   q(X) = sum_rows E(row) * sum_i (H_i(X)^2 - rho_i H_i(X)),   q(0) from H at 0, q(inf) from the pair delta,
// with H_i the gamma-pre-scaled address-folded selector columns (rho_i = gamma^i), index-encoded (LAZY) or dense.
struct BooleanityArgs {
    const Fr* tabs[kMaxBatchTables];  // dense state: column i
    Fr rho[kMaxBatchTables];
    int n;
    // lazy state
    const uint8_t* idx;
    uint32_t wide;
    const Fr* branch;
    size_t cycles0;
    uint32_t width, K;
};
template <bool LAZY>
static __global__ __launch_bounds__(kBlock) void k_split_eq_booleanity(BooleanityArgs a, const Fr* __restrict__ e_out, const Fr* __restrict__ e_in, int in_bits,
                                                                       size_t rows, Fr* __restrict__ partials, uint32_t ticket, uint32_t slot, RoundDone rd) {
    Fr acc[2] = {Fr::zero(), Fr::zero()};
    const size_t mask = ((size_t)1 << in_bits) - 1;
    const size_t per_poly = (size_t)a.width * a.K;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t row = (size_t)blockIdx.x * kBlock + threadIdx.x; row < rows; row += stride) {
        Fr constant = Fr::zero(), leading = Fr::zero();  // plain field sums: the reference's deferred-reduction lanes give the same values
        for (int i = 0; i < a.n; ++i) {
            Fr h0, h1;
            if constexpr (LAZY) {
                h0 = onehot_gather(a.branch + (size_t)i * per_poly, hot_col(a.idx, (size_t)i * a.cycles0, a.wide), a.width, a.K, 2 * row, a.wide);
                h1 = onehot_gather(a.branch + (size_t)i * per_poly, hot_col(a.idx, (size_t)i * a.cycles0, a.wide), a.width, a.K, 2 * row + 1, a.wide);
            } else {
                h0 = ld_fr(a.tabs[i] + 2 * row);
                h1 = ld_fr(a.tabs[i] + 2 * row + 1);
            }
            Fr delta = sub(h1, h0);
            constant = add(constant, mul(h0, sub(h0, a.rho[i])));
            leading = add(leading, sqr(delta));
        }
        const Fr w = mul(ld_fr(e_out + (row >> in_bits)), ld_fr(e_in + (row & mask)));
        acc[0] = add(acc[0], mul(w, constant));
        acc[1] = add(acc[1], mul(w, leading));
    }
    block_reduce_store<2>(acc, partials);
    finish_member(partials, 2, ticket, slot, rd);
}

// The index-encoded rounds of the same member with the branch tables in LDS (round 6, second session): n columns x width x K entries spill the 32 KiB L1 from the
// second lazy round on (36 columns of K = 16: 18 / 36 / 72 / 144 KiB), and every 32-byte lookup then costs an L2 line.  The columns are dealt to `col_groups` groups of
// workgroups, `cols_per_group` columns each (their tables, one zero entry per row for the cold index: <= 48 KiB); the summand is a SUM over columns, so the groups' partial
// sums simply add.  blockIdx.x = group * blocks_per_group + block within the group; the 2 * width index bytes of a pair are one aligned load.
static __global__ __launch_bounds__(kBlock) void k_split_eq_booleanity_lds(BooleanityArgs a, uint32_t cols_per_group, uint32_t blocks_per_group, const Fr* __restrict__ e_out,
                                                                           const Fr* __restrict__ e_in, int in_bits, size_t rows, Fr* __restrict__ partials, uint32_t ticket,
                                                                           uint32_t slot, RoundDone rd) {
    extern __shared__ uint4 bool_lds_raw[];
    Fr* tab = reinterpret_cast<Fr*>(bool_lds_raw);
    const uint32_t g = blockIdx.x / blocks_per_group, bx = blockIdx.x - g * blocks_per_group;
    const uint32_t c0 = g * cols_per_group, nc = min(cols_per_group, (uint32_t)a.n - c0);
    const uint32_t KP = a.K + 1, per_col_lds = a.width * KP, per_col = a.width * a.K;
    for (uint32_t e = threadIdx.x; e < nc * per_col_lds; e += kBlock) {
        const uint32_t k = e / per_col_lds, r = e - k * per_col_lds, off = r / KP, entry = r - off * KP;
        tab[e] = entry < a.K ? ld_fr(a.branch + (size_t)(c0 + k) * per_col + (size_t)off * a.K + entry) : Fr::zero();
    }
    __syncthreads();
    Fr acc[2] = {Fr::zero(), Fr::zero()};
    const size_t mask = ((size_t)1 << in_bits) - 1;
    const uint32_t width = a.width;
    for (size_t row = (size_t)bx * kBlock + threadIdx.x; row < rows; row += (size_t)blocks_per_group * kBlock) {
        Fr constant = Fr::zero(), leading = Fr::zero();
        for (uint32_t k = 0; k < nc; ++k) {
            const uint8_t* col = a.idx + (size_t)(c0 + k) * a.cycles0 + 2 * row * width;  // lo: bytes [0, width), hi: [width, 2 width)
            uint64_t b_lo, b_hi;
            if (width == 8) { uint4 q = *reinterpret_cast<const uint4*>(col); b_lo = (uint64_t)q.x | ((uint64_t)q.y << 32); b_hi = (uint64_t)q.z | ((uint64_t)q.w << 32); }
            else if (width == 4) { uint2 q = *reinterpret_cast<const uint2*>(col); b_lo = q.x; b_hi = q.y; }
            else if (width == 2) { uint32_t q = *reinterpret_cast<const uint32_t*>(col); b_lo = q & 0xFFFFu; b_hi = q >> 16; }
            else { uint32_t q = *reinterpret_cast<const uint16_t*>(col); b_lo = q & 0xFFu; b_hi = q >> 8; }
            const Fr* tk = tab + k * per_col_lds;
            Fr h0 = Fr::zero(), h1 = Fr::zero();
            for (uint32_t off = 0; off < width; ++off) {
                uint32_t i0 = (uint32_t)(b_lo >> (8 * off)) & 0xFFu, i1 = (uint32_t)(b_hi >> (8 * off)) & 0xFFu;
                i0 = i0 == kOneHotCold ? a.K : i0;
                i1 = i1 == kOneHotCold ? a.K : i1;
                h0 = add(h0, tk[off * KP + i0]);
                h1 = add(h1, tk[off * KP + i1]);
            }
            const Fr delta = sub(h1, h0);
            constant = add(constant, mul(h0, sub(h0, a.rho[c0 + k])));
            leading = add(leading, sqr(delta));
        }
        const Fr w = mul(ld_fr(e_out + (row >> in_bits)), ld_fr(e_in + (row & mask)));
        acc[0] = add(acc[0], mul(w, constant));
        acc[1] = add(acc[1], mul(w, leading));
    }
    block_reduce_store<2>(acc, partials);
    finish_member(partials, 2, ticket, slot, rd);
}

// row-major form (uniform_rows_body): one item per pair, the V products inside; used while a round has enough pairs to fill the chip
template <int F>
static __global__ __launch_bounds__(kBlock) void k_split_eq_uniform_lazy_rows(LazyArgs a, const Fr* __restrict__ e_out, const Fr* __restrict__ e_in, int in_bits,
                                                                              size_t rows, Fr* __restrict__ partials, uint32_t ticket, uint32_t slot, RoundDone rd) {
    Fr acc[F];
#pragma unroll
    for (int t = 0; t < F; ++t) acc[t] = Fr::zero();
    const size_t per_poly = (size_t)a.width * a.K;
    auto load = [&](int v, int k, size_t row, Fr& lo, Fr& hi) {
        const size_t p = (size_t)v * F + k;
        lo = onehot_gather(a.branch + p * per_poly, hot_col(a.idx, p * a.cycles0, a.wide), a.width, a.K, 2 * row, a.wide);
        hi = onehot_gather(a.branch + p * per_poly, hot_col(a.idx, p * a.cycles0, a.wide), a.width, a.K, 2 * row + 1, a.wide);
    };
    uniform_rows_body<F>(load, a.V, a.coeff, a.coeff_one, e_out, e_in, in_bits, rows, acc);
    block_reduce_store<F>(acc, partials);
    finish_member(partials, F, ticket, slot, rd);
}

// First round of a lazily bound F = 4 member (branch width 1): a factor pair's values are table entries, so the quadratic half
// f0*f1 on {0,1,2} is four gathers from the 17x17 pair table P[a][b] = T0[a]*T1[b] (index 16: cold cycle = 0):
//   A(0) = P[i0][i1],  A(1) = P[j0][j1],  A(2) = (2 h0 - l0)(2 h1 - l1) = 4 P[j0][j1] - 2 P[j0][i1] - 2 P[i0][j1] + P[i0][i1]
// -- no multiplication until the four products A(t)*B(t): 37 multiplies per pair instead of 85 for V = 8.
struct LazyPairArgs {
    const uint8_t* idx;  // [poly][cycles0]
    const Fr* pair;      // [v][2][289]
    size_t cycles0;
    int V;
};
__device__ __forceinline__ void pair_quadratic(const Fr* __restrict__ P, uint32_t i0, uint32_t i1, uint32_t j0, uint32_t j1, Fr& a0, Fr& a1, Fr& a2) {
    a0 = ld_fr(P + i0 * 17 + i1);
    a1 = ld_fr(P + j0 * 17 + j1);
    const Fr x = add(ld_fr(P + j0 * 17 + i1), ld_fr(P + i0 * 17 + j1));  // h0 l1 + l0 h1
    a2 = add(sub(dbl(dbl(a1)), dbl(x)), a0);
}
static __global__ __launch_bounds__(kBlock) void k_split_eq_uniform_lazy_first(LazyPairArgs a, const Fr* __restrict__ e_out, const Fr* __restrict__ e_in, int in_bits,
                                                                               size_t rows, Fr* __restrict__ partials, uint32_t ticket, uint32_t slot, RoundDone rd) {
    Fr acc[4] = {Fr::zero(), Fr::zero(), Fr::zero(), Fr::zero()};
    const size_t mask = ((size_t)1 << in_bits) - 1;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t row = (size_t)blockIdx.x * kBlock + threadIdx.x; row < rows; row += stride) {
        Fr s[4] = {Fr::zero(), Fr::zero(), Fr::zero(), Fr::zero()};
        for (int v = 0; v < a.V; ++v) {
            uint32_t i[4], j[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint8_t* col = a.idx + (size_t)(4 * v + k) * a.cycles0 + 2 * row;
                const uint32_t x = col[0], y = col[1];
                i[k] = x == kOneHotCold ? 16u : x;
                j[k] = y == kOneHotCold ? 16u : y;
            }
            Fr A0, A1, A2, B0, B1, B2, q[4];
            pair_quadratic(a.pair + (size_t)(2 * v) * 289, i[0], i[1], j[0], j[1], A0, A1, A2);
            pair_quadratic(a.pair + (size_t)(2 * v + 1) * 289, i[2], i[3], j[2], j[3], B0, B1, B2);
            uniform_quadratic_halves(A0, A1, A2, B0, B1, B2, q);
#pragma unroll
            for (int t = 0; t < 4; ++t) s[t] = add(s[t], q[t]);
        }
        const Fr w = mul(ld_fr(e_out + (row >> in_bits)), ld_fr(e_in + (row & mask)));
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = add(acc[t], mul(s[t], w));
    }
    block_reduce_store<4>(acc, partials);
    finish_member(partials, 4, ticket, slot, rd);
}

}  // namespace jolt
