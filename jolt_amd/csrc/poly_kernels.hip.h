// jolt_amd/csrc/poly_kernels.hip.h -- dense-table kernels: bind, eq/LT/eq+1 expansion, small-scalar promotion, sums.
//
// Layout in HBM: a table is a contiguous array of 32-byte Fr (the reference's Vec<Fr>), index = big-endian boolean
// point.  All of these kernels are HBM-bandwidth bound (1 Fr multiply per 96 B for bind) -- no MFMA, no GEMM shape.
#pragma once
#include "desc.hpp"

namespace jolt {

__device__ __forceinline__ Fr ld_fr(const Fr* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
__device__ __forceinline__ void st_fr(Fr* p, const Fr& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// Device-coherent store / load of one field element (relaxed agent-scope atomics on its four 64-bit halves): written through
// to where every XCD sees it, read past the reader's own caches.  The per-workgroup partial sums travel this way, so that a
// workgroup does not need a device-scope release fence (an L2 write-back of everything dirty, bound tables included: measured
// 0.35 ms of a 7.1 ms pass over 256 workgroups x ~300 kernels) before it takes its completion ticket.
__device__ __forceinline__ void st_fr_agent(Fr* p, const Fr& v) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        __hip_atomic_store(q + k, (unsigned long long)v.l[2 * k] | ((unsigned long long)v.l[2 * k + 1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_fr_system(Fr* p, const Fr& v) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        __hip_atomic_store(q + k, (unsigned long long)v.l[2 * k] | ((unsigned long long)v.l[2 * k + 1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Wait until every store this lane has issued is acknowledged.  The written-through stores above (sc1 / sc0 sc1) are
// acknowledged once they have reached the coherence point of their scope; a workgroup barrier alone only orders LDS traffic
// (s_waitcnt lgkmcnt), so the lanes that stored partial sums, round sums or counter resets drain vmcnt themselves before the
// barrier / ticket that publishes them.  Far cheaper than a release fence (no L2 write-back of bound tables).
__device__ __forceinline__ void wait_stores_acked() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void st_u32_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ Fr ld_fr_agent(const Fr* p) {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    Fr v;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned long long w = __hip_atomic_load(q + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.l[2 * k] = (uint32_t)w;
        v.l[2 * k + 1] = (uint32_t)(w >> 32);
    }
    return v;
}

// lo + r*(hi - lo); SHIFTED = r has its four low limbs zero (the 125-bit challenge shape) -> half the multiplies
template <bool SHIFTED>
__device__ __forceinline__ Fr bind_pair(const Fr& lo, const Fr& hi, const Fr& r) {
    Fr d = sub(hi, lo);
    Fr m;
    if constexpr (SHIFTED) {
        uint32_t chi[4] = {r.l[4], r.l[5], r.l[6], r.l[7]};
        m = mul_shifted(d, chi);
    } else {
        m = mul(d, r);
    }
    return add(lo, m);
}

struct BindBatch {
    const Fr* in[kMaxBatchTables];
    Fr* out[kMaxBatchTables];
    size_t half[kMaxBatchTables];  // output length per table (tables of different members may differ in a batch round)
};

// Polynomial::bind_low_to_high (crates/jolt-poly/src/dense.rs:223-303) for blockIdx.y-many tables in one launch:
// out[y] = in[2y] + r*(in[2y+1]-in[2y]).  Algorithmic traffic 96 B per output (64 read + 32 written).
template <bool SHIFTED>
static __global__ __launch_bounds__(kBlock) void k_bind_low_to_high(BindBatch b, Fr r) {
    const Fr* __restrict__ in = b.in[blockIdx.y];
    Fr* __restrict__ out = b.out[blockIdx.y];
    const size_t half = b.half[blockIdx.y];
    size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t y = (size_t)blockIdx.x * kBlock + threadIdx.x; y < half; y += stride) {
        Fr lo = ld_fr(in + 2 * y), hi = ld_fr(in + 2 * y + 1);
        st_fr(out + y, bind_pair<SHIFTED>(lo, hi, r));
    }
}

// Polynomial::bind_high_to_low (dense.rs:188-220): in place, t[i] += r*(t[i+half]-t[i])
template <bool SHIFTED>
static __global__ __launch_bounds__(kBlock) void k_bind_high_to_low(BindBatch b, Fr r) {
    const Fr* __restrict__ in = b.in[blockIdx.y];
    Fr* out = b.out[blockIdx.y];
    const size_t half = b.half[blockIdx.y];
    size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < half; i += stride) {
        Fr lo = ld_fr(in + i), hi = ld_fr(in + i + half);
        st_fr(out + i, bind_pair<SHIFTED>(lo, hi, r));
    }
}

// One tensor step of EqPolynomial::evals (crates/jolt-poly/src/eq.rs:221-231, big-endian): with the table `prev`
// over the first a variables already scaled, out[x] = prev[x >> c] * L[x & (2^c-1)], L = eq over the next c <= 8
// variables (built per block in LDS).  Field multiplication is exact, so this tensor order gives the same entries as
// the reference's layer-by-layer doubling; every entry costs one multiply and the kernel is write-bound.
struct EqChunk {
    Fr r[8];
    int c;
};
static __global__ __launch_bounds__(kBlock) void k_eq_expand(const Fr* __restrict__ prev, Fr* __restrict__ out, size_t out_len, EqChunk ch) {
    __shared__ Fr L[256];
    const int c = ch.c;
    if ((int)threadIdx.x < (1 << c)) {
        Fr acc = Fr::one();
        for (int k = 0; k < c; ++k) {
            int bit = (threadIdx.x >> (c - 1 - k)) & 1;
            Fr f = bit ? ch.r[k] : sub(Fr::one(), ch.r[k]);
            acc = mul(acc, f);
        }
        L[threadIdx.x] = acc;
    }
    __syncthreads();
    size_t stride = (size_t)gridDim.x * kBlock;
    size_t mask = ((size_t)1 << c) - 1;
    for (size_t x = (size_t)blockIdx.x * kBlock + threadIdx.x; x < out_len; x += stride) {
        Fr p = ld_fr(prev + (x >> c));
        st_fr(out + x, mul(p, L[x & mask]));
    }
}

// evals_cached (crates/jolt-poly/src/eq.rs:317-340) for a SHORT point in ONE launch: levels[j] = eq over the first j coordinates, j = 0 .. n, each built from the one before by
// the reference's doubling step (out[2x] = prev[x] (1 - r_j), out[2x + 1] = prev[x] r_j: big-endian), one workgroup, a barrier per level.  The split-eq members cache the levels
// of their two half points (n <= 13): as n launches of k_eq_expand each these were ~200 tiny launches per proof.
constexpr int kEqLevelsMax = 14;
struct EqLevels {
    Fr* level[kEqLevelsMax + 1];
    Fr r[kEqLevelsMax];
    int n;
};
static __global__ __launch_bounds__(kBlock) void k_eq_levels(EqLevels a, Fr scale) {
    if (threadIdx.x == 0) st_fr(a.level[0], scale);
    for (int j = 0; j < a.n; ++j) {
        __threadfence_block();
        __syncthreads();
        const Fr* __restrict__ prev = a.level[j];
        Fr* __restrict__ out = a.level[j + 1];
        const Fr rj = a.r[j];
        const size_t len = (size_t)1 << j;
        for (size_t x = threadIdx.x; x < len; x += kBlock) {
            const Fr p = ld_fr(prev + x), hi = mul(p, rj);
            st_fr(out + 2 * x + 1, hi);
            st_fr(out + 2 * x, sub(p, hi));
        }
    }
}

// LtPolynomial::evaluations (crates/jolt-poly/src/lt.rs:144-156) by the split identity the reference itself states
// (lt.rs:19-21): with r = r_hi || r_lo,  LT(j_hi||j_lo, r) = LT(j_hi, r_hi) + eq(j_hi, r_hi) * LT(j_lo, r_lo).
// out[x] = lt_hi[x>>c] + eq_hi[x>>c] * lt_lo[x & mask]; lt_lo (<= 256 entries) is built per block in LDS from
// the closed form sum_i (1-x_i) r_i eq(x[..i], r[..i]) (lt.rs:126-137).
static __global__ __launch_bounds__(kBlock) void k_lt_expand(const Fr* __restrict__ lt_hi, const Fr* __restrict__ eq_hi, Fr* __restrict__ out,
                                                     size_t out_len, EqChunk ch) {
    __shared__ Fr L[256];
    const int c = ch.c;
    if ((int)threadIdx.x < (1 << c)) {
        Fr lt = Fr::zero(), pre = Fr::one();
        for (int k = 0; k < c; ++k) {
            int bit = (threadIdx.x >> (c - 1 - k)) & 1;
            Fr rk = ch.r[k];
            if (!bit) lt = add(lt, mul(rk, pre));  // (1 - x_k) r_k eq_prefix with x_k = 0
            Fr f = bit ? rk : sub(Fr::one(), rk);
            pre = mul(pre, f);
        }
        L[threadIdx.x] = lt;
    }
    __syncthreads();
    size_t stride = (size_t)gridDim.x * kBlock;
    size_t mask = ((size_t)1 << c) - 1;
    for (size_t x = (size_t)blockIdx.x * kBlock + threadIdx.x; x < out_len; x += stride) {
        Fr h = ld_fr(lt_hi + (x >> c)), e = ld_fr(eq_hi + (x >> c));
        st_fr(out + x, add(h, mul(e, L[x & mask])));
    }
}

// EqPlusOnePolynomial::evals (crates/jolt-poly/src/eq_plus_one.rs:71-130): eq+1(r, j) is nonzero only through the
// decomposition j = (prefix, 1, 0...0) with k trailing zeros:  eq+1[j] = eq(r[..i], prefix) * (1-r[i]) * prod_{m>i} r[m],
// i = n-1-k.  `eq_prefix_tables` is not materialised: eq(r[..i], prefix) = eq_full[j - 2^k] / ... is avoided by using
// the identity eq(r[..i],prefix) = sum over the cleared suffix, i.e. eq_full[j] / (r[i] * prod_{m>i}(1-r[m])) is NOT
// used either (no division): instead the host passes suffix products and the kernel multiplies the prefix eq taken
// from a strided read of the eq table over the first i variables (prefix_tables[i]).
struct EqP1Args {
    const Fr* prefix[33];  // prefix[i] = eq table over r[..i] (2^i entries), i = 0..n-1
    Fr lower[32];          // lower[i] = (1 - r[i]) * prod_{m>i} r[m]
    int n;
};
static __global__ __launch_bounds__(kBlock) void k_eq_plus_one(EqP1Args a, Fr* __restrict__ out, size_t out_len) {
    size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x; j < out_len; j += stride) {
        if (j == 0) { st_fr(out, Fr::zero()); continue; }
        int k = __builtin_ctzll((unsigned long long)j);  // trailing zeros
        int i = a.n - 1 - k;
        Fr e = ld_fr(a.prefix[i] + (j >> (k + 1)));
        st_fr(out + j, mul(e, a.lower[i]));
    }
}

// Ring::from_u64 / from_i64 per entry (crates/jolt-field/src/bn254/mod.rs:265-278): Montgomery form of the integer
static __global__ __launch_bounds__(kBlock) void k_from_u64(const uint64_t* __restrict__ in, Fr* __restrict__ out, size_t n) {
    size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) st_fr(out + i, fr_from_u64(in[i]));
}
static __global__ __launch_bounds__(kBlock) void k_from_i64(const int64_t* __restrict__ in, Fr* __restrict__ out, size_t n) {
    size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        int64_t v = in[i];
        uint64_t mag = v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
        Fr m = fr_from_u64(mag);
        st_fr(out + i, v < 0 ? neg(m) : m);
    }
}

// ---- block-level reduction of NE field accumulators (wave64 shuffles, then LDS across the block's 4 waves) ----
// block_reduce_store_at: the block's NE sums to dst[0 .. NE); block_reduce_store: to the slot of blockIdx.x
template <int NE>
__device__ __forceinline__ void block_reduce_store_at(Fr (&acc)[NE], Fr* __restrict__ dst) {
    __shared__ Fr sm[kBlock / 64][NE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int t = 0; t < NE; ++t) {
            Fr o;
#pragma unroll
            for (int k = 0; k < 8; ++k) o.l[k] = __shfl_xor(acc[t].l[k], off, 64);
            acc[t] = add(acc[t], o);
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t < NE; ++t) sm[wave][t] = acc[t];
    }
    __syncthreads();
    if (threadIdx.x < NE) {
        Fr s = sm[0][threadIdx.x];
        for (int w = 1; w < kBlock / 64; ++w) s = add(s, sm[w][threadIdx.x]);
        st_fr_agent(dst + threadIdx.x, s);
        wait_stores_acked();  // acknowledged before the barrier in front of this workgroup's ticket (finish_member)
    }
}
template <int NE>
__device__ __forceinline__ void block_reduce_store(Fr (&acc)[NE], Fr* __restrict__ partials) {
    block_reduce_store_at<NE>(acc, partials + (size_t)blockIdx.x * NE);
}

// ---- in-kernel completion of a batch round ------------------------------------------------------------------------
// Called by every workgroup after block_reduce_store.  The LAST workgroup of a member (agent-scope release/acquire
// around a ticket counter, MI355X guide G16) sums that member's per-block partials and writes the round sums directly
// into host-mapped pinned memory; the last MEMBER of the round then publishes `seq` to the host flag.  This replaces
// a second-stage kernel + device->host copy + stream synchronise per round with zero extra launches.
constexpr int kGroupTicket = 32;
struct RoundDone {
    uint32_t* counters;   // device memory, zero between rounds
    Fr* results;          // host-mapped pinned memory
    Fr* results_dev;      // device copy of the same sums (send buffer of the multi-GPU all-gather), may be null
    uint64_t* flag;       // host-mapped pinned memory
    uint64_t seq;
    uint32_t group_total; // members in this batch round
};
// LAYOUT_T_MAJOR = false: partials[b*ne + t] from gridDim.x blocks; true: partials[t*nblocks + b] with `expected` tickets.
// Tickets are two-level above kSubTickets workgroups: a same-address device-scope atomic costs ~35-40 ns of serialised time
// on this part (8 XCDs: it is resolved on the memory side), so 2048 workgroups on ONE counter put a ~80 us floor under every
// round kernel however little data it touched (measured: kernel time linear in the workgroup count).  Workgroup p takes a
// ticket on counter p % 64 of its member (one 128-byte line each); the last of each takes one of 64 top-level tickets.
constexpr uint32_t kSubTickets = 64;
constexpr uint32_t kSubTicketStride = 32;   // u32 per sub counter: its own cache line
constexpr uint32_t kSubTicketBase = 64;     // counters[0..31] member tickets, [32] group ticket, sub counters from [64]
constexpr size_t kTicketWords = kSubTicketBase + (size_t)kGroupTicket * kSubTickets * kSubTicketStride;
template <bool LAYOUT_T_MAJOR = false>
__device__ __forceinline__ void finish_member(const Fr* __restrict__ partials, int ne, uint32_t member_ticket, uint32_t slot, const RoundDone& rd,
                                              int nblocks_arg = 0, uint32_t expected = 0, uint32_t pidx = 0xFFFFFFFFu) {
    __shared__ uint32_t s_last;
    __syncthreads();  // this block's partials are written and acknowledged (block_reduce_store: st_fr_agent + wait_stores_acked)
    if (threadIdx.x == 0) {
        // no release fence: the partials were stored device-coherently and their lanes drained vmcnt before the barrier above
        const uint32_t total = LAYOUT_T_MAJOR ? expected : gridDim.x;
        uint32_t last = 0;
        if (total <= kSubTickets) {
            last = atomicAdd(&rd.counters[member_ticket], 1u) == total - 1 ? 1u : 0u;
        } else {
            const uint32_t p = pidx == 0xFFFFFFFFu ? blockIdx.x : pidx;
            const uint32_t r = p % kSubTickets;
            const uint32_t in_sub = (total - r + kSubTickets - 1) / kSubTickets;
            uint32_t* sub = rd.counters + kSubTicketBase + ((size_t)member_ticket * kSubTickets + r) * kSubTicketStride;
            if (atomicAdd(sub, 1u) == in_sub - 1) {
                st_u32_agent(sub, 0u);  // ready for the next round: acknowledged before the ticket that leads to the host flag
                wait_stores_acked();
                last = atomicAdd(&rd.counters[member_ticket], 1u) == kSubTickets - 1 ? 1u : 0u;
            }
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    // one wavefront per sum: the ne sums are reduced side by side instead of one after the other (this epilogue is on the
    // critical path of every round: the host is spinning on the flag)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nblocks = LAYOUT_T_MAJOR ? nblocks_arg : (int)gridDim.x;
    for (int t = wave; t < ne; t += kBlock / 64) {
        Fr s = Fr::zero();
        for (int b = lane; b < nblocks; b += 64)
            s = add(s, ld_fr_agent(partials + (LAYOUT_T_MAJOR ? (size_t)t * nblocks + b : (size_t)b * ne + t)));
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            Fr o;
#pragma unroll
            for (int k = 0; k < 8; ++k) o.l[k] = __shfl_xor(s.l[k], off, 64);
            s = add(s, o);
        }
        if (lane == 0) {
            st_fr_system(rd.results + slot + t, s);  // written through to host memory; the barrier below waits for the write
            if (rd.results_dev) st_fr_system(rd.results_dev + slot + t, s);  // RCCL send buffer: read by other streams / peers
            wait_stores_acked();  // every wave's sums are acknowledged before the barrier below
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        st_u32_agent(&rd.counters[member_ticket], 0u);  // ready for the next round
        wait_stores_acked();
        // No system-scope fence (an L2 write-back of all dirty data on the critical path of the round): this member's sums have
        // been written through and acknowledged before the group ticket is taken, so the flag -- written the same way by
        // whoever takes the last ticket -- is issued after every member's sums.
        uint32_t g = atomicAdd(&rd.counters[kGroupTicket], 1u);
        if (g == rd.group_total - 1) {
            st_u32_agent(&rd.counters[kGroupTicket], 0u);
            wait_stores_acked();  // the next round (possibly on another stream) starts only after the host has seen the flag
            __hip_atomic_store(rd.flag, rd.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

static __global__ void k_fill_fr(Fr* __restrict__ out, Fr v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) st_fr(out + i, v);
}

// second stage: out[t] = sum_b partials[b*ne + t]   (one block)
static __global__ __launch_bounds__(kBlock) void k_reduce_partials(const Fr* __restrict__ partials, int nblocks, int ne, Fr* __restrict__ out) {
    __shared__ Fr sm[kBlock];
    for (int t = 0; t < ne; ++t) {
        Fr s = Fr::zero();
        for (int b = threadIdx.x; b < nblocks; b += kBlock) s = add(s, ld_fr(partials + (size_t)b * ne + t));
        sm[threadIdx.x] = s;
        __syncthreads();
        for (int off = kBlock / 2; off >= 1; off >>= 1) {
            if ((int)threadIdx.x < off) sm[threadIdx.x] = add(sm[threadIdx.x], sm[threadIdx.x + off]);
            __syncthreads();
        }
        if (threadIdx.x == 0) st_fr(out + t, sm[0]);
        __syncthreads();
    }
}

// second stage for a whole batch round: blockIdx.x = member
struct ReduceGroupArgs {
    uint32_t part_off[24];
    uint32_t nblocks[24];
    uint32_t ne[24];
    uint32_t slot[24];
};
static __global__ __launch_bounds__(kBlock) void k_reduce_partials_group(const Fr* __restrict__ partials, ReduceGroupArgs a, Fr* __restrict__ results) {
    __shared__ Fr sm[kBlock / 64];
    const int m = blockIdx.x;
    const Fr* __restrict__ p = partials + a.part_off[m];
    const int nblocks = (int)a.nblocks[m], ne = (int)a.ne[m];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = 0; t < ne; ++t) {
        Fr s = Fr::zero();
        for (int b = threadIdx.x; b < nblocks; b += kBlock) s = add(s, ld_fr(p + (size_t)b * ne + t));
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            Fr o;
#pragma unroll
            for (int k = 0; k < 8; ++k) o.l[k] = __shfl_xor(s.l[k], off, 64);
            s = add(s, o);
        }
        if (lane == 0) sm[wave] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            Fr tot = sm[0];
            for (int w = 1; w < kBlock / 64; ++w) tot = add(tot, sm[w]);
            st_fr(results + a.slot[m] + t, tot);
        }
        __syncthreads();
    }
}

// sum of a table / dot product with a second table (Polynomial::evaluate = dot with the eq table, dense.rs:340-366)
template <bool DOT>
static __global__ __launch_bounds__(kBlock) void k_sum_or_dot(const Fr* __restrict__ a, const Fr* __restrict__ b, size_t n, Fr* __restrict__ partials) {
    Fr acc[1] = {Fr::zero()};
    size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        Fr v = ld_fr(a + i);
        if constexpr (DOT) v = mul(v, ld_fr(b + i));
        acc[0] = add(acc[0], v);
    }
    block_reduce_store<1>(acc, partials);
}

}  // namespace jolt
