// jolt_amd/csrc/msm_kernels.hip.h -- the bucket-method kernels shared by the G1 MSM (msm.hip) and the Dory tier-1 row
// commitments (dory.hip).  A "window" here is any independent bucket set: window w of one MSM, or (row, window) of a
// batch of row MSMs over the same bases, or one chunk of a one-hot column.  Arrays are window-major:
// keys/sorted[w*n + i] (n = points per window), hist/offsets/cursor/buckets[w*(B+1) + |digit|].
#pragma once
#include "g1.hip.h"
#include "fq_limb.hip.h"
#include "poly_kernels.hip.h"

#ifndef JOLT_BUCKET_WAVES
#define JOLT_BUCKET_WAVES 3  /* light bucket kernel: 168 VGPRs + 44 B of scratch at 3 waves per SIMD measured 2 % faster than 177 VGPRs at 2; 4 (128 VGPRs, 208 B scratch) is slower */
#endif
namespace jolt {
namespace msmk {
namespace {  // kernels have internal linkage: each including .hip carries its own copies

#ifndef JOLT_BUCKET_XYZZ
#define JOLT_BUCKET_XYZZ 1  // bucket accumulators in XYZZ coordinates (g1.hip.h); 0: Jacobian madd-2007-bl, for A/B builds
#endif
constexpr int kLaneCap = 128;    // a bucket whose points-per-lane would exceed max(this, 4x the average) is heavy ...
constexpr int kHeavySeg = 1024;  // ... and is summed by one wavefront per segment of this many points, segment sums combined afterwards
constexpr int kPeelMax = 16;     // max rounds of same-key aggregation before falling back to per-lane atomics

__device__ __forceinline__ G1Affine ld_aff(const G1Affine* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    G1Affine r;
    r.x.l[0] = a.x; r.x.l[1] = a.y; r.x.l[2] = a.z; r.x.l[3] = a.w; r.x.l[4] = b.x; r.x.l[5] = b.y; r.x.l[6] = b.z; r.x.l[7] = b.w;
    r.y.l[0] = c.x; r.y.l[1] = c.y; r.y.l[2] = c.z; r.y.l[3] = c.w; r.y.l[4] = d.x; r.y.l[5] = d.y; r.y.l[6] = d.z; r.y.l[7] = d.w;
    return r;
}

// Same-key aggregation inside a wavefront for the histogram / scatter atomics.  Small scalars put half of all points into
// one bucket (the carry window) and the top window of 254-bit scalars has a handful of distinct digits: without this the
// same-address atomics serialise (measured 6 ms of a 2^20 MSM).  Rounds: the lanes sharing the key of the first pending
// lane elect it as their leader; stops after kPeelMax rounds or once a round past the first found no duplicate (random keys).
// Afterwards each lane with `do_atomic` issues ONE atomicAdd(&base[key], count); a lane's slot = leader's old value + rank.
struct WaveAgg {
    int src;         // lane holding the atomic's return value for this lane
    uint32_t rank;   // position among the lanes sharing the key
    uint32_t count;  // lanes folded into this lane's atomic (meaningful when do_atomic)
    bool do_atomic;
};
__device__ __forceinline__ WaveAgg wave_aggregate(uint32_t key, bool valid) {
    const uint32_t lane = threadIdx.x & 63;
    WaveAgg r{(int)lane, 0u, 1u, valid};
    uint64_t todo = __ballot(valid);
    for (int it = 0; it < kPeelMax && todo; ++it) {
        int leader = __ffsll((unsigned long long)todo) - 1;
        uint32_t lkey = (uint32_t)__shfl((int)key, leader, 64);
        uint64_t same = __ballot(valid && key == lkey) & todo;
        if ((same >> lane) & 1) {
            r.src = leader;
            r.rank = (uint32_t)__popcll(same & ((1ull << lane) - 1));
            r.count = (uint32_t)__popcll(same);
            r.do_atomic = (int)lane == leader;
        }
        todo &= ~same;
        if (it >= 1 && __popcll(same) < 2) break;
    }
    return r;
}

// ---- 1. digits + histogram -------------------------------------------------------------------------------------
// keys[w*n + i] = |digit| | (negative << 31); hist[w*(B+1) + |digit|] counts non-zero digits
// (hist == nullptr: keys only -- the histogram is then built per workgroup in LDS by k_msm_hist_lds)
__global__ __launch_bounds__(kBlock) void k_msm_digits(const Fr* __restrict__ scalars, size_t n, int c, int W, uint32_t* __restrict__ keys,
                                                      uint32_t* __restrict__ hist) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < n;
    Fr s = live ? from_mont(ld_fr(scalars + i)) : Fr::zero();
    const uint32_t B = 1u << (c - 1);
    uint32_t carry = 0;
    for (int w = 0; w < W; ++w) {
        int bit = w * c;
        uint32_t raw = 0;
        if (bit < 256) {
            int limb = bit >> 5, off = bit & 31;
            uint64_t two = (uint64_t)s.l[limb] | (limb + 1 < 8 ? (uint64_t)s.l[limb + 1] << 32 : 0ull);
            raw = (uint32_t)(two >> off) & ((1u << c) - 1);
        }
        raw += carry;
        uint32_t mag, negf;
        if (raw > B) { mag = (1u << c) - raw; negf = 1; carry = 1; }
        else { mag = raw; negf = 0; carry = 0; }
        if (live) keys[(size_t)w * n + i] = mag | (negf << 31);
        if (hist) {  // kernel-uniform
            WaveAgg ag = wave_aggregate(mag, live && mag != 0);
            if (ag.do_atomic) atomicAdd(&hist[(size_t)w * (B + 1) + mag], ag.count);
        }
    }
}

// ---- 1b / 3b. counting sort through LDS ------------------------------------------------------------------------------
// One device-scope atomic per key costs ~90 ps on this 8-XCD part (64 M histogram + 64 M scatter atomics were 8.4 of the
// 19.7 ms of a 2^22-term MSM).  A window's B + 1 <= 32769 counters fit in the 160 KiB LDS of a CU, so a workgroup of 1024
// threads counts its slice of the window's keys with LDS atomics and touches global memory once per NON-EMPTY bin: 8x fewer
// global atomics at 16 slices per window.  gridDim = (slices, windows).
constexpr int kSortBlock = 1024;
__device__ __forceinline__ void lds_count_slice(const uint32_t* __restrict__ wkeys, size_t lo, size_t hi, uint32_t* __restrict__ sh) {
    for (size_t base = lo; base < hi; base += kSortBlock) {  // whole wavefronts walk the loop together (ballots inside)
        size_t i = base + threadIdx.x;
        uint32_t mag = i < hi ? wkeys[i] & 0x7FFFFFFFu : 0u;
        WaveAgg ag = wave_aggregate(mag, mag != 0);
        if (ag.do_atomic) atomicAdd(&sh[mag], ag.count);
    }
}
__global__ __launch_bounds__(kSortBlock) void k_msm_hist_lds(const uint32_t* __restrict__ keys, size_t n, uint32_t B, uint32_t* __restrict__ hist) {
    extern __shared__ uint32_t msm_sh[];  // B + 1 counters
    const size_t w = blockIdx.y;
    for (uint32_t b = threadIdx.x; b <= B; b += kSortBlock) msm_sh[b] = 0;
    __syncthreads();
    const size_t per = (n + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    if (lo < hi) lds_count_slice(keys + w * n, lo, hi, msm_sh);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b <= B; b += kSortBlock) {
        uint32_t cnt = msm_sh[b];
        if (cnt && b) atomicAdd(&hist[w * (B + 1) + b], cnt);
    }
}
__global__ __launch_bounds__(kSortBlock) void k_msm_scatter_lds(const uint32_t* __restrict__ keys, size_t n, uint32_t B, uint32_t* __restrict__ cursor,
                                                               uint32_t* __restrict__ sorted) {
    extern __shared__ uint32_t msm_sh[];
    const size_t w = blockIdx.y;
    for (uint32_t b = threadIdx.x; b <= B; b += kSortBlock) msm_sh[b] = 0;
    __syncthreads();
    const size_t per = (n + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    const uint32_t* wkeys = keys + w * n;
    if (lo < hi) lds_count_slice(wkeys, lo, hi, msm_sh);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b <= B; b += kSortBlock) {  // reserve this slice's range of every non-empty bucket
        uint32_t cnt = msm_sh[b];
        msm_sh[b] = cnt && b ? atomicAdd(&cursor[w * (B + 1) + b], cnt) : 0u;
    }
    __syncthreads();
    for (size_t base = lo; base < hi; base += kSortBlock) {
        size_t i = base + threadIdx.x;
        uint32_t key = i < hi ? wkeys[i] : 0u;
        uint32_t mag = key & 0x7FFFFFFFu;
        WaveAgg ag = wave_aggregate(mag, mag != 0);
        uint32_t first = 0;
        if (ag.do_atomic) first = atomicAdd(&msm_sh[mag], ag.count);
        uint32_t pos = (uint32_t)__shfl((int)first, ag.src, 64) + ag.rank;
        if (mag) sorted[w * n + pos] = (uint32_t)i | (key & 0x80000000u);
    }
}

// ---- 2. per-window exclusive scan of the histogram ---------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_msm_scan(const uint32_t* __restrict__ hist, uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursor,
                                                    uint32_t B, uint32_t heavy_threshold, uint32_t* __restrict__ heavy_list,
                                                    uint32_t* __restrict__ heavy_count, uint32_t heavy_cap) {
    __shared__ uint32_t sm[kBlock];
    const int w = blockIdx.x;
    const uint32_t* h = hist + (size_t)w * (B + 1);
    uint32_t per = (B + kBlock) / kBlock;  // entries 0..B inclusive
    uint32_t lo = threadIdx.x * per, hi = min(lo + per, B + 1);
    uint32_t local = 0;
    for (uint32_t k = lo; k < hi; ++k) local += h[k];
    sm[threadIdx.x] = local;
    __syncthreads();
    for (int off = 1; off < kBlock; off <<= 1) {  // Hillis-Steele inclusive scan
        uint32_t v = (int)threadIdx.x >= off ? sm[threadIdx.x - off] : 0;
        __syncthreads();
        sm[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = sm[threadIdx.x] - local;
    for (uint32_t k = lo; k < hi; ++k) {
        uint32_t cnt = h[k];
        offsets[(size_t)w * (B + 1) + k] = run;
        cursor[(size_t)w * (B + 1) + k] = run;
        if (cnt > heavy_threshold) {  // one list entry per segment: (bucket slot, segment index), contiguous per bucket
            uint32_t nseg = (cnt + kHeavySeg - 1) / kHeavySeg;
            uint32_t first = atomicAdd(heavy_count, nseg);
            for (uint32_t sgi = 0; sgi < nseg && first + sgi < heavy_cap; ++sgi) {
                heavy_list[2 * (first + sgi)] = (uint32_t)((size_t)w * (B + 1) + k);
                heavy_list[2 * (first + sgi) + 1] = sgi;
            }
        }
        run += cnt;
    }
}

// ---- 3. scatter into bucket order -------------------------------------------------------------------------------
// (windows beyond the 65535 limit of gridDim.y continue in gridDim.z: window = z * gridDim.y + y, n_windows of them)
__global__ __launch_bounds__(kBlock) void k_msm_scatter(const uint32_t* __restrict__ keys, size_t n, uint32_t B, uint32_t* __restrict__ cursor,
                                                       uint32_t* __restrict__ sorted, size_t n_windows) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t w = (size_t)blockIdx.z * gridDim.y + blockIdx.y;
    if (w >= n_windows) return;  // block-uniform
    uint32_t key = i < n ? keys[(size_t)w * n + i] : 0u;
    uint32_t mag = key & 0x7FFFFFFFu;
    WaveAgg ag = wave_aggregate(mag, mag != 0);
    uint32_t first = 0;
    if (ag.do_atomic) first = atomicAdd(&cursor[(size_t)w * (B + 1) + mag], ag.count);
    uint32_t pos = (uint32_t)__shfl((int)first, ag.src, 64) + ag.rank;
    if (mag) sorted[(size_t)w * n + pos] = (uint32_t)i | (key & 0x80000000u);
}

// Butterfly sum of a G1 accumulator over `width` (power of two <= 64) adjacent lanes; every lane of the wave must call it.
__device__ __forceinline__ G1Jac wave_sum_g1(G1Jac acc, int width) {
    for (int off = width >> 1; off >= 1; off >>= 1) {
        G1Jac o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            o.x.l[k] = (uint32_t)__shfl_xor((int)acc.x.l[k], off, 64);
            o.y.l[k] = (uint32_t)__shfl_xor((int)acc.y.l[k], off, 64);
            o.z.l[k] = (uint32_t)__shfl_xor((int)acc.z.l[k], off, 64);
        }
        acc = g1_add(acc, o);
    }
    return acc;
}

// PIPELINED: the next point's index and coordinates are in flight while the current mixed addition runs.  It pays on the
// long lists of the MSM (2^22 u64 scalars: 6.07 -> 5.2 ms; uniform 254-bit scalars: neutral) and costs occupancy where one lane
// walks a short bucket (Dory rows, ~32 points per bucket: 9.6 -> 12.6 ms), so the row commitments use the plain loop.
template <bool PIPELINED>
__device__ __forceinline__ G1Jac sum_bucket_points(const uint32_t* __restrict__ src, const G1Affine* __restrict__ bases, uint32_t lo, uint32_t hi,
                                                   uint32_t stride) {
#if JOLT_BUCKET_XYZZ
    G1Xyzz acc = g1x_identity();
#define JOLT_BUCKET_ADD(a, p) g1x_add_mixed(a, p)
#define JOLT_BUCKET_OUT(a) g1x_to_jac(a)
#else
    G1Jac acc = g1_identity();
#define JOLT_BUCKET_ADD(a, p) g1_add_mixed(a, p)
#define JOLT_BUCKET_OUT(a) (a)
#endif
    if constexpr (!PIPELINED) {
        for (uint32_t k = lo; k < hi; k += stride) {
            uint32_t v = src[k];
            G1Affine p = ld_aff(bases + (v & 0x7FFFFFFFu));
            if (v >> 31) p.y = neg(p.y);
            acc = JOLT_BUCKET_ADD(acc, p);
        }
        return JOLT_BUCKET_OUT(acc);
    } else {
        if (lo >= hi) return JOLT_BUCKET_OUT(acc);
        uint32_t v = src[lo];
        G1Affine p = ld_aff(bases + (v & 0x7FFFFFFFu));
        for (uint32_t k = lo;;) {
            const uint32_t kn = k + stride;
            const bool more = kn < hi;
            uint32_t vn = 0;
            G1Affine pn = p;
            if (more) {
                vn = src[kn];
                pn = ld_aff(bases + (vn & 0x7FFFFFFFu));
            }
            if (v >> 31) p.y = neg(p.y);
            acc = JOLT_BUCKET_ADD(acc, p);
            if (!more) break;
            v = vn;
            p = pn;
            k = kn;
        }
        return JOLT_BUCKET_OUT(acc);
    }
#undef JOLT_BUCKET_ADD
#undef JOLT_BUCKET_OUT
}

// The same sum with the bases in L-FORM (fq_limb.hip.h: coordinates held as x * 2^261 mod p, i.e. the window tables of msm_fixed.hip) and
// the accumulator in limb-form XYZZ: ~2300 instead of ~3000 VALU instructions per point.  Returns a standard Jacobian point.
struct LformConsts {
    Fq one_l;  // L-form of 1 = the standard Montgomery form of 32
    Fq r256;   // 2^256 mod p = the words of Fq::one(): multiplying an L-form value by it gives the standard form
};
__device__ __forceinline__ G1Jac sum_bucket_points_lform(const uint32_t* __restrict__ src, const G1Affine* __restrict__ bases, uint32_t lo, uint32_t hi, uint32_t stride,
                                                         const LformConsts& lc) {
    G1XyzzL acc = g1xl_identity();
    if (lo >= hi) return g1_identity();
    const FqL one = fql_from_words(lc.one_l);
    uint32_t v = src[lo];
    G1Affine p = ld_aff(bases + (v & 0x7FFFFFFFu));
    for (uint32_t k = lo;;) {  // software pipeline: the next index and point are in flight during the addition
        const uint32_t kn = k + stride;
        const bool more = kn < hi;
        uint32_t vn = 0;
        G1Affine pn = p;
        if (more) {
            vn = src[kn];
            pn = ld_aff(bases + (vn & 0x7FFFFFFFu));
        }
        if (!g1_aff_is_inf(p)) {
            if (v >> 31) p.y = neg(p.y);  // the words are a canonical field element (the L-form of y): negation commutes
            acc = g1xl_add_mixed(acc, fql_from_words(p.x), fql_from_words(p.y), one);
        }
        if (!more) break;
        v = vn;
        p = pn;
        k = kn;
    }
    if (g1xl_is_identity(acc)) return g1_identity();
    const FqL r256 = fql_from_words(lc.r256);
    G1Jac out;  // (X, Y, ZZ, ZZZ) ~ Jacobian (X ZZ^2, Y ZZZ^2, ZZZ)
    out.x = fql_to_std(fql_mul(acc.x, fql_sqr(acc.zz)), r256);
    out.y = fql_to_std(fql_mul(acc.y, fql_sqr(acc.zzz)), r256);
    out.z = fql_to_std(acc.zzz, r256);
    return out;
}

// The same sum for ONE LANE PER LIST (stride 1) with the list's base indices STAGED THROUGH LDS in chunks of CH (round 5).  A lane walks its own list: 4 bytes of a
// line every ~4 us, and between two visits of a line ~30 MB of gathered points stream through the XCD's 4 MB L2 -- the line is gone each time, so a 4-byte index cost a
// whole fabric request (profiles/r04_open_traffic.txt: 545 GB fetched per opening for 200 GB of points and indices; ~99 B per index).  Here a lane copies the next CH
// indices of its list into its own LDS column in one burst (the line is fetched once and used whole, straight away) and the loop reads them from there:
// stage[j * kBlock + threadIdx.x], a column per lane -- conflict-free, no barrier (a lane only ever reads what it wrote itself).
template <int CH>
__device__ __forceinline__ void stage_indices(uint32_t* __restrict__ col, const uint32_t* __restrict__ src, uint32_t first, uint32_t cnt) {
#pragma unroll 1
    for (int j0 = 0; j0 < CH; j0 += 8) {  // eight loads in flight: the budget of 170 registers (3 waves per SIMD) has no room for a whole chunk
        uint32_t t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = first + j0 + j < cnt ? src[first + j0 + j] : 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j) col[(j0 + j) * kBlock] = t[j];
        if (first + j0 + 8 >= cnt) break;
    }
}
template <int CH>
__device__ __forceinline__ G1Jac sum_bucket_points_lform_staged(const uint32_t* __restrict__ src, const G1Affine* __restrict__ bases, uint32_t cnt, const LformConsts& lc,
                                                                uint32_t* __restrict__ col /* this lane's LDS column: entry j at col[j * kBlock] */) {
    if (cnt == 0) return g1_identity();
    G1XyzzL acc = g1xl_identity();
    const FqL one = fql_from_words(lc.one_l);
    stage_indices<CH>(col, src, 0u, cnt);
    uint32_t v = col[0], first = 0;
    G1Affine p = ld_aff(bases + (v & 0x7FFFFFFFu));
    for (uint32_t k = 0;;) {  // software pipeline as in sum_bucket_points_lform: the next point is in flight during the addition
        const uint32_t kn = k + 1;
        const bool more = kn < cnt;
        uint32_t vn = 0;
        G1Affine pn = p;
        if (more) {
            if (kn - first == (uint32_t)CH) {  // chunk used up (wave-uniform for lists of equal length, which is how the bucket order hands them out)
                first = kn;
                stage_indices<CH>(col, src, first, cnt);
            }
            vn = col[(kn - first) * kBlock];
            pn = ld_aff(bases + (vn & 0x7FFFFFFFu));
        }
        if (!g1_aff_is_inf(p)) {
            if (v >> 31) p.y = neg(p.y);
            acc = g1xl_add_mixed(acc, fql_from_words(p.x), fql_from_words(p.y), one);
        }
        if (!more) break;
        v = vn;
        p = pn;
        k = kn;
    }
    if (g1xl_is_identity(acc)) return g1_identity();
    const FqL r256 = fql_from_words(lc.r256);
    G1Jac out;
    out.x = fql_to_std(fql_mul(acc.x, fql_sqr(acc.zz)), r256);
    out.y = fql_to_std(fql_mul(acc.y, fql_sqr(acc.zzz)), r256);
    out.z = fql_to_std(acc.zzz, r256);
    return out;
}

// ---- 4a. light buckets: L adjacent lanes per bucket (L = 1 when there are enough buckets to fill the chip) ----------
template <bool PIPELINED>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(JOLT_BUCKET_WAVES, JOLT_BUCKET_WAVES))) void k_msm_buckets_light(const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offsets,
                                                             const uint32_t* __restrict__ sorted, const G1Affine* __restrict__ bases, size_t n,
                                                             uint32_t B, int L, uint32_t heavy_threshold, G1Jac* __restrict__ buckets,
                                                             size_t n_windows) {
    uint32_t gt = blockIdx.x * kBlock + threadIdx.x;
    uint32_t b = gt / L + 1, sub = gt % L;  // bucket magnitude 1..B
    const size_t wi = (size_t)blockIdx.z * gridDim.y + blockIdx.y;
    if (wi >= n_windows) return;  // block-uniform
    const size_t w = n_windows - 1 - wi;  // top window first: its buckets are the fullest for 254-bit scalars
    bool mine = b <= B;
    size_t slot = (size_t)w * (B + 1) + (mine ? b : 0);
    uint32_t cnt = mine ? hist[slot] : 0;
    if (cnt > heavy_threshold) { cnt = 0; mine = false; }  // the heavy kernels own it
    G1Jac acc = sum_bucket_points<PIPELINED>(sorted + (size_t)w * n + offsets[slot], bases, sub, cnt, (uint32_t)L);
    acc = wave_sum_g1(acc, L);
    if (mine && sub == 0) buckets[slot] = acc;
}

// ---- 4b. heavy buckets: one wavefront per kHeavySeg-point segment, then one wavefront per bucket adds its segment sums ----
template <bool LFORM = false>
__global__ __launch_bounds__(kBlock) void k_msm_buckets_heavy(const uint32_t* __restrict__ heavy_list, const uint32_t* __restrict__ heavy_count,
                                                             const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offsets,
                                                             const uint32_t* __restrict__ sorted, const G1Affine* __restrict__ bases, size_t n,
                                                             uint32_t B, G1Jac* __restrict__ seg_sums, LformConsts lc = LformConsts{}) {
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, n_waves = gridDim.x * (kBlock / 64);
    const uint32_t total = *heavy_count;
    for (uint32_t h = wave; h < total; h += n_waves) {
        uint32_t slot = heavy_list[2 * h], sgi = heavy_list[2 * h + 1];
        uint32_t w = slot / (B + 1);
        uint32_t cnt = hist[slot];
        uint32_t lo = sgi * kHeavySeg, hi = min(lo + (uint32_t)kHeavySeg, cnt);
        G1Jac acc = LFORM ? sum_bucket_points_lform(sorted + (size_t)w * n + offsets[slot], bases, lo + lane, hi, 64u, lc)
                          : sum_bucket_points<true>(sorted + (size_t)w * n + offsets[slot], bases, lo + lane, hi, 64u);
        acc = wave_sum_g1(acc, 64);
        if (lane == 0) seg_sums[h] = acc;
    }
}
__global__ __launch_bounds__(kBlock) void k_msm_heavy_combine(const uint32_t* __restrict__ heavy_list, const uint32_t* __restrict__ heavy_count,
                                                             const uint32_t* __restrict__ hist, const G1Jac* __restrict__ seg_sums,
                                                             G1Jac* __restrict__ buckets) {
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, n_waves = gridDim.x * (kBlock / 64);
    const uint32_t total = *heavy_count;
    for (uint32_t h = wave; h < total; h += n_waves) {
        if (heavy_list[2 * h + 1] != 0) continue;  // only the first segment entry of a bucket combines (wave-uniform)
        uint32_t slot = heavy_list[2 * h];
        uint32_t nseg = (hist[slot] + kHeavySeg - 1) / kHeavySeg;
        G1Jac acc = g1_identity();
        for (uint32_t k = lane; k < nseg; k += 64) acc = g1_add(acc, seg_sums[h + k]);
        acc = wave_sum_g1(acc, 64);
        if (lane == 0) buckets[slot] = acc;
    }
}

// ---- 5. window reduction: partial[w][blockIdx.x] = sum over this block's bucket range of b * bucket[b] ------------------
__global__ __launch_bounds__(kBlock) void k_msm_window_reduce(const G1Jac* __restrict__ buckets, uint32_t B, uint32_t G, G1Jac* __restrict__ partial) {
    __shared__ G1Jac sm[kBlock];
    const int w = blockIdx.y;
    uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    uint64_t lo = (uint64_t)t * G + 1, hi = lo + G - 1;  // inclusive bucket range
    G1Jac contrib = g1_identity();
    if (lo <= B) {
        if (hi > B) hi = B;
        const G1Jac* bk = buckets + (size_t)w * (B + 1);
        G1Jac running = g1_identity(), acc = g1_identity();
        for (uint64_t b = hi; b >= lo; --b) {
            running = g1_add(running, bk[b]);
            acc = g1_add(acc, running);
        }
        // sum (b - lo + 1) B_b = acc  ->  sum b B_b = acc + (lo - 1) * running
        contrib = g1_add(acc, g1_mul_small(running, (uint32_t)(lo - 1)));  // skips the leading zero bits
    }
    sm[threadIdx.x] = contrib;
    __syncthreads();
    for (int off = kBlock / 2; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) sm[threadIdx.x] = g1_add(sm[threadIdx.x], sm[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)w * gridDim.x + blockIdx.x] = sm[0];
}

// ---- 5b. one wavefront per window adds that window's nb block partials ---------------------------------------------------
__global__ __launch_bounds__(64) void k_msm_window_combine(const G1Jac* __restrict__ partial, uint32_t nb, G1Jac* __restrict__ window_sums) {
    const int w = blockIdx.x;
    G1Jac acc = g1_identity();
    for (uint32_t k = threadIdx.x; k < nb; k += 64) acc = g1_add(acc, partial[(size_t)w * nb + k]);
    acc = wave_sum_g1(acc, 64);
    if (threadIdx.x == 0) window_sums[w] = acc;
}

}  // namespace
}  // namespace msmk
}  // namespace jolt
