// jolt_amd/csrc/msm_fixed.hip -- fixed-base G1 MSM over window-precomputed bases (the HyperKZG SRS never changes between proofs).
//
// JoltGroup::msm (crates/jolt-crypto/src/ec/group.rs:63-70) as called by kzg_commit / kzg_open_batch
// (crates/jolt-hyperkzg/src/kzg.rs:15-27,108-116) always multiplies prefixes of ONE long-lived base vector (g1_powers).  With
// pre[w][i] = 2^(c*w) * P_i resident (jolt_srs_precompute_windows), digit w of scalar i addresses base pre[w][i] and ALL windows
// share one bucket set:
//     sum_i s_i P_i = sum_b b * ( sum_{(i,w): |digit_w(s_i)| = b} sign * pre[w][i] )
// The bucket additions are the same W*n as in the per-window method, but there is ONE bucket reduction of 2^(c-1) buckets instead
// of W of them, so c can grow to 24 bits (W = 11 windows instead of 16 at c = 16: 31 % fewer point additions for 254-bit scalars)
// and the host-side Horner over the windows disappears.  Cost: W copies of the bases in HBM (11 x 4 GiB for 2^26 points).
//
// Pipeline (integer VALU work, no MFMA; DESIGN.md 3.5c has the measurements):
//   1. histogram: k_fx_hist_scalars -- the W signed c-bit digits of every scalar (scalars above r/2 negated, unsigned top window) recomputed from the scalar
//                 itself and counted per SEGMENT of 256 buckets in per-workgroup LDS histograms; no key array (the 8-byte-entry path behind JOLT_FX_SOA=0
//                 keeps k_fx_digits / k_fx_hist)
//   2. partition: two coalesced passes -- groups of 128 segments straight from the scalars (k_fx_partition_groups_scalars), then the segments of each group
//                 (k_fx_partition_segments_soa): a workgroup ranks a tile by bin with LDS atomics, lays it out bin-major in LDS and copies it out; entries travel
//                 as 4-byte base index | sign plus one byte of bucket-in-segment
//   3. segments : ONE workgroup per segment sorts it by bucket INSIDE the LDS of its CU and copies the sorted base indices out coalesced
//                 (k_fx_segment_sort_staged), emitting the bucket table, the heavy list and the length classes
//   4. order    : buckets handed out in order of decreasing list length (k_fx_order): a wavefront's lanes sum lists of equal length
//   5. buckets  : one lane per bucket, XYZZ accumulator in limb form over the L-form tables (k_fx_buckets_ordered); over-full buckets (repeated scalars, the
//                 carry window of 64-bit witness scalars) as segments of 128 entries, one lane each, folded per bucket (k_fx_heavy_segments / _combine)
//   6. reduce   : sum_b b * B_b by rows and columns of the bucket matrix (k_fx_red_cols / _rows / _fold) and two small running-sum reductions
// With pair_shift the phases 5-6 run twice over the same sorted lists, the second time against the tables moved by pair_shift points (two MSMs over one set of
// scalars: the witness commitments at r and -r of a HyperKZG opening).
#include <algorithm>
#include <cstdlib>

#include "ctx.hpp"
#include "msm_kernels.hip.h"
#include "srs.hpp"

using namespace jolt;
using namespace jolt::msmk;

struct MsmJob {
    size_t n = 0;
    int lane = 0, c = 0, W = 0;
    uint32_t nb = 0;
    int results = 1;  // 2: a pair of MSMs over the same scalars (jolt_internal_msm_fixed_enqueue with pair_shift): the second result follows the first in msm_host
};

namespace {

// buckets per segment = 2^LO: 256 up to 24-bit windows (a whole sorted segment then fits the LDS of a CU: k_fx_segment_sort_staged),
// 2048 for 25/26-bit windows (keeps the partition at <= 16385 bins of LDS)
constexpr int kLoBitsMin = 8;
static inline int fx_lo_bits(int c) { return c > 24 ? 11 : 8; }
// Over-full buckets (repeated scalars: the level-1 polynomial of an opening over one-hot columns takes a few thousand distinct values
// ~10^3..10^4 times each; the carry window of 64-bit witness scalars) are cut into segments of kFxHeavySeg entries and ONE LANE sums
// one segment -- the same loop, trip count and gather pattern as a light bucket -- before a wavefront per bucket adds the segment
// sums.  (The per-window method's one-wavefront-per-1024-entries kernel gives every lane 16 entries and a 6-round butterfly of full
// additions: 60 % of the light rate; 28 ms of the configs[2] step were spent there.)
constexpr uint32_t kFxHeavySeg = 128;
// list-length classes of the bucket order: exact lengths up to kClasses - 1; empty and heavy buckets share class 0 (no light work)
constexpr uint32_t kClasses = 512;
__device__ __forceinline__ uint32_t bucket_class(uint32_t count, uint32_t heavy_threshold) {
    return count > heavy_threshold ? 0u : (count < kClasses ? count : kClasses - 1);
}

// next[i] = 2^c * prev[i]
__global__ __launch_bounds__(kBlock) void k_fx_next_window(const G1Affine* __restrict__ prev, G1Affine* __restrict__ next, size_t n, int c) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    G1Jac p = g1_from_affine(ld_aff(prev + i));
    for (int k = 0; k < c; ++k) p = g1_double(p);
    next[i] = g1_to_affine(p);
}

// ---- 1. digits: scalars above (r - 1) / 2 are negated (s P = (r - s)(-P)), c-bit signed windows, UNSIGNED top window -----------------
// After the negation a scalar has at most 253 bits, so W = ceil(253 / c) windows suffice and the top window needs no carry out: its
// digit is taken as is (raw + carry in, < top_max).  With c = 23 the 253 bits are exactly 11 windows -- no short top window piling its n
// digits onto a few thousand buckets (c = 24: 14 bits -> 2^13 buckets) -- and the bucket set is max(2^(c-1), top_max) = 6.34 M
// buckets instead of the 2^25 of c = 26, whose running-sum reduction cost 10.4 ms per MSM however short the MSM was.
// keys[w] = |digit_w| | sign << 31 for one canonical scalar (also built for the host: jolt_host_fx_digits, pinned in the CPU suite)
JOLT_HD void fx_digits_of(Fr s /* canonical integer, not Montgomery */, int c, int W, uint32_t* __restrict__ keys, size_t key_stride) {
    bool flip = false;
#pragma unroll
    for (int j = 7; j >= 0; --j) {  // s > (r - 1) / 2 ?   ((r - 1) / 2 = r >> 1, r odd)
        const uint32_t half = ((uint32_t)FrParams::P[j] >> 1) | (j < 7 ? (uint32_t)FrParams::P[j + 1] << 31 : 0u);
        if (s.l[j] != half) { flip = s.l[j] > half; break; }
    }
    if (flip) {
        uint32_t br = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) s.l[j] = __builtin_subc((uint32_t)FrParams::P[j], s.l[j], br, &br);
    }
    const uint32_t B = 1u << (c - 1), fl = flip ? 0x80000000u : 0u;
    uint32_t carry = 0;
    for (int w = 0; w < W; ++w) {
        const int bit = w * c, limb = bit >> 5, off = bit & 31;
        const uint64_t two = (uint64_t)s.l[limb] | (limb + 1 < 8 ? (uint64_t)s.l[limb + 1] << 32 : 0ull);
        if (w + 1 < W) {
            uint32_t raw = ((uint32_t)(two >> off) & ((1u << c) - 1)) + carry;
            uint32_t mag, negf;
            if (raw > B) { mag = (1u << c) - raw; negf = 0x80000000u; carry = 1; }
            else { mag = raw; negf = 0u; carry = 0; }
            keys[(size_t)w * key_stride] = mag | (negf ^ fl);
        } else {
            keys[(size_t)w * key_stride] = (((uint32_t)(two >> off) & 0x7FFFFFFFu) + carry) | fl;  // everything that is left of the <= 253 bits
        }
    }
}
__global__ __launch_bounds__(kBlock) void k_fx_digits(const Fr* __restrict__ scalars, size_t n, int c, int W, uint32_t* __restrict__ keys) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    fx_digits_of(from_mont(ld_fr(scalars + i)), c, W, keys + i, n);
}

// ---- 2. partition by the high bits of |digit| ------------------------------------------------------------------------------
template <int LO>
__global__ __launch_bounds__(kSortBlock) void k_fx_hist(const uint32_t* __restrict__ keys, size_t total, uint32_t nb1, uint32_t* __restrict__ hist1) {
    extern __shared__ uint32_t fx_sh[];
    for (uint32_t b = threadIdx.x; b < nb1; b += kSortBlock) fx_sh[b] = 0;
    __syncthreads();
    const size_t per = (total + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < total ? lo + per : total;
    for (size_t base = lo; base < hi; base += kSortBlock) {  // whole wavefronts walk the loop together (ballots inside)
        size_t k = base + threadIdx.x;
        uint32_t mag = k < hi ? keys[k] & 0x7FFFFFFFu : 0u;
        WaveAgg ag = wave_aggregate(mag >> LO, mag != 0);
        if (ag.do_atomic) atomicAdd(&fx_sh[mag >> LO], ag.count);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb1; b += kSortBlock) {
        uint32_t cnt = fx_sh[b];
        if (cnt) atomicAdd(&hist1[b], cnt);
    }
}
// exclusive scan of the nb1 segment counts (one workgroup); info[0] = largest segment, info[1] = non-zero digits in total
__global__ __launch_bounds__(kSortBlock) void k_fx_scan(const uint32_t* __restrict__ hist1, uint32_t nb1, uint32_t* __restrict__ offs1, uint32_t* __restrict__ cursor1,
                                                       uint32_t* __restrict__ info, const uint32_t* __restrict__ run_if = nullptr) {
    __shared__ uint32_t sm[kSortBlock];
    __shared__ uint32_t smax[kSortBlock];
    if (run_if && *run_if == 0) return;
    const uint32_t per = (nb1 + kSortBlock - 1) / kSortBlock;
    const uint32_t lo = min(threadIdx.x * per, nb1), hi = min(lo + per, nb1);
    uint32_t local = 0, mx = 0;
    for (uint32_t k = lo; k < hi; ++k) { local += hist1[k]; mx = max(mx, hist1[k]); }
    sm[threadIdx.x] = local;
    smax[threadIdx.x] = mx;
    __syncthreads();
    for (int off = 1; off < kSortBlock; off <<= 1) {
        uint32_t v = (int)threadIdx.x >= off ? sm[threadIdx.x - off] : 0;
        uint32_t m = (int)threadIdx.x >= off ? smax[threadIdx.x - off] : 0;
        __syncthreads();
        sm[threadIdx.x] += v;
        smax[threadIdx.x] = max(smax[threadIdx.x], m);
        __syncthreads();
    }
    uint32_t run = sm[threadIdx.x] - local;
    for (uint32_t k = lo; k < hi; ++k) {
        offs1[k] = run;
        cursor1[k] = run;
        run += hist1[k];
    }
    if (threadIdx.x == kSortBlock - 1) { info[0] = smax[threadIdx.x]; info[1] = sm[threadIdx.x]; info[3] = 0; }  // info[3]: the entries a capacity sort counted (0: info[1] is the count)
}
// entries[pos] = (low bits of |digit|) << 32 | (w * stride + i) | sign << 31, grouped by segment
template <int LO>
__global__ __launch_bounds__(kSortBlock) void k_fx_scatter(const uint32_t* __restrict__ keys, size_t total, size_t n, size_t stride, uint32_t nb1,
                                                          uint32_t* __restrict__ cursor1, uint64_t* __restrict__ entries) {
    extern __shared__ uint32_t fx_sh[];
    for (uint32_t b = threadIdx.x; b < nb1; b += kSortBlock) fx_sh[b] = 0;
    __syncthreads();
    const size_t per = (total + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < total ? lo + per : total;
    for (size_t base = lo; base < hi; base += kSortBlock) {
        size_t k = base + threadIdx.x;
        uint32_t mag = k < hi ? keys[k] & 0x7FFFFFFFu : 0u;
        WaveAgg ag = wave_aggregate(mag >> LO, mag != 0);
        if (ag.do_atomic) atomicAdd(&fx_sh[mag >> LO], ag.count);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb1; b += kSortBlock) {  // reserve this slice's range of every non-empty segment
        uint32_t cnt = fx_sh[b];
        fx_sh[b] = cnt ? atomicAdd(&cursor1[b], cnt) : 0u;
    }
    __syncthreads();
    for (size_t base = lo; base < hi; base += kSortBlock) {
        size_t k = base + threadIdx.x;
        uint32_t key = k < hi ? keys[k] : 0u;
        uint32_t mag = key & 0x7FFFFFFFu;
        WaveAgg ag = wave_aggregate(mag >> LO, mag != 0);
        uint32_t first = 0;
        if (ag.do_atomic) first = atomicAdd(&fx_sh[mag >> LO], ag.count);
        uint32_t pos = (uint32_t)__shfl((int)first, ag.src, 64) + ag.rank;
        if (mag) {
            const size_t w = k / n, i = k - w * n;
            const uint32_t value = (uint32_t)(w * stride + i) | (key & 0x80000000u);
            entries[pos] = ((uint64_t)(mag & ((1u << LO) - 1)) << 32) | value;
        }
    }
}

// ---- 2b. the same partition in two coalesced passes (JOLT_FX_PARTITION=2, the default) ---------------------------------------
// k_fx_scatter writes each 8-byte entry to one of 16385 open segments: with (slices x segments) write fronts nothing merges in L2 and
// the 5.4 GB of entries go out as isolated 8-byte stores (14.6 ms at 2^26 terms, c = 26).  Here a workgroup takes a tile of 8192
// entries, ranks them by bin with LDS atomics, lays the tile out bin-major in LDS and copies it out so that adjacent lanes write
// adjacent addresses (runs of ~64 entries = 512 B per bin and tile): pass 1 over the 129 groups of 128 segments, pass 2 over the 128
// segments inside each group.  Every pass reserves its output range with ONE global atomic per non-empty bin and tile.
constexpr int kPartThreads = 1024, kPartPer = 8, kPartTile = kPartThreads * kPartPer;
constexpr int kGroupBits = 7, kGroupBins = 1 << kGroupBits;  // segments per group
constexpr int kPartBins = 256;                                // >= groups (129) and >= kGroupBins

struct PartShared {
    uint64_t stage[kPartTile];
    uint32_t cnt[kPartBins], lstart[kPartBins], gbase[kPartBins], wsum[4];
};

// one tile: `item[u]` (~0 = dropped) with bin `bin[u]`; cursors[bin] are the global write positions
__device__ __forceinline__ void partition_tile(PartShared& sh, const uint64_t (&item)[kPartPer], const uint32_t (&bin)[kPartPer], uint32_t nbins,
                                               uint32_t* __restrict__ cursors, uint64_t* __restrict__ out, int bin_shift, uint32_t bin_mask) {
    const uint32_t tid = threadIdx.x;
    if (tid < kPartBins) sh.cnt[tid] = 0;
    __syncthreads();
    uint32_t rank[kPartPer];
#pragma unroll
    for (int u = 0; u < kPartPer; ++u) rank[u] = item[u] != ~0ull ? atomicAdd(&sh.cnt[bin[u]], 1u) : 0u;
    __syncthreads();
    uint32_t v = 0, incl = 0;
    if (tid < kPartBins) {  // exclusive scan of the bin counts: wave scans + 4 wave totals
        v = tid < nbins ? sh.cnt[tid] : 0u;
        incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
            if ((int)(tid & 63) >= off) incl += o;
        }
        if ((tid & 63) == 63) sh.wsum[tid >> 6] = incl;
    }
    __syncthreads();
    if (tid < kPartBins) {
        uint32_t before = 0;
        for (uint32_t k = 0; k < (tid >> 6); ++k) before += sh.wsum[k];
        sh.lstart[tid] = before + incl - v;
        sh.gbase[tid] = v ? atomicAdd(&cursors[tid], v) : 0u;
    }
    __syncthreads();
    const uint32_t valid = sh.lstart[kPartBins - 1] + sh.cnt[kPartBins - 1];  // bins >= nbins are empty
#pragma unroll
    for (int u = 0; u < kPartPer; ++u)
        if (item[u] != ~0ull) sh.stage[sh.lstart[bin[u]] + rank[u]] = item[u];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kPartPer; ++u) {
        const uint32_t j = u * kPartThreads + tid;
        if (j < valid) {
            const uint64_t it = sh.stage[j];
            const uint32_t b = ((uint32_t)(it >> 32) >> bin_shift) & bin_mask;
            out[sh.gbase[b] + (j - sh.lstart[b])] = it;
        }
    }
    __syncthreads();
}

// pass 1: keys -> entries grouped by segment group; entry = |digit| << 32 | (w * stride + i) | sign << 31
template <int LO>
__global__ __launch_bounds__(kPartThreads) void k_fx_partition_groups(const uint32_t* __restrict__ keys, size_t total, size_t n, size_t stride, uint32_t n_groups,
                                                                     uint32_t* __restrict__ group_cursor, uint64_t* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char fx_part_raw[];
    PartShared& sh = *reinterpret_cast<PartShared*>(fx_part_raw);
    const size_t n_tiles = (total + kPartTile - 1) / kPartTile;
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        uint64_t item[kPartPer];
        uint32_t bin[kPartPer];
#pragma unroll
        for (int u = 0; u < kPartPer; ++u) {
            const size_t k = t * kPartTile + (size_t)u * kPartThreads + threadIdx.x;
            const uint32_t key = k < total ? keys[k] : 0u;
            const uint32_t mag = key & 0x7FFFFFFFu;
            const size_t w = k / n, i = k - w * n;
            item[u] = mag ? ((uint64_t)mag << 32) | (uint32_t)(w * stride + i) | (key & 0x80000000u) : ~0ull;
            bin[u] = mag >> (LO + kGroupBits);
        }
        partition_tile(sh, item, bin, n_groups, group_cursor, out, LO + kGroupBits, 0xFFFFFFFFu);
    }
}
// group_cursor[g] = offset of the group's first segment
__global__ __launch_bounds__(kPartBins) void k_fx_group_cursors(const uint32_t* __restrict__ offs1, uint32_t n_groups, uint32_t* __restrict__ group_cursor,
                                                               const uint32_t* __restrict__ run_if = nullptr) {
    if (run_if && *run_if == 0) return;
    if (threadIdx.x < n_groups) group_cursor[threadIdx.x] = offs1[threadIdx.x << kGroupBits];
}
// pass 2: inside every group, by segment (cursor1[segment] starts at the segment's offset)
template <int LO>
__global__ __launch_bounds__(kPartThreads) void k_fx_partition_segments(const uint64_t* __restrict__ grouped, const uint32_t* __restrict__ offs1, uint32_t nb1, const uint32_t* __restrict__ info,
                                                                       uint32_t* __restrict__ cursor1, uint64_t* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char fx_part_raw[];
    PartShared& sh = *reinterpret_cast<PartShared*>(fx_part_raw);
    __shared__ uint32_t tiles_before[kPartBins + 1];
    const uint32_t n_groups = (nb1 + kGroupBins - 1) >> kGroupBits;
    const uint32_t total = info[1];  // non-zero digits (k_fx_scan)
    if (threadIdx.x == 0) {  // tiles of the groups, prefix-summed (129 entries: serial is fine)
        uint32_t run = 0;
        for (uint32_t g = 0; g < n_groups; ++g) {
            const uint32_t lo = offs1[g << kGroupBits], hi = ((g + 1) << kGroupBits) < nb1 ? offs1[(g + 1) << kGroupBits] : total;
            tiles_before[g] = run;
            run += (hi - lo + kPartTile - 1) / kPartTile;
        }
        tiles_before[n_groups] = run;
    }
    __syncthreads();
    const uint32_t n_tiles = tiles_before[n_groups];
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        uint32_t g_lo = 0, g_hi = n_groups;  // last group whose first tile is <= t
        while (g_hi - g_lo > 1) {
            const uint32_t mid = (g_lo + g_hi) >> 1;
            if (tiles_before[mid] <= t) g_lo = mid; else g_hi = mid;
        }
        const uint32_t g = g_lo;
        const uint32_t lo = offs1[g << kGroupBits], hi = ((g + 1) << kGroupBits) < nb1 ? offs1[(g + 1) << kGroupBits] : total;
        const uint32_t first = lo + (t - tiles_before[g]) * kPartTile;
        const uint32_t nbins = min((uint32_t)kGroupBins, nb1 - (g << kGroupBits));
        uint64_t item[kPartPer];
        uint32_t bin[kPartPer];
#pragma unroll
        for (int u = 0; u < kPartPer; ++u) {
            const uint32_t k = first + u * kPartThreads + threadIdx.x;
            item[u] = k < hi ? grouped[k] : ~0ull;
            bin[u] = ((uint32_t)(item[u] >> 32) >> LO) & (kGroupBins - 1);
        }
        partition_tile(sh, item, bin, nbins, cursor1 + (g << kGroupBits), out, LO, kGroupBins - 1);
    }
}


// ---- 2c. the same two passes WITHOUT a key array and with split entries (JOLT_FX_SOA, the default for 8-bit segments) --------------
// The digit / histogram / partition / segment-sort phases move ~49 GB per 2^26-term MSM in the form above (4-byte keys written once and read
// twice, 8-byte entries written twice and read up to five times) and run at the HBM rate.  Here the digits are recomputed from the scalars
// where they are needed (one Montgomery multiply per scalar: free next to 32 bytes of traffic), the histogram is fused into the digit
// pass, and an entry is carried as a 4-byte value (base index | sign) plus ONLY the digit bits the remaining passes still need: 15 bits
// (2 bytes) after the group pass, 8 bits (1 byte) after the segment pass, which the segment sort's counting pass then reads alone:
// ~28 GB per 2^26-term MSM.
constexpr int kPartPerS = 12;  // entries per thread of the scalar-fed group pass = windows per scalar (W <= 12: the main tables' 11 windows of 23 bits)
constexpr int kPartPerWide = 13;  // ... and the instantiation for the mid table set's 13 windows of 20 bits (round 4: the mid-length level commitments sorted from the scalars too)
template <int PER, int THREADS = kPartThreads>
struct PartSharedN {
    uint64_t stage[THREADS * PER];
    uint32_t cnt[kPartBins], lstart[kPartBins], gbase[kPartBins], wsum[4];
};
// lim_a / lim_b (capacity regions, section 2d): bin b may be written up to position lim_a[b] (+ lim_b[b]); a tile that would cross it drops the bin's entries and raises
// *overflow -- the exact passes then redo the sort.  lim_a == nullptr: exact offsets, nothing to check.
constexpr uint32_t kFxDropped = 0xFFFFFFFFu;
template <int PER, typename LOW, int THREADS = kPartThreads>
__device__ __forceinline__ void partition_tile_soa(PartSharedN<PER, THREADS>& sh, const uint64_t (&item)[PER], const uint32_t (&bin)[PER], uint32_t nbins, uint32_t* __restrict__ cursors,
                                                   uint32_t* __restrict__ out_val, LOW* __restrict__ out_low, int bin_shift, uint32_t bin_mask, uint32_t low_mask,
                                                   const uint32_t* __restrict__ lim_a = nullptr, const uint32_t* __restrict__ lim_b = nullptr, uint32_t* __restrict__ overflow = nullptr) {
    const uint32_t tid = threadIdx.x;
    if (tid < kPartBins) sh.cnt[tid] = 0;
    __syncthreads();
    uint32_t rank[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) rank[u] = item[u] != ~0ull ? atomicAdd(&sh.cnt[bin[u]], 1u) : 0u;
    __syncthreads();
    uint32_t v = 0, incl = 0;
    if (tid < kPartBins) {
        v = tid < nbins ? sh.cnt[tid] : 0u;
        incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
            if ((int)(tid & 63) >= off) incl += o;
        }
        if ((tid & 63) == 63) sh.wsum[tid >> 6] = incl;
    }
    __syncthreads();
    if (tid < kPartBins) {
        uint32_t before = 0;
        for (uint32_t k = 0; k < (tid >> 6); ++k) before += sh.wsum[k];
        sh.lstart[tid] = before + incl - v;
        uint32_t g = v ? atomicAdd(&cursors[tid], v) : 0u;
        if (lim_a && v) {
            const uint32_t limit = lim_a[tid] + (lim_b ? lim_b[tid] : 0u);
            if (g > limit || v > limit - g) {  // the region is full: nothing of this bin is written, the exact passes take over
                g = kFxDropped;
                atomicOr(overflow, 1u);
            }
        }
        sh.gbase[tid] = g;
    }
    __syncthreads();
    const uint32_t valid = sh.lstart[kPartBins - 1] + sh.cnt[kPartBins - 1];
#pragma unroll
    for (int u = 0; u < PER; ++u)
        if (item[u] != ~0ull) sh.stage[sh.lstart[bin[u]] + rank[u]] = item[u];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const uint32_t j = u * THREADS + tid;
        if (j < valid) {
            const uint64_t it = sh.stage[j];
            const uint32_t hi = (uint32_t)(it >> 32), b = (hi >> bin_shift) & bin_mask;
            const uint32_t gb = sh.gbase[b];
            if (gb != kFxDropped) {
                const uint32_t pos = gb + (j - sh.lstart[b]);
                out_val[pos] = (uint32_t)it;
                out_low[pos] = (LOW)(hi & low_mask);
            }
        }
    }
    __syncthreads();
}

// digits of one scalar straight into the per-workgroup segment histogram (no key array)
template <int LO, int PERS = kPartPerS, bool PLAIN = true>
__global__ __launch_bounds__(kSortBlock) void k_fx_hist_scalars(const Fr* __restrict__ scalars, size_t n, int c, int W, uint32_t nb1, uint32_t* __restrict__ hist1,
                                                               const uint32_t* __restrict__ run_if = nullptr) {
    extern __shared__ uint32_t fx_sh[];
    if (run_if && *run_if == 0) return;  // the fallback of the capacity-region sort (section 2d): only when a region overflowed
    for (uint32_t b = threadIdx.x; b < nb1; b += kSortBlock) fx_sh[b] = 0;
    __syncthreads();
    const size_t per = (((n + gridDim.x - 1) / gridDim.x) + kSortBlock - 1) / kSortBlock * kSortBlock, lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (size_t base = lo; base < hi; base += kSortBlock) {  // whole wavefronts walk the loop together (ballots inside)
        const size_t i = base + threadIdx.x;
        const bool live = i < hi;
        uint32_t keys[PERS];
#pragma unroll
        for (int w = 0; w < PERS; ++w) keys[w] = 0;
        if (live) fx_digits_of(from_mont(ld_fr(scalars + i)), c, W, keys, 1);
#pragma unroll
        for (int w = 0; w < PERS; ++w) {
            if (w >= W) break;  // kernel-uniform
            const uint32_t mag = keys[w] & 0x7FFFFFFFu;
            // plain LDS atomics: the ballot peeling of wave_aggregate cost ~60 instructions per window and scalar (660 of the ~1100 this loop spent per scalar) to save
            // same-address serialisation that the LDS resolves in about the same time when it does occur (runs of equal scalars)
            if (PLAIN) {
                if (mag) atomicAdd(&fx_sh[mag >> LO], 1u);
            } else {
                WaveAgg ag = wave_aggregate(mag >> LO, mag != 0);
                if (ag.do_atomic) atomicAdd(&fx_sh[mag >> LO], ag.count);
            }
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb1; b += kSortBlock) {
        uint32_t cnt = fx_sh[b];
        if (cnt) atomicAdd(&hist1[b], cnt);
    }
}
// pass 1 from the scalars: entries grouped by segment group; value = (w * stride + i) | sign << 31, low = |digit| mod 2^(LO + 7)
constexpr int kPartThreadsS = 1024;  // 1024 x 12 entries = 96 KiB of stage, one workgroup per CU: 3.7 ms per 2^26 terms; 512 threads (three per CU) measured 4.3 ms -- the runs per bin get too short
template <int LO, int PERS = kPartPerS>
__global__ __launch_bounds__(kPartThreadsS) void k_fx_partition_groups_scalars(const Fr* __restrict__ scalars, size_t n, int c, int W, size_t stride, uint32_t n_groups,
                                                                             uint32_t* __restrict__ group_cursor, uint32_t* __restrict__ out_val, uint16_t* __restrict__ out_low,
                                                                             const uint32_t* __restrict__ group_limit = nullptr, uint32_t* __restrict__ overflow = nullptr,
                                                                             const uint32_t* __restrict__ run_if = nullptr) {
    extern __shared__ __align__(16) unsigned char fx_part_raw[];
    PartSharedN<PERS, kPartThreadsS>& sh = *reinterpret_cast<PartSharedN<PERS, kPartThreadsS>*>(fx_part_raw);
    if (run_if && *run_if == 0) return;
    const size_t n_tiles = (n + kPartThreadsS - 1) / kPartThreadsS;
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const size_t i = t * kPartThreadsS + threadIdx.x;
        uint32_t keys[PERS];
#pragma unroll
        for (int w = 0; w < PERS; ++w) keys[w] = 0;
        if (i < n) fx_digits_of(from_mont(ld_fr(scalars + i)), c, W, keys, 1);
        uint64_t item[PERS];
        uint32_t bin[PERS];
#pragma unroll
        for (int u = 0; u < PERS; ++u) {
            const uint32_t key = u < W ? keys[u] : 0u, mag = key & 0x7FFFFFFFu;
            item[u] = mag ? ((uint64_t)mag << 32) | (uint32_t)((size_t)u * stride + i) | (key & 0x80000000u) : ~0ull;
            bin[u] = mag >> (LO + kGroupBits);
        }
        partition_tile_soa<PERS, uint16_t, kPartThreadsS>(sh, item, bin, n_groups, group_cursor, out_val, out_low, LO + kGroupBits, 0xFFFFFFFFu, (1u << (LO + kGroupBits)) - 1,
                                                          group_limit, nullptr, overflow);
    }
}
// pass 2: inside every group, by segment; low = |digit| mod 2^LO afterwards
template <int LO>
__global__ __launch_bounds__(kPartThreads) void k_fx_partition_segments_soa(const uint32_t* __restrict__ g_val, const uint16_t* __restrict__ g_low, const uint32_t* __restrict__ offs1, uint32_t nb1,
                                                                           const uint32_t* __restrict__ info, uint32_t* __restrict__ cursor1, uint32_t* __restrict__ out_val,
                                                                           uint8_t* __restrict__ out_low, const uint32_t* __restrict__ group_end = nullptr,
                                                                           const uint32_t* __restrict__ group_limit = nullptr, const uint32_t* __restrict__ seg_cap = nullptr,
                                                                           uint32_t* __restrict__ overflow = nullptr, const uint32_t* __restrict__ run_if = nullptr) {
    extern __shared__ __align__(16) unsigned char fx_part_raw[];
    PartSharedN<kPartPer>& sh = *reinterpret_cast<PartSharedN<kPartPer>*>(fx_part_raw);
    __shared__ uint32_t tiles_before[kPartBins + 1];
    if (run_if && *run_if == 0) return;
    const uint32_t n_groups = (nb1 + kGroupBins - 1) >> kGroupBits;
    const uint32_t total = info[1];
    // group_end (capacity regions): group g's entries are [offs1[first segment], group_end[g]) -- where pass 1's cursor stopped --, not up to the next group's start
    auto group_hi = [&](uint32_t g) -> uint32_t {
        if (group_end) return min(group_end[g], group_limit[g]);  // (a cursor that ran past its region: overflow, already flagged -- nothing beyond the region is read)
        return ((g + 1) << kGroupBits) < nb1 ? offs1[(g + 1) << kGroupBits] : total;
    };
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t g = 0; g < n_groups; ++g) {
            const uint32_t lo = offs1[g << kGroupBits], hi = max(group_hi(g), lo);
            tiles_before[g] = run;
            run += (hi - lo + kPartTile - 1) / kPartTile;
        }
        tiles_before[n_groups] = run;
    }
    __syncthreads();
    const uint32_t n_tiles = tiles_before[n_groups];
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        uint32_t g_lo = 0, g_hi = n_groups;
        while (g_hi - g_lo > 1) {
            const uint32_t mid = (g_lo + g_hi) >> 1;
            if (tiles_before[mid] <= t) g_lo = mid; else g_hi = mid;
        }
        const uint32_t g = g_lo;
        const uint32_t lo = offs1[g << kGroupBits], hi = max(group_hi(g), lo);
        const uint32_t first = lo + (t - tiles_before[g]) * kPartTile;
        const uint32_t nbins = min((uint32_t)kGroupBins, nb1 - (g << kGroupBits));
        uint64_t item[kPartPer];
        uint32_t bin[kPartPer];
#pragma unroll
        for (int u = 0; u < kPartPer; ++u) {
            const uint32_t k = first + u * kPartThreads + threadIdx.x;
            const uint32_t low = k < hi ? g_low[k] : 0u;
            item[u] = k < hi ? ((uint64_t)low << 32) | g_val[k] : ~0ull;
            bin[u] = (low >> LO) & (kGroupBins - 1);
        }
        partition_tile_soa<kPartPer, uint8_t>(sh, item, bin, nbins, cursor1 + (g << kGroupBits), out_val, out_low, LO, kGroupBins - 1, (1u << LO) - 1,
                                              seg_cap ? offs1 + (g << kGroupBits) : nullptr, seg_cap ? seg_cap + (g << kGroupBits) : nullptr, overflow);
    }
}

// ---- 2d. the same sort WITHOUT the histogram pass, for scalars the caller knows to be uniform field elements (round 5; JOLT_FX_CAPACITY=0 switches it off) ----------
// The histogram pass exists to give every segment its exact place; for uniform scalars the digit counts are predictable -- a bucket below 2^(c-1) collects the signed
// digits of W - 1 windows (2 (W - 1) n / 2^c on average) and, up to top_max, the unsigned top window (n / top_max) -- so every segment gets a REGION of its expected size
// plus 8 standard deviations and 64 entries, the two partition passes run straight from the cursors of those regions, the counts fall out of the cursors afterwards, and
// 1.65 ms of digit recomputation per 2^26-term MSM is not spent.  Nothing depends on the scalars BEING uniform: a pass that would cross a region's end drops the bin,
// raises a flag in device memory, and the exact passes (histogram, scan, both partitions) -- enqueued behind every capacity sort, returning at once while the flag is
// clear -- redo the sort.  No host round trip either way.
struct FxCapModel {
    double n, p_signed, p_top;   // terms; per-bucket probability of a signed lower-window digit / of the top window's digit
    uint32_t half, top_max;      // 2^(c-1); the top window's digits are < top_max
};
JOLT_HD uint32_t fx_segment_capacity(const FxCapModel& m, uint32_t seg, uint32_t seg_buckets) {
    const uint64_t lo = (uint64_t)seg * seg_buckets, hi = lo + seg_buckets;  // buckets [lo, hi)
    const uint64_t below_half = lo < m.half ? (hi < m.half ? hi : m.half) - lo : 0, below_top = lo < m.top_max ? (hi < m.top_max ? hi : m.top_max) - lo : 0;
    const bool has_half = lo <= m.half && m.half < hi;  // |digit| = 2^(c-1) itself: one of the two signs only
    const double mean = m.n * ((double)below_half * m.p_signed + (has_half ? 0.5 * m.p_signed : 0.0) + (double)below_top * m.p_top);
    double root = 0.0;
    if (mean > 0.0) {  // integer square root by Newton steps: the host and the device must agree to the last entry
        root = mean > 1.0 ? mean : 1.0;
        for (int k = 0; k < 40; ++k) root = 0.5 * (root + mean / root);
    }
    const uint64_t cap = (uint64_t)(mean + 8.0 * root) + 64;
    return (uint32_t)((cap + 3) & ~(uint64_t)3);
}
// hist1[seg] = the region size of segment seg (k_fx_scan then turns them into offs1 / cursor1 / info[1] like counts); info[2] = the overflow flag, cleared
__global__ __launch_bounds__(kBlock) void k_fx_capacity_regions(FxCapModel m, uint32_t nb1, uint32_t seg_buckets, uint32_t* __restrict__ hist1, uint32_t* __restrict__ info) {
    const uint32_t seg = blockIdx.x * kBlock + threadIdx.x;
    if (seg == 0) info[2] = 0;
    if (seg < nb1) hist1[seg] = fx_segment_capacity(m, seg, seg_buckets);
}
// group_cursor[g] = start of the group's first region, group_limit[g] = end of its last
__global__ __launch_bounds__(kPartBins) void k_fx_group_regions(const uint32_t* __restrict__ offs1, const uint32_t* __restrict__ cap, uint32_t nb1, uint32_t n_groups,
                                                               uint32_t* __restrict__ group_cursor, uint32_t* __restrict__ group_limit) {
    const uint32_t g = threadIdx.x;
    if (g >= n_groups) return;
    const uint32_t last = min(((g + 1) << kGroupBits), nb1) - 1;
    group_cursor[g] = offs1[g << kGroupBits];
    group_limit[g] = offs1[last] + cap[last];
}
// after the two passes: hist1[seg] = entries the segment received (its region size until now); a cursor beyond its region is an overflow the passes already flagged
__global__ __launch_bounds__(kBlock) void k_fx_counts_from_cursors(const uint32_t* __restrict__ offs1, const uint32_t* __restrict__ cursor1, uint32_t nb1, uint32_t* __restrict__ hist1,
                                                                  uint32_t* __restrict__ info) {
    const uint32_t seg = blockIdx.x * kBlock + threadIdx.x;
    if (seg >= nb1) return;
    const uint32_t cap = hist1[seg], got = cursor1[seg] - offs1[seg];
    if (got > cap) atomicOr(&info[2], 1u);
    hist1[seg] = got > cap ? 0u : got;
    if (got && got <= cap) atomicAdd(&info[3], got);  // the sort's true size (info[1] holds the regions' total): what jolt_msm_profile_buckets_last reports
}
__global__ __launch_bounds__(kBlock) void k_fx_clear_if(uint32_t* __restrict__ words, uint32_t count, const uint32_t* __restrict__ run_if) {
    if (*run_if == 0) return;
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < count) words[i] = 0;
}

// ---- 3. one workgroup per segment: counting sort by the low bits in LDS; emits the bucket table of the shared bucket kernels ----
// hist[b] / offsets[b] for bucket b = seg * 512 + low bits (the layout k_msm_buckets_light / _heavy read with one "window"), the base
// indices of every bucket contiguous in `sorted`, and one heavy-list entry per kHeavySeg points of an over-full bucket (what
// k_msm_scan emits for the per-window method).
template <int LO>
__global__ __launch_bounds__(kBlock) void k_fx_segment_sort(const uint32_t* __restrict__ hist1, const uint32_t* __restrict__ offs1, const uint64_t* __restrict__ entries,
                                                           uint32_t* __restrict__ sorted, uint32_t* __restrict__ hist, uint32_t* __restrict__ offsets,
                                                           uint32_t heavy_threshold, uint32_t* __restrict__ heavy_list, uint32_t* __restrict__ heavy_count,
                                                           uint32_t heavy_cap, uint32_t* __restrict__ class_hist) {
    constexpr uint32_t kSegBuckets = 1u << LO;
    __shared__ uint32_t cls[kClasses];
    for (uint32_t b = threadIdx.x; b < kClasses; b += kBlock) cls[b] = 0;
    constexpr uint32_t PER = kSegBuckets / kBlock;  // counts per thread in the scan
    __shared__ uint32_t cnt[kSegBuckets], cur[kSegBuckets];
    __shared__ uint32_t scan_sm[kBlock];
    const uint32_t seg = blockIdx.x;
    const uint32_t total = hist1[seg];
    const uint32_t base = offs1[seg];
    for (uint32_t b = threadIdx.x; b < kSegBuckets; b += kBlock) cnt[b] = 0;
    __syncthreads();
    // plain LDS atomics, four entries in flight per thread: with >= 512 bins same-address lanes are rare for uniform digits, and where
    // they are not (the carry bucket of small scalars) the LDS serialises 64 lanes in about the time the ballot peeling would take
    for (uint32_t k0 = 0; k0 < total; k0 += 4 * kBlock) {
        uint64_t e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t k = k0 + u * kBlock + threadIdx.x;
            e[u] = k < total ? entries[base + k] : ~0ull;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (e[u] != ~0ull) atomicAdd(&cnt[(uint32_t)(e[u] >> 32) & (kSegBuckets - 1)], 1u);
    }
    __syncthreads();
    {   // exclusive scan of the counts: PER consecutive buckets per thread
        uint32_t c_loc[PER], sum = 0;
#pragma unroll
        for (uint32_t h = 0; h < PER; ++h) { c_loc[h] = cnt[PER * threadIdx.x + h]; sum += c_loc[h]; }
        scan_sm[threadIdx.x] = sum;
        __syncthreads();
        for (int off = 1; off < kBlock; off <<= 1) {
            uint32_t v = (int)threadIdx.x >= off ? scan_sm[threadIdx.x - off] : 0;
            __syncthreads();
            scan_sm[threadIdx.x] += v;
            __syncthreads();
        }
        uint32_t run = scan_sm[threadIdx.x] - sum;
#pragma unroll
        for (uint32_t h = 0; h < PER; ++h) {
            const uint32_t slot = seg * kSegBuckets + PER * threadIdx.x + h, c = c_loc[h], st = base + run;
            cur[PER * threadIdx.x + h] = run;
            run += c;
            hist[slot] = c;
            offsets[slot] = st;
            atomicAdd(&cls[bucket_class(c, heavy_threshold)], 1u);
            if (c > heavy_threshold) {
                const uint32_t nseg = (c + kFxHeavySeg - 1) / kFxHeavySeg;
                const uint32_t first = atomicAdd(heavy_count, nseg);
                for (uint32_t sgi = 0; sgi < nseg && first + sgi < heavy_cap; ++sgi) {
                    heavy_list[2 * (first + sgi)] = slot;
                    heavy_list[2 * (first + sgi) + 1] = sgi;
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < kClasses; b += kBlock)
        if (cls[b]) atomicAdd(&class_hist[b], cls[b]);
    uint32_t* out = sorted + base;
    for (uint32_t k0 = 0; k0 < total; k0 += 4 * kBlock) {
        uint64_t e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t k = k0 + u * kBlock + threadIdx.x;
            e[u] = k < total ? entries[base + k] : ~0ull;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (e[u] != ~0ull) out[atomicAdd(&cur[(uint32_t)(e[u] >> 32) & (kSegBuckets - 1)], 1u)] = (uint32_t)e[u];
    }
}


// ---- 3'. the same with the sorted segment STAGED IN LDS (LO = 8: ~30 K base indices per segment at 2^26 terms = 120 KB) --------------
// k_fx_segment_sort scatters each 4-byte base index straight to global memory: with ~2000 segments in flight (0.5 GB of open output)
// nothing merges in L2 and the 738 M isolated stores made the kernel the slowest of the sort (11.3 ms at 1.3 TB/s).  Here one
// 1024-thread workgroup owns the LDS of a CU, scatters into it and copies the finished segment out with adjacent lanes on adjacent
// addresses.  Segments that do not fit (skewed digits) fall back to the direct scatter.
constexpr int kSegThreads = 1024;
template <int LO, bool SOA = false>
__global__ __launch_bounds__(kSegThreads) void k_fx_segment_sort_staged(const uint32_t* __restrict__ hist1, const uint32_t* __restrict__ offs1, const uint64_t* __restrict__ entries,
                                                                       const uint32_t* __restrict__ e_val, const uint8_t* __restrict__ e_low,
                                                                       uint32_t* __restrict__ sorted, uint32_t* __restrict__ hist, uint32_t* __restrict__ offsets,
                                                                       uint32_t heavy_threshold, uint32_t* __restrict__ heavy_list, uint32_t* __restrict__ heavy_count,
                                                                       uint32_t heavy_cap, uint32_t* __restrict__ class_hist, uint32_t stage_cap) {
    constexpr uint32_t kSegBuckets = 1u << LO;
    static_assert(kSegBuckets <= (uint32_t)kSegThreads, "one bucket per thread in the scan");
    extern __shared__ uint32_t fx_stage[];
    __shared__ uint32_t cnt[kSegBuckets], cur[kSegBuckets], first_pos[kSegBuckets], cls[kClasses], wsum[kSegThreads / 64], s_max;
    const uint32_t seg = blockIdx.x, tid = threadIdx.x;
    const uint32_t total = hist1[seg], base = offs1[seg];
    if (tid < kSegBuckets) cnt[tid] = 0;
    if (tid < kClasses) cls[tid] = 0;
    if (tid == 0) s_max = 0;
    __syncthreads();
    // SOA: a thread takes the aligned 4-entry word g / 4 (one 4-byte load of the low bytes, one 16-byte load of the values) and keeps the entries of [base, base + total)
    const uint32_t word_lo = base >> 2, word_hi = (base + total + 3) >> 2;
    if constexpr (SOA) {
        const uint32_t* low_words = reinterpret_cast<const uint32_t*>(e_low);
        for (uint32_t wd = word_lo + tid; wd < word_hi; wd += kSegThreads) {
            const uint32_t lows = low_words[wd];  // the counting pass reads one byte per entry
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t g = 4 * wd + u;
                if (g >= base && g < base + total) atomicAdd(&cnt[(lows >> (8 * u)) & 0xFFu], 1u);
            }
        }
    } else {
        for (uint32_t k0 = 0; k0 < total; k0 += 4 * kSegThreads) {
            uint32_t bk[4];  // bucket inside the segment, or ~0
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t k = k0 + u * kSegThreads + tid;
                bk[u] = k < total ? (uint32_t)(entries[base + k] >> 32) & (kSegBuckets - 1) : ~0u;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (bk[u] != ~0u) atomicAdd(&cnt[bk[u]], 1u);
        }
    }
    __syncthreads();
    uint32_t c = 0, incl = 0;
    if (tid < kSegBuckets) {  // exclusive scan of the bucket counts: one bucket per thread, wave scans + wave totals
        c = cnt[tid];
        incl = c;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
            if ((int)(tid & 63) >= off) incl += o;
        }
        if ((tid & 63) == 63) wsum[tid >> 6] = incl;
        atomicMax(&s_max, c);
    }
    __syncthreads();
    if (tid < kSegBuckets) {
        uint32_t before = 0;
        for (uint32_t k = 0; k < (tid >> 6); ++k) before += wsum[k];
        const uint32_t excl = before + incl - c, slot = seg * kSegBuckets + tid;
        cur[tid] = excl;
        first_pos[tid] = excl;
        hist[slot] = c;
        offsets[slot] = base + excl;
        atomicAdd(&cls[bucket_class(c, heavy_threshold)], 1u);
        if (c > heavy_threshold) {
            const uint32_t nseg = (c + kFxHeavySeg - 1) / kFxHeavySeg;
            const uint32_t first = atomicAdd(heavy_count, nseg);
            for (uint32_t sgi = 0; sgi < nseg && first + sgi < heavy_cap; ++sgi) {
                heavy_list[2 * (first + sgi)] = slot;
                heavy_list[2 * (first + sgi) + 1] = sgi;
            }
        }
    }
    __syncthreads();
    if (tid < kClasses && cls[tid]) atomicAdd(&class_hist[tid], cls[tid]);
    // The staged output is produced in windows of `span` positions: window h takes the buckets whose first position lies in
    // [h * span, (h + 1) * span); their entries end before (h + 1) * span + (largest bucket), which fits the stage by the choice of span.
    // A segment of 43.7 K entries (the low 2^22 buckets at c = 23, 2^26 terms) needs two windows of the 37 K-entry stage.
    const uint32_t largest = s_max;
    const bool staged = stage_cap > 2 * largest && stage_cap >= 4096;  // block-uniform; else: the direct scatter
    const uint32_t span = staged ? stage_cap - largest : total + 1;
    uint32_t* out = sorted + base;
    const uint32_t n_stage_windows = staged ? (total + span - 1) / span : 1;
    uint32_t copied = 0;  // positions already written out (the previous window's last bucket may reach into this window's range)
    for (uint32_t wi = 0; wi < n_stage_windows; ++wi) {
        const uint32_t w_lo = wi * span;
        for (uint32_t k0 = 0; k0 < (SOA ? (word_hi - word_lo) * 4 : total); k0 += 4 * kSegThreads) {
            uint32_t bks[4], vals[4];
            if constexpr (SOA) {
                const uint32_t wd = word_lo + k0 / 4 + tid;
                uint32_t lows = 0;
                uint4 v4 = make_uint4(0u, 0u, 0u, 0u);
                if (wd < word_hi) {
                    lows = reinterpret_cast<const uint32_t*>(e_low)[wd];
                    v4 = reinterpret_cast<const uint4*>(e_val)[wd];
                }
                vals[0] = v4.x; vals[1] = v4.y; vals[2] = v4.z; vals[3] = v4.w;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t g = 4 * wd + u;
                    bks[u] = (wd < word_hi && g >= base && g < base + total) ? (lows >> (8 * u)) & 0xFFu : ~0u;
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t k = k0 + u * kSegThreads + tid;
                    const uint64_t e = k < total ? entries[base + k] : ~0ull;
                    bks[u] = e != ~0ull ? (uint32_t)(e >> 32) & (kSegBuckets - 1) : ~0u;
                    vals[u] = (uint32_t)e;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (bks[u] == ~0u) continue;
                const uint32_t bk = bks[u];
                if (staged) {
                    const uint32_t fp = first_pos[bk];
                    if (fp < w_lo || fp >= w_lo + span) continue;
                    fx_stage[atomicAdd(&cur[bk], 1u) - w_lo] = vals[u];
                } else {
                    out[atomicAdd(&cur[bk], 1u)] = vals[u];
                }
            }
        }
        if (!staged) break;
        __syncthreads();
        // this window's entries: from w_lo to the end of its last bucket = the first position of the next window's first bucket
        uint32_t w_hi = total;
        {   // the smallest first position >= w_lo + span (positions are increasing in the bucket index)
            __shared__ uint32_t s_hi;
            if (tid == 0) s_hi = total;
            __syncthreads();
            if (tid < kSegBuckets && first_pos[tid] >= w_lo + span) atomicMin(&s_hi, first_pos[tid]);
            __syncthreads();
            w_hi = s_hi;
        }
        for (uint32_t k = copied + tid; k < w_hi; k += kSegThreads) out[k] = fx_stage[k - w_lo];
        copied = w_hi;
        __syncthreads();
    }
}

// ---- 3b. bucket order: longest lists first, equal lengths side by side -----------------------------------------------------
// One lane sums one bucket, so a wavefront takes as long as its fullest bucket: with Poisson-distributed list lengths (mean 88 at
// c = 24, 2^26 terms) the 64 lanes of a wavefront idle ~22 % of the time when buckets are taken in index order.  Buckets are therefore
// handed out in order of decreasing length (counting sort over kClasses length classes; empty and heavy buckets last), which makes the
// trip count wave-uniform.  8 M bucket ids per MSM: noise next to the 738 M additions it balances.
__global__ __launch_bounds__(kClasses) void k_fx_order_scan(const uint32_t* __restrict__ class_hist, uint32_t* __restrict__ class_cursor) {
    __shared__ uint32_t sm[kClasses];
    const uint32_t mine = class_hist[kClasses - 1 - threadIdx.x];  // thread 0 = longest class
    sm[threadIdx.x] = mine;
    __syncthreads();
    for (int off = 1; off < (int)kClasses; off <<= 1) {
        uint32_t v = (int)threadIdx.x >= off ? sm[threadIdx.x - off] : 0;
        __syncthreads();
        sm[threadIdx.x] += v;
        __syncthreads();
    }
    class_cursor[kClasses - 1 - threadIdx.x] = sm[threadIdx.x] - mine;
}
constexpr uint32_t kOrderPer = 8;  // buckets per thread
__global__ __launch_bounds__(kBlock) void k_fx_order(const uint32_t* __restrict__ hist, uint32_t n_buckets, uint32_t heavy_threshold, uint32_t* __restrict__ class_cursor,
                                                    uint32_t* __restrict__ order) {
    __shared__ uint32_t cls[kClasses];
    for (uint32_t b = threadIdx.x; b < kClasses; b += kBlock) cls[b] = 0;
    __syncthreads();
    const uint32_t first = blockIdx.x * (kBlock * kOrderPer);
    uint32_t k_cls[kOrderPer];
#pragma unroll
    for (uint32_t u = 0; u < kOrderPer; ++u) {
        const uint32_t slot = first + u * kBlock + threadIdx.x;
        k_cls[u] = slot < n_buckets ? bucket_class(hist[slot], heavy_threshold) : kClasses;
        if (k_cls[u] < kClasses) atomicAdd(&cls[k_cls[u]], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < kClasses; b += kBlock) {
        const uint32_t cnt = cls[b];
        cls[b] = cnt ? atomicAdd(&class_cursor[b], cnt) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < kOrderPer; ++u)
        if (k_cls[u] < kClasses) order[atomicAdd(&cls[k_cls[u]], 1u)] = first + u * kBlock + threadIdx.x;
}

// ---- 4. light buckets in that order (the heavy ones keep k_msm_buckets_heavy / _heavy_combine) ---------------------------
template <bool LFORM>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(JOLT_BUCKET_WAVES, JOLT_BUCKET_WAVES))) void k_fx_buckets_ordered(
    const uint32_t* __restrict__ order, uint32_t n_buckets, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ sorted,
    const G1Affine* __restrict__ bases, uint32_t heavy_threshold, G1Jac* __restrict__ buckets, LformConsts lc) {
    const uint32_t gt = blockIdx.x * kBlock + threadIdx.x;
    if (gt >= n_buckets) return;
    const uint32_t slot = order[gt];
    const uint32_t cnt = hist[slot];
    if (cnt == 0 || cnt > heavy_threshold) return;  // empty: the memset identity stands; heavy: the segmented kernels own it
    buckets[slot] = LFORM ? sum_bucket_points_lform(sorted + offsets[slot], bases, 0u, cnt, 1u, lc) : sum_bucket_points<true>(sorted + offsets[slot], bases, 0u, cnt, 1u);
}
// ... with the base indices staged through LDS (msm_kernels.hip.h: sum_bucket_points_lform_staged) -- the default for L-form tables; JOLT_FX_STAGE_IDX=0 keeps the
// kernel above for an A/B.  kFxIdxChunk indices per lane and refill: 32 = one 128-byte line; 3 workgroups of 256 lanes per CU hold 96 KB of the CU's 160 KB.
constexpr int kFxIdxChunk = 32;
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(JOLT_BUCKET_WAVES, JOLT_BUCKET_WAVES))) void k_fx_buckets_ordered_staged(
    const uint32_t* __restrict__ order, uint32_t n_buckets, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ sorted,
    const G1Affine* __restrict__ bases, uint32_t heavy_threshold, G1Jac* __restrict__ buckets, LformConsts lc) {
    __shared__ uint32_t idx_stage[kFxIdxChunk * kBlock];
    const uint32_t gt = blockIdx.x * kBlock + threadIdx.x;
    if (gt >= n_buckets) return;
    const uint32_t slot = order[gt];
    const uint32_t cnt = hist[slot];
    if (cnt == 0 || cnt > heavy_threshold) return;
    buckets[slot] = sum_bucket_points_lform_staged<kFxIdxChunk>(sorted + offsets[slot], bases, cnt, lc, idx_stage + threadIdx.x);
}
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(JOLT_BUCKET_WAVES, JOLT_BUCKET_WAVES))) void k_fx_heavy_segments_staged(
    const uint32_t* __restrict__ heavy_list, const uint32_t* __restrict__ heavy_count, uint32_t heavy_cap, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offsets,
    const uint32_t* __restrict__ sorted, const G1Affine* __restrict__ bases, G1Jac* __restrict__ seg_sums, LformConsts lc) {
    __shared__ uint32_t idx_stage[kFxIdxChunk * kBlock];
    const uint32_t total = min(*heavy_count, heavy_cap);
    for (uint32_t h = blockIdx.x * kBlock + threadIdx.x; h < total; h += gridDim.x * kBlock) {
        const uint32_t slot = heavy_list[2 * h], sgi = heavy_list[2 * h + 1];
        const uint32_t cnt = hist[slot], lo = sgi * kFxHeavySeg, len = min(kFxHeavySeg, cnt - lo);
        seg_sums[h] = sum_bucket_points_lform_staged<kFxIdxChunk>(sorted + offsets[slot] + lo, bases, len, lc, idx_stage + threadIdx.x);
    }
}

// ---- 4b. heavy buckets: one lane per kFxHeavySeg-entry segment, then one wavefront per bucket over its segment sums -------------
template <bool LFORM>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(JOLT_BUCKET_WAVES, JOLT_BUCKET_WAVES))) void k_fx_heavy_segments(
    const uint32_t* __restrict__ heavy_list, const uint32_t* __restrict__ heavy_count, uint32_t heavy_cap, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offsets,
    const uint32_t* __restrict__ sorted, const G1Affine* __restrict__ bases, G1Jac* __restrict__ seg_sums, LformConsts lc) {
    const uint32_t total = min(*heavy_count, heavy_cap);
    for (uint32_t h = blockIdx.x * kBlock + threadIdx.x; h < total; h += gridDim.x * kBlock) {
        const uint32_t slot = heavy_list[2 * h], sgi = heavy_list[2 * h + 1];
        const uint32_t cnt = hist[slot], lo = sgi * kFxHeavySeg, len = min(kFxHeavySeg, cnt - lo);
        const uint32_t* src = sorted + offsets[slot] + lo;
        seg_sums[h] = LFORM ? sum_bucket_points_lform(src, bases, 0u, len, 1u, lc) : sum_bucket_points<true>(src, bases, 0u, len, 1u);
    }
}
__global__ __launch_bounds__(kBlock) void k_fx_heavy_combine(const uint32_t* __restrict__ heavy_list, const uint32_t* __restrict__ heavy_count, uint32_t heavy_cap,
                                                            const uint32_t* __restrict__ hist, const G1Jac* __restrict__ seg_sums, G1Jac* __restrict__ buckets) {
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, n_waves = gridDim.x * (kBlock / 64);
    const uint32_t total = min(*heavy_count, heavy_cap);
    // the segments of a bucket are adjacent list entries; the wave that finds a bucket's FIRST segment among its 64 entries adds that bucket
    for (uint32_t h0 = wave * 64; h0 < total; h0 += n_waves * 64) {
        const uint32_t h = h0 + lane;
        uint64_t firsts = __ballot(h < total && heavy_list[2 * h + 1] == 0);
        while (firsts) {
            const int l = __ffsll((unsigned long long)firsts) - 1;
            firsts &= firsts - 1;
            const uint32_t hb = h0 + (uint32_t)l, slot = heavy_list[2 * hb];
            const uint32_t nseg = (hist[slot] + kFxHeavySeg - 1) / kFxHeavySeg;
            G1Jac acc = g1_identity();
            for (uint32_t k = lane; k < nseg && hb + k < total; k += 64) acc = g1_add(acc, seg_sums[hb + k]);
            acc = wave_sum_g1(acc, 64);
            if (lane == 0) buckets[slot] = acc;
        }
    }
}

// ---- 5. reduction sum_b b * B_b of ONE large bucket set, by rows and columns ------------------------------------------------------
// With b = h * 2^S + l:   sum_b b B_b = 2^S * sum_h h R_h + sum_l l C_l,   R_h = sum_l B_(h,l) (row sums),  C_l = sum_h B_(h,l) (column sums).
// Two additions per bucket, all of them in independent chains of 16 (k_fx_red_cols: adjacent lanes on adjacent buckets; k_fx_red_rows: a
// lane on 16 consecutive buckets), then wave-per-item folds of the partial sums and two SMALL running-sum reductions (2^S columns,
// B / 2^S rows) through the shared kernels; the two weighted sums leave as "windows" 0 and 1 of an S-bit Horner on the host.  The
// running-sum kernel it replaces for B >= 2^16 gave every thread 24 buckets: a serial chain of 48 additions plus a 23-bit
// double-and-add by the range offset, at one wavefront per SIMD -- 3.1 ms per MSM however short, 38 ms of the configs[2] step.
constexpr int kRedS = 11;
constexpr uint32_t kRedCols = 1u << kRedS, kRedRC = 16, kRedCW = 16;
__global__ __launch_bounds__(kBlock) void k_fx_red_cols(const G1Jac* __restrict__ buckets, uint32_t B, uint32_t H, G1Jac* __restrict__ colpart) {
    const uint32_t l = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t h0 = blockIdx.y * kRedRC, h1 = min(h0 + kRedRC, H);
    G1Jac acc = g1_identity();
    for (uint32_t h = h0; h < h1; ++h) {
        const uint32_t b = (h << kRedS) + l;
        if (b <= B) acc = g1_add(acc, buckets[b]);  // bucket 0 is never written: the memset identity
    }
    colpart[(size_t)blockIdx.y * kRedCols + l] = acc;
}
__global__ __launch_bounds__(kBlock) void k_fx_red_rows(const G1Jac* __restrict__ buckets, uint32_t B, uint32_t H, G1Jac* __restrict__ rowpart) {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    constexpr uint32_t per_row = kRedCols / kRedCW;
    const uint32_t h = t / per_row, cc = t % per_row;
    if (h >= H) return;
    const uint32_t b0 = (h << kRedS) + cc * kRedCW;
    G1Jac acc = g1_identity();
    for (uint32_t k = 0; k < kRedCW; ++k)
        if (b0 + k <= B) acc = g1_add(acc, buckets[b0 + k]);
    rowpart[t] = acc;
}
// out[i] = sum_k part[k * kstride + i * istride], one wavefront per item
__global__ __launch_bounds__(kBlock) void k_fx_red_fold(const G1Jac* __restrict__ part, uint32_t items, uint32_t K, size_t kstride, size_t istride, G1Jac* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63, i = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (i >= items) return;  // wave-uniform
    G1Jac acc = g1_identity();
    for (uint32_t k = lane; k < K; k += 64) acc = g1_add(acc, part[(size_t)k * kstride + (size_t)i * istride]);
    acc = wave_sum_g1(acc, 64);
    if (lane == 0) out[i] = acc;
}

// window tables -> L-form (fq_limb.hip.h): every coordinate times 32, i.e. a product with the Montgomery form of 32; (0, 0) stays (0, 0)
__global__ __launch_bounds__(kBlock) void k_fx_to_lform(G1Affine* __restrict__ pts, size_t count, Fq mont32) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    G1Affine p = ld_aff(pts + i);
    p.x = mul(p.x, mont32);
    p.y = mul(p.y, mont32);
    pts[i] = p;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// precomputation
// ------------------------------------------------------------------------------------------------------------------
static int32_t fx_precompute_into(jolt_ctx* ctx, jolt_srs* srs, uint32_t window_bits, size_t min_terms);
extern "C" int32_t jolt_srs_precompute_windows(jolt_ctx* ctx, jolt_srs* srs, uint32_t window_bits, size_t min_terms) {
    if (!ctx || !srs) return JOLT_ERR_INVALID_ARG;
    if (srs->pre) return JOLT_OK;
    JOLT_TRY(fx_precompute_into(ctx, srs, window_bits, min_terms));
    // the mid table set (srs.hpp): only next to a main set with the wide default windows over >= 2^25 points, whose own crossover lies above the mid set's
    constexpr size_t kMidN = (size_t)1 << 23, kMidMin = (size_t)1 << 19;
    const char* mid = std::getenv("JOLT_MSM_MID");
    if (window_bits == 0 && srs->n >= 4 * kMidN && srs->pre_c >= 23 && srs->pre_min_n > kMidMin && !(mid && std::atoi(mid) == 0) && !srs->mid_tables) {
        jolt_srs* mt = new (std::nothrow) jolt_srs();
        if (!mt) return JOLT_ERR_OOM;
        mt->ctx = ctx;
        mt->pts = srs->pts;  // not owned
        mt->n = kMidN;
        const int32_t s = fx_precompute_into(ctx, mt, 20, kMidMin);
        // the mid set is an optimisation next to main tables that are already built and usable: ANY failure building it (unsupported shape, out of memory, a
        // HIP error from the allocation) leaves the SRS as it was -- the call succeeds exactly as the next one would through the `if (srs->pre)` early exit
        if (s != JOLT_OK) { delete mt; (void)hipGetLastError(); return JOLT_OK; }
        srs->mid_tables = mt;
    }
    return JOLT_OK;
}
static int32_t fx_precompute_into(jolt_ctx* ctx, jolt_srs* srs, uint32_t window_bits, size_t min_terms) {
    if (srs->n == 0) return JOLT_ERR_INVALID_ARG;
    int lg = 0;
    while (((size_t)2 << lg) <= srs->n) lg++;
    // c = 23: the 253 bits of a (negated where needed) scalar are exactly 11 windows and the bucket set stays small (k_fx_digits)
    int c = window_bits ? (int)window_bits : (lg >= 24 ? 23 : std::max(kLoBitsMin + 1, std::min(24, lg - 2)));
    if (c <= kLoBitsMin || c > 26) return JOLT_ERR_UNSUPPORTED;
    const int W = (253 + c - 1) / c;
    // buckets: magnitudes of the signed windows (<= 2^(c-1)) and of the unsigned top window (<= ((r - 1) / 2 >> c (W - 1)) + 1)
    const int shift = c * (W - 1) + 1;  // (r - 1) / 2 >> k = r >> (k + 1), r odd
    if (254 - shift > 31) return JOLT_ERR_UNSUPPORTED;  // the top window must fit a 31-bit magnitude
    auto limb32 = [](int j) -> uint64_t { return j < 8 ? (uint64_t)(uint32_t)FrParams::P[j] : 0ull; };
    const uint64_t top_max = ((limb32(shift >> 5) | (limb32((shift >> 5) + 1) << 32)) >> (shift & 31)) + 1;  // <= 31 significant bits: two limbs cover them
    if (top_max >= ((uint64_t)1 << 31)) return JOLT_ERR_UNSUPPORTED;
    const uint32_t n_bucket_max = (uint32_t)std::max<uint64_t>((uint64_t)1 << (c - 1), top_max);
    if ((size_t)W * srs->n >= ((size_t)1 << 31)) return JOLT_ERR_UNSUPPORTED;  // base index + sign share 32 bits
    G1Affine* pre = nullptr;
    hipError_t e = hipMalloc((void**)&pre, (size_t)W * srs->n * sizeof(G1Affine));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ctx->last_error = std::string("precompute windows: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? JOLT_ERR_OOM : JOLT_ERR_HIP;
    }
    e = hipMemcpyAsync(pre, srs->pts, srs->n * sizeof(G1Affine), hipMemcpyDeviceToDevice, ctx->stream);
    const unsigned grid = (unsigned)((srs->n + kBlock - 1) / kBlock);
    for (int w = 1; w < W && e == hipSuccess; ++w) {
        hipLaunchKernelGGL(k_fx_next_window, dim3(grid), dim3(kBlock), 0, ctx->stream, (const G1Affine*)(pre + (size_t)(w - 1) * srs->n), pre + (size_t)w * srs->n, srs->n, c);
        e = hipGetLastError();
    }
    const bool lform = ctx->msm_fx_lform;
    if (e == hipSuccess && lform) {  // the tables are only ever read by the limb-form bucket sums: store them in L-form
        Fq thirty_two = Fq::zero();
        thirty_two.l[0] = 32;
        const size_t count = (size_t)W * srs->n;
        hipLaunchKernelGGL(k_fx_to_lform, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, pre, count, to_mont(thirty_two));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(pre); ctx->last_error = std::string("precompute windows: ") + hipGetErrorString(e); return JOLT_ERR_HIP; }
    srs->pre = pre;
    srs->pre_lform = lform;
    srs->pre_c = c;
    srs->pre_W = W;
    srs->pre_B = n_bucket_max;
    srs->pre_stride = srs->n;
    // crossover against the per-window method (prefix MSMs over 2^26-point tables)
    srs->pre_min_n = min_terms ? min_terms : std::max<size_t>((size_t)1 << (c - 2), 256);
    return JOLT_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// one MSM on lane `lane`; JOLT_ERR_UNSUPPORTED (W*n >= 2^32, or the runtime refuses the LDS the partition needs) sends the caller back
// to the per-window method
// ------------------------------------------------------------------------------------------------------------------
constexpr size_t kMsmHostEntries = 128;

// pair_shift > 0: TWO MSMs over the same scalars, sum_i s_i P_i and sum_i s_i P_(i + pair_shift): the digits, and with them the whole sort, are the scalars' alone,
// so the second MSM is one more pass of bucket sums and reduction over the same sorted lists with the table pointer moved by pair_shift points (the witness
// commitments at r and -r of a HyperKZG opening are such a pair: hyperkzg.hip).  n + pair_shift must not exceed the tables' point count.
// the digit model of n uniform scalars under c-bit signed windows with an unsigned top window (section 2d); false when the top window does not fit the model
static bool fx_capacity_model(size_t n, int c, int W, FxCapModel* m) {
    const int top_shift = c * (W - 1) + 1;  // the top window's range: ((r - 1) / 2) >> (c (W - 1)), as jolt_host_fx_digits computes it
    if (254 - top_shift > 31 || top_shift < 1) return false;
    auto limb32 = [](int j) -> uint64_t { return j < 8 ? (uint64_t)(uint32_t)FrParams::P[j] : 0ull; };
    const uint64_t top_max = ((limb32(top_shift >> 5) | (limb32((top_shift >> 5) + 1) << 32)) >> (top_shift & 31)) + 1;
    if (top_max == 0 || top_max > 0xFFFFFFFFull) return false;
    m->n = (double)n;
    m->p_signed = 2.0 / (double)((uint64_t)1 << c) * (double)(W - 1);
    m->p_top = 1.0 / (double)top_max;
    m->half = 1u << (c - 1);
    m->top_max = (uint32_t)top_max;
    return true;
}
// the region a segment of 256 buckets gets in the capacity sort of an n-term MSM with window_bits-bit windows: pinned in the CPU suite against the digit histogram of
// uniform scalars (no region may be smaller than what the digits of real scalars put there, and the regions together stay within a few per cent of the entries)
extern "C" int32_t jolt_host_fx_segment_capacity(uint64_t n, uint32_t window_bits, uint32_t segment, uint32_t* capacity) {
    if (!capacity || window_bits < 2 || window_bits > 26) return JOLT_ERR_INVALID_ARG;
    const int c = (int)window_bits, W = (253 + c - 1) / c;
    FxCapModel m;
    if (!fx_capacity_model((size_t)n, c, W, &m)) return JOLT_ERR_UNSUPPORTED;
    *capacity = fx_segment_capacity(m, segment, 1u << fx_lo_bits(c));
    return JOLT_OK;
}

int32_t jolt_internal_msm_fixed_enqueue(jolt_ctx* ctx, const jolt_srs* srs, const Fr* d_scalars, size_t n, int lane, MsmJob* job, size_t pair_shift) {
    const int c = srs->pre_c, W = srs->pre_W;
    const int lo_bits = fx_lo_bits(c);
    const uint32_t kSegBuckets = 1u << lo_bits;
    const uint32_t B = srs->pre_B;                          // largest |digit| (signed windows: 2^(c-1); unsigned top window: top_max)
    const uint32_t nb1 = (B >> lo_bits) + 1;               // the segment of |digit| = B included
    const size_t total = (size_t)W * n;
    if (total >= ((size_t)1 << 32)) return JOLT_ERR_UNSUPPORTED;
    const size_t n_buckets = (size_t)nb1 * kSegBuckets;    // >= B + 1
    // reduction: sum_b b * B_b over buckets 1..B with up to 262144 threads, G buckets each (a serial chain of 2G additions per thread, then one
    // multiplication by the range's offset; B / 96 threads measured faster at 2^26 terms (99.2 -> 96.7 ms) but slower on the short prefix
    // MSMs (2^22: 9.3 -> 10.5 ms): the chains are latency bound at one wavefront per SIMD)
    const uint32_t threads = (uint32_t)std::min<size_t>(std::max<size_t>(B / (size_t)ctx->msm_fx_reduce_div, std::min<size_t>(B, 4096)), 262144);
    const uint32_t nb = (threads + kBlock - 1) / kBlock;
    const uint32_t G = (B + nb * kBlock - 1) / (nb * kBlock);
    // large bucket sets: the row / column reduction (k_fx_red_*); JOLT_FX_REDUCE=0 keeps the running sums for an A/B
    const bool grid_reduce = B >= (1u << 16) && ctx->msm_fx_grid_reduce;
    const uint32_t red_H = (B >> kRedS) + 1, red_chunks = (red_H + kRedRC - 1) / kRedRC, red_per_row = kRedCols / kRedCW;
    const size_t red_points = grid_reduce ? (size_t)red_chunks * kRedCols + (size_t)red_H * red_per_row + kRedCols + red_H + 64 : 0;
    // a bucket holding more than max(kLaneCap, 4x the average) points is summed per 1024-point segment by whole wavefronts: the
    // partial top window of 254-bit scalars (14 bits at c = 24) piles ~n / 2^14 extra points on each of the lowest 2^14 buckets,
    // and small (witness) scalars fill the carry window's bucket 1 -- both take that path, as in the per-window method
    const size_t avg = (total + B - 1) / B;
    // (at most kClasses - 2, so that every light bucket sits in the class of its exact length)
    const uint32_t heavy_threshold = (uint32_t)std::min<size_t>(std::max<size_t>(2 * kLaneCap, 4 * avg), kClasses - 2);
    const uint32_t heavy_cap = (uint32_t)(total / kFxHeavySeg + total / heavy_threshold + 16);
    // split entries and no key array (section 2c): 8-bit segments, the two-pass partition, W <= 12
    static const bool wide_soa = !(std::getenv("JOLT_FX_SOA13") && std::atoi(std::getenv("JOLT_FX_SOA13")) == 0);
    const bool wide = W > kPartPerS;  // 13 windows: the second instantiation of the scalar-fed passes
    const bool soa = ctx->msm_fx_soa && lo_bits == 8 && ctx->msm_fx_partition == 2 && (W <= kPartPerS || (W <= kPartPerWide && wide_soa)) &&
                     ((nb1 + kGroupBins - 1) >> kGroupBits) <= (uint32_t)kPartBins &&
                     (wide ? sizeof(PartSharedN<kPartPerWide, kPartThreadsS>) : sizeof(PartSharedN<kPartPerS, kPartThreadsS>)) + 2048 <= ctx->max_lds_per_block;
    // capacity regions instead of a histogram pass (section 2d): scalars the caller marked as uniform field elements, the split-entry path, long enough to matter
    static const bool capacity_on = !(std::getenv("JOLT_FX_CAPACITY") && std::atoi(std::getenv("JOLT_FX_CAPACITY")) == 0);
    FxCapModel cap_model;
    size_t cap_total = 0;
    bool capacity = capacity_on && soa && ctx->msm_uniform_scalars && n >= ((size_t)1 << 16);
    if (capacity) {
        capacity = fx_capacity_model(n, c, W, &cap_model);
        const uint64_t top_max = cap_model.top_max;
        if (capacity) {
            // the region sizes take a handful of distinct values (whole segments below 2^(c-1), below top_max, beyond both, and the boundary segments): evaluated once each;
            // 8 entries per segment on top, should the device's arithmetic round a size differently (the device's own sizes are what the offsets come from)
            uint64_t key_prev = ~0ull;
            uint32_t cap_prev = 0;
            for (uint32_t sgm = 0; sgm < nb1; ++sgm) {
                const uint64_t lo = (uint64_t)sgm * kSegBuckets, hi = lo + kSegBuckets;
                const uint64_t below_half = lo < cap_model.half ? std::min<uint64_t>(hi, cap_model.half) - lo : 0, below_top = lo < top_max ? std::min<uint64_t>(hi, top_max) - lo : 0;
                const uint64_t key = below_half | (below_top << 20) | ((uint64_t)(lo <= cap_model.half && cap_model.half < hi) << 40);
                if (key != key_prev) { cap_prev = fx_segment_capacity(cap_model, sgm, kSegBuckets); key_prev = key; }
                cap_total += cap_prev + 8;
            }
            capacity = cap_total < ((size_t)1 << 32) - 4096;
        }
    }
    const size_t span = capacity ? std::max(cap_total, total) : total;  // positions the sort's buffers are addressed with
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    // Split-entry path: the key array is first WRITTEN by the segment sort (its output: the per-segment sorted base indices), when the grouped entries (6 B each, read for the
    // last time by the segment partition that precedes the sort on the same stream) are dead -- the 4-byte keys live in the grouped buffer instead of beside it
    // (2.75 GiB less per 2^26-term lane; JOLT_FX_ALIAS_KEYS=0 keeps them apart, for an A/B).
    static const bool alias_on = !(std::getenv("JOLT_FX_ALIAS_KEYS") && std::atoi(std::getenv("JOLT_FX_ALIAS_KEYS")) == 0);
    const bool alias_keys = soa && alias_on;
    size_t o_keys = alias_keys ? 0 : take(span * 4);
    const size_t o_entries = take(soa ? span * 5 + 512 : total * 8), o_hist = take((size_t)nb1 * 4), o_offs = take((size_t)nb1 * 4), o_cur = take((size_t)nb1 * 4),
                 o_info = take(256), o_buckets = take(n_buckets * sizeof(G1Jac)), o_part = take((size_t)nb * sizeof(G1Jac)), o_wsum = take(2 * sizeof(G1Jac)),
                 o_red = take(std::max<size_t>(red_points, 1) * sizeof(G1Jac)),
                 o_bhist = take(n_buckets * 4), o_boffs = take(n_buckets * 4), o_heavy = take((size_t)heavy_cap * 8), o_hcnt = take(256),
                 o_seg = take((size_t)heavy_cap * sizeof(G1Jac)), o_cls = take(kClasses * 4 * 2), o_order = take(n_buckets * 4),
                 o_grouped = take(soa ? span * 6 + 512 : (ctx->msm_fx_partition == 2 ? total * 8 : 256)), o_gcur = take(kPartBins * 4), o_glim = take(kPartBins * 4);
    if (alias_keys) o_keys = o_grouped;
    // a pair's second pass sums into its OWN bucket set and reduces through its own scratch, so that the first pass's reduction (latency bound: chains of additions on
    // a few thousand threads) runs on the auxiliary stream under the second pass's bucket sums instead of between the two
    const bool overlap_reduction = pair_shift != 0 && grid_reduce && ctx->msm_pair_overlap;
    const size_t o_buckets2 = take(overlap_reduction ? n_buckets * sizeof(G1Jac) : 256), o_red2 = take(overlap_reduction ? red_points * sizeof(G1Jac) : 256),
                 o_wsum2 = take(2 * sizeof(G1Jac));
    hipStream_t st = lane == 0 ? ctx->stream : ctx->side[lane - 1];
    // phases: sort (HBM bound) -> bucket sums (multiply-add bound) -> reduction (latency bound).  One stream by default; with JOLT_MSM_CU_SPLIT the
    // sort and the reduction run on the lane's CU-masked "sort" stream and the bucket sums on its "bucket" stream (ctx.hpp), chained by events
    const bool split = ctx->msm_cu_split > 0 && ctx->sort_stream[lane] && ctx->bucket_stream[lane];
    hipStream_t sst = split ? ctx->sort_stream[lane] : st, bst = split ? ctx->bucket_stream[lane] : st;
    if (off > ctx->msm_ws_cap[lane]) {
        if (ctx->msm_ws[lane]) {
            JOLT_HIP_TRY(ctx, hipStreamSynchronize(st));
            if (split) { JOLT_HIP_TRY(ctx, hipStreamSynchronize(sst)); JOLT_HIP_TRY(ctx, hipStreamSynchronize(bst)); }
            JOLT_HIP_TRY(ctx, hipFree(ctx->msm_ws[lane]));
            ctx->msm_ws[lane] = nullptr;
            ctx->msm_ws_cap[lane] = 0;
        }
        JOLT_HIP_TRY(ctx, hipMalloc(&ctx->msm_ws[lane], off));
        ctx->msm_ws_cap[lane] = off;
    }
    if (!ctx->msm_host[lane]) JOLT_HIP_TRY(ctx, hipHostMalloc(&ctx->msm_host[lane], kMsmHostEntries * sizeof(G1Jac), hipHostMallocDefault));
    char* ws = (char*)ctx->msm_ws[lane];
    uint32_t* keys = (uint32_t*)(ws + o_keys);  // after the partition the same buffer holds the per-segment sorted base indices
    uint64_t* entries = (uint64_t*)(ws + o_entries);
    uint32_t *hist1 = (uint32_t*)(ws + o_hist), *offs1 = (uint32_t*)(ws + o_offs), *cur1 = (uint32_t*)(ws + o_cur), *info = (uint32_t*)(ws + o_info);
    G1Jac *buckets = (G1Jac*)(ws + o_buckets), *part = (G1Jac*)(ws + o_part), *wsum = (G1Jac*)(ws + o_wsum), *seg = (G1Jac*)(ws + o_seg);
    uint32_t *hist = (uint32_t*)(ws + o_bhist), *offs = (uint32_t*)(ws + o_boffs), *heavy = (uint32_t*)(ws + o_heavy), *hcnt = (uint32_t*)(ws + o_hcnt);
    uint32_t *class_hist = (uint32_t*)(ws + o_cls), *class_cursor = class_hist + kClasses, *order = (uint32_t*)(ws + o_order);
    uint64_t* grouped = (uint64_t*)(ws + o_grouped);
    uint32_t* group_cursor = (uint32_t*)(ws + o_gcur);
    uint32_t* group_limit = (uint32_t*)(ws + o_glim);
    const uint32_t n_groups = (nb1 + kGroupBins - 1) >> kGroupBits;
    const size_t lds_bytes = (size_t)nb1 * sizeof(uint32_t);
    if (lds_bytes > ctx->max_lds_per_block) return JOLT_ERR_UNSUPPORTED;
    if (!ctx->msm_fx_attr_set) {
        hipError_t a1 = hipFuncSetAttribute((const void*)k_fx_hist<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->max_lds_per_block);
        hipError_t a2 = hipFuncSetAttribute((const void*)k_fx_scatter<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->max_lds_per_block);
        hipError_t a3 = hipFuncSetAttribute((const void*)k_fx_hist<11>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->max_lds_per_block);
        hipError_t a4 = hipFuncSetAttribute((const void*)k_fx_scatter<11>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->max_lds_per_block);
        (void)hipFuncSetAttribute((const void*)k_fx_segment_sort_staged<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(ctx->max_lds_per_block > 12288 ? ctx->max_lds_per_block - 12288 : 0));
        (void)hipGetLastError();
        hipError_t p1 = hipFuncSetAttribute((const void*)k_fx_partition_groups<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PartShared));
        hipError_t p2 = hipFuncSetAttribute((const void*)k_fx_partition_groups<11>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PartShared));
        hipError_t p3 = hipFuncSetAttribute((const void*)k_fx_partition_segments<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PartShared));
        hipError_t p4 = hipFuncSetAttribute((const void*)k_fx_partition_segments<11>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PartShared));
        hipError_t q1 = hipFuncSetAttribute((const void*)k_fx_hist_scalars<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->max_lds_per_block);
        hipError_t q2 = hipFuncSetAttribute((const void*)k_fx_partition_groups_scalars<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PartSharedN<kPartPerS, kPartThreadsS>));
        (void)hipFuncSetAttribute((const void*)k_fx_hist_scalars<8, kPartPerWide>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->max_lds_per_block);
        (void)hipFuncSetAttribute((const void*)k_fx_hist_scalars<8, kPartPerS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->max_lds_per_block);
        (void)hipFuncSetAttribute((const void*)k_fx_partition_groups_scalars<8, kPartPerWide>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PartSharedN<kPartPerWide, kPartThreadsS>));
        (void)hipGetLastError();
        hipError_t q3 = hipFuncSetAttribute((const void*)k_fx_partition_segments_soa<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PartSharedN<kPartPer>));
        (void)hipFuncSetAttribute((const void*)k_fx_segment_sort_staged<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(ctx->max_lds_per_block > 12288 ? ctx->max_lds_per_block - 12288 : 0));
        (void)hipGetLastError();
        if (q1 != hipSuccess || q2 != hipSuccess || q3 != hipSuccess) ctx->msm_fx_soa = false;
        if (p1 != hipSuccess || p2 != hipSuccess || p3 != hipSuccess || p4 != hipSuccess) {
            (void)hipGetLastError();
            ctx->msm_fx_partition = 1;  // the one-pass scatter needs no more LDS than the histogram
        }
        if (a1 != hipSuccess || a2 != hipSuccess || a3 != hipSuccess || a4 != hipSuccess) {
            (void)hipGetLastError();
            if (lds_bytes > 64 * 1024) return JOLT_ERR_UNSUPPORTED;
        }
        ctx->msm_fx_attr_set = true;
    }
    // sort token (ctx.hpp ev_sort): this MSM's HBM-bound phase starts when the previous MSM's has finished, so that it runs under
    // that MSM's bucket sums instead of beside its sort
    if (ctx->msm_stagger && ctx->sort_seq > 0 && ctx->sort_last_lane != lane)
        JOLT_HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_sort[(ctx->sort_seq - 1) % 8], 0));
    if (split) {  // the sort starts after whatever the lane's stream holds (the scalars' producers, the lane's previous MSM)
        JOLT_HIP_TRY(ctx, hipEventRecord(ctx->ev_phase[lane][0], st));
        JOLT_HIP_TRY(ctx, hipStreamWaitEvent(sst, ctx->ev_phase[lane][0], 0));
    }
    const unsigned gn = (unsigned)((n + kBlock - 1) / kBlock);
    const unsigned slices = (unsigned)std::max<size_t>(1, std::min<size_t>((size_t)ctx->num_cus * 2, total / 16384));
    const unsigned hist_grid = (unsigned)std::max<size_t>(1, std::min<size_t>((size_t)ctx->num_cus * 2, n / 4096));
    static const bool plain_hist = !(std::getenv("JOLT_FX_HIST_PLAIN") && std::atoi(std::getenv("JOLT_FX_HIST_PLAIN")) == 0);
    // the exact histogram from the scalars (run_if != nullptr: the fallback behind a capacity sort, a no-op while that sort's overflow flag is clear)
    auto hist_from_scalars = [&](const uint32_t* run_if) {
        if (wide) hipLaunchKernelGGL((k_fx_hist_scalars<8, kPartPerWide>), dim3(hist_grid), dim3(kSortBlock), lds_bytes, sst, d_scalars, n, c, W, nb1, hist1, run_if);
        else if (plain_hist) hipLaunchKernelGGL(k_fx_hist_scalars<8>, dim3(hist_grid), dim3(kSortBlock), lds_bytes, sst, d_scalars, n, c, W, nb1, hist1, run_if);
        else hipLaunchKernelGGL((k_fx_hist_scalars<8, kPartPerS, false>), dim3(hist_grid), dim3(kSortBlock), lds_bytes, sst, d_scalars, n, c, W, nb1, hist1, run_if);
    };
    if (capacity) {  // region sizes take the histogram's place; k_fx_scan below turns them into offsets and cursors like counts
        hipLaunchKernelGGL(k_fx_capacity_regions, dim3((nb1 + kBlock - 1) / kBlock), dim3(kBlock), 0, sst, cap_model, nb1, kSegBuckets, hist1, info);
    } else if (soa) {  // digits straight into the segment histogram: no key array
        JOLT_HIP_TRY(ctx, hipMemsetAsync(hist1, 0, (size_t)nb1 * 4, sst));
        hist_from_scalars(nullptr);
    } else {
        JOLT_HIP_TRY(ctx, hipMemsetAsync(hist1, 0, (size_t)nb1 * 4, sst));
        hipLaunchKernelGGL(k_fx_digits, dim3(gn), dim3(kBlock), 0, sst, d_scalars, n, c, W, keys);
        if (lo_bits == 8) hipLaunchKernelGGL(k_fx_hist<8>, dim3(slices), dim3(kSortBlock), lds_bytes, sst, (const uint32_t*)keys, total, nb1, hist1);
        else hipLaunchKernelGGL(k_fx_hist<11>, dim3(slices), dim3(kSortBlock), lds_bytes, sst, (const uint32_t*)keys, total, nb1, hist1);
    }
    hipLaunchKernelGGL(k_fx_scan, dim3(1), dim3(kSortBlock), 0, sst, (const uint32_t*)hist1, nb1, offs1, cur1, info);
    JOLT_HIP_TRY(ctx, hipGetLastError());
    JOLT_HIP_TRY(ctx, hipMemsetAsync(buckets, 0, n_buckets * sizeof(G1Jac), sst));  // z = 0: identity
    JOLT_HIP_TRY(ctx, hipMemsetAsync(hcnt, 0, 256, sst));
    JOLT_HIP_TRY(ctx, hipMemsetAsync(class_hist, 0, kClasses * 4, sst));
    const bool two_pass = ctx->msm_fx_partition == 2 && n_groups <= (uint32_t)kPartBins && sizeof(PartShared) + 2048 <= ctx->max_lds_per_block;
    const unsigned part_grid = (unsigned)std::max<size_t>(1, std::min<size_t>((size_t)ctx->num_cus * 2, (total + kPartTile - 1) / kPartTile));
    // split-entry arrays of the SoA path inside the two entry buffers
    uint32_t* g_val = (uint32_t*)grouped;
    uint16_t* g_low = (uint16_t*)((char*)grouped + ((span * 4 + 255) & ~(size_t)255));
    uint32_t* s_val = (uint32_t*)entries;
    uint8_t* s_low = (uint8_t*)((char*)entries + ((span * 4 + 255) & ~(size_t)255));
    if (soa) {
        const unsigned grid_s = (unsigned)std::max<size_t>(1, std::min<size_t>((size_t)ctx->num_cus, (n + kPartThreadsS - 1) / kPartThreadsS));
        // the two partition passes: over capacity regions (limits checked, overflow flagged) or over exact offsets (run_if: only when that flag is up)
        auto partition_passes = [&](bool regions, const uint32_t* run_if) {
            const uint32_t* glim = regions ? group_limit : nullptr;
            uint32_t* flag = regions ? info + 2 : nullptr;
            if (wide)
                hipLaunchKernelGGL((k_fx_partition_groups_scalars<8, kPartPerWide>), dim3(grid_s), dim3(kPartThreadsS), sizeof(PartSharedN<kPartPerWide, kPartThreadsS>), sst, d_scalars, n, c, W,
                                   srs->pre_stride, n_groups, group_cursor, g_val, g_low, glim, flag, run_if);
            else
                hipLaunchKernelGGL(k_fx_partition_groups_scalars<8>, dim3(grid_s), dim3(kPartThreadsS), sizeof(PartSharedN<kPartPerS, kPartThreadsS>), sst, d_scalars, n, c, W, srs->pre_stride,
                                   n_groups, group_cursor, g_val, g_low, glim, flag, run_if);
            hipLaunchKernelGGL(k_fx_partition_segments_soa<8>, dim3(part_grid), dim3(kPartThreads), sizeof(PartSharedN<kPartPer>), sst, (const uint32_t*)g_val, (const uint16_t*)g_low,
                               (const uint32_t*)offs1, nb1, (const uint32_t*)info, cur1, s_val, s_low, regions ? (const uint32_t*)group_cursor : nullptr, glim,
                               regions ? (const uint32_t*)hist1 : nullptr, flag, run_if);
        };
        if (capacity) {
            hipLaunchKernelGGL(k_fx_group_regions, dim3(1), dim3(kPartBins), 0, sst, (const uint32_t*)offs1, (const uint32_t*)hist1, nb1, n_groups, group_cursor, group_limit);
            partition_passes(true, nullptr);
            hipLaunchKernelGGL(k_fx_counts_from_cursors, dim3((nb1 + kBlock - 1) / kBlock), dim3(kBlock), 0, sst, (const uint32_t*)offs1, (const uint32_t*)cur1, nb1, hist1, info);
            // the exact sort, enqueued behind it: every kernel returns at its first instruction unless a region overflowed
            const uint32_t* flag = info + 2;
            hipLaunchKernelGGL(k_fx_clear_if, dim3((nb1 + kBlock - 1) / kBlock), dim3(kBlock), 0, sst, hist1, nb1, flag);
            hist_from_scalars(flag);
            hipLaunchKernelGGL(k_fx_scan, dim3(1), dim3(kSortBlock), 0, sst, (const uint32_t*)hist1, nb1, offs1, cur1, info, flag);
            hipLaunchKernelGGL(k_fx_group_cursors, dim3(1), dim3(kPartBins), 0, sst, (const uint32_t*)offs1, n_groups, group_cursor, flag);
            partition_passes(false, flag);
        } else {
            hipLaunchKernelGGL(k_fx_group_cursors, dim3(1), dim3(kPartBins), 0, sst, (const uint32_t*)offs1, n_groups, group_cursor);
            partition_passes(false, nullptr);
        }
    } else if (two_pass) {
        hipLaunchKernelGGL(k_fx_group_cursors, dim3(1), dim3(kPartBins), 0, sst, (const uint32_t*)offs1, n_groups, group_cursor);
        if (lo_bits == 8) {
            hipLaunchKernelGGL(k_fx_partition_groups<8>, dim3(part_grid), dim3(kPartThreads), sizeof(PartShared), sst, (const uint32_t*)keys, total, n, srs->pre_stride, n_groups,
                               group_cursor, grouped);
            hipLaunchKernelGGL(k_fx_partition_segments<8>, dim3(part_grid), dim3(kPartThreads), sizeof(PartShared), sst, (const uint64_t*)grouped, (const uint32_t*)offs1, nb1,
                               (const uint32_t*)info, cur1, entries);
        } else {
            hipLaunchKernelGGL(k_fx_partition_groups<11>, dim3(part_grid), dim3(kPartThreads), sizeof(PartShared), sst, (const uint32_t*)keys, total, n, srs->pre_stride, n_groups,
                               group_cursor, grouped);
            hipLaunchKernelGGL(k_fx_partition_segments<11>, dim3(part_grid), dim3(kPartThreads), sizeof(PartShared), sst, (const uint64_t*)grouped, (const uint32_t*)offs1, nb1,
                               (const uint32_t*)info, cur1, entries);
        }
    } else if (lo_bits == 8) {
        hipLaunchKernelGGL(k_fx_scatter<8>, dim3(slices), dim3(kSortBlock), lds_bytes, sst, (const uint32_t*)keys, total, n, srs->pre_stride, nb1, cur1, entries);
    } else {
        hipLaunchKernelGGL(k_fx_scatter<11>, dim3(slices), dim3(kSortBlock), lds_bytes, sst, (const uint32_t*)keys, total, n, srs->pre_stride, nb1, cur1, entries);
    }
    if (lo_bits == 8) {
        // LDS for the staged segment: twice the average segment (uniform digits spread by ~1 %), at most what a CU has
        const size_t lds_max = ctx->max_lds_per_block > 12288 ? ctx->max_lds_per_block - 12288 : 0;  // cnt / cur / cls / wave sums live next to it
        const size_t want = (2 * (total / nb1) + 2048) * 4;
        const size_t stage_bytes = ctx->msm_fx_stage ? std::min(lds_max, want) : 0;
        if (soa)
            hipLaunchKernelGGL((k_fx_segment_sort_staged<8, true>), dim3(nb1), dim3(kSegThreads), stage_bytes, sst, (const uint32_t*)hist1, (const uint32_t*)offs1, (const uint64_t*)nullptr,
                               (const uint32_t*)s_val, (const uint8_t*)s_low, keys, hist, offs, heavy_threshold, heavy, hcnt, heavy_cap, class_hist, (uint32_t)(stage_bytes / 4));
        else
            hipLaunchKernelGGL((k_fx_segment_sort_staged<8, false>), dim3(nb1), dim3(kSegThreads), stage_bytes, sst, (const uint32_t*)hist1, (const uint32_t*)offs1, (const uint64_t*)entries,
                               (const uint32_t*)nullptr, (const uint8_t*)nullptr, keys, hist, offs, heavy_threshold, heavy, hcnt, heavy_cap, class_hist, (uint32_t)(stage_bytes / 4));
    }
    else
        hipLaunchKernelGGL(k_fx_segment_sort<11>, dim3(nb1), dim3(kBlock), 0, sst, (const uint32_t*)hist1, (const uint32_t*)offs1, (const uint64_t*)entries, keys, hist, offs,
                           heavy_threshold, heavy, hcnt, heavy_cap, class_hist);
    // bucket sums: the kernels of the per-window method with ONE window of B buckets (bases = the window tables)
    const unsigned gh = std::min<uint32_t>((heavy_cap + kBlock - 1) / kBlock, (uint32_t)ctx->num_cus * 64);  // grid-stride over the heavy list (its length lives on the device)
    hipLaunchKernelGGL(k_fx_order_scan, dim3(1), dim3(kClasses), 0, sst, (const uint32_t*)class_hist, class_cursor);
    hipLaunchKernelGGL(k_fx_order, dim3((unsigned)((n_buckets + kBlock * kOrderPer - 1) / (kBlock * kOrderPer))), dim3(kBlock), 0, sst, (const uint32_t*)hist, (uint32_t)n_buckets,
                       heavy_threshold, class_cursor, order);
    if (ctx->msm_stagger) {
        JOLT_HIP_TRY(ctx, hipEventRecord(ctx->ev_sort[ctx->sort_seq % 8], sst));
        ctx->sort_seq++;
        ctx->sort_last_lane = lane;
    }
    if (split) {
        JOLT_HIP_TRY(ctx, hipEventRecord(ctx->ev_phase[lane][1], sst));
        JOLT_HIP_TRY(ctx, hipStreamWaitEvent(bst, ctx->ev_phase[lane][1], 0));
    }
    LformConsts lc;
    {
        Fq thirty_two = Fq::zero();
        thirty_two.l[0] = 32;
        lc.one_l = to_mont(thirty_two);
        lc.r256 = Fq::one();
    }
    const unsigned bucket_grid = (unsigned)((n_buckets + kBlock - 1) / kBlock);
    const int per_result = grid_reduce ? 2 : 1;  // partial sums a result leaves in msm_host for the collect step's Horner
    if ((size_t)per_result * (pair_shift ? 2 : 1) > kMsmHostEntries) return JOLT_ERR_UNSUPPORTED;
    // bucket sums over the sorted lists against the tables at `bases` into `buckets` (on the bucket stream) ...
    auto bucket_sums = [&](const G1Affine* bases, G1Jac* buckets) -> int32_t {
        const bool profile = ctx->fx_profile && ctx->ev_fx[0] && ctx->ev_fx[1];
        if (profile) JOLT_HIP_TRY(ctx, hipEventRecord(ctx->ev_fx[0], bst));
        static const bool stage_idx = !(std::getenv("JOLT_FX_STAGE_IDX") && std::atoi(std::getenv("JOLT_FX_STAGE_IDX")) == 0);
        if (srs->pre_lform) {
            if (stage_idx)
                hipLaunchKernelGGL(k_fx_buckets_ordered_staged, dim3(bucket_grid), dim3(kBlock), 0, bst, (const uint32_t*)order, (uint32_t)n_buckets, (const uint32_t*)hist,
                                   (const uint32_t*)offs, (const uint32_t*)keys, bases, heavy_threshold, buckets, lc);
            else
                hipLaunchKernelGGL(k_fx_buckets_ordered<true>, dim3(bucket_grid), dim3(kBlock), 0, bst, (const uint32_t*)order, (uint32_t)n_buckets, (const uint32_t*)hist,
                                   (const uint32_t*)offs, (const uint32_t*)keys, bases, heavy_threshold, buckets, lc);
            if (profile) {  // the LAST profiled launch is the one reported
                JOLT_HIP_TRY(ctx, hipEventRecord(ctx->ev_fx[1], bst));
                ctx->fx_profile_info = info;
                ctx->fx_profile_valid = true;
            }
            if (stage_idx)
                hipLaunchKernelGGL(k_fx_heavy_segments_staged, dim3(gh), dim3(kBlock), 0, bst, (const uint32_t*)heavy, (const uint32_t*)hcnt, heavy_cap, (const uint32_t*)hist,
                                   (const uint32_t*)offs, (const uint32_t*)keys, bases, seg, lc);
            else
                hipLaunchKernelGGL(k_fx_heavy_segments<true>, dim3(gh), dim3(kBlock), 0, bst, (const uint32_t*)heavy, (const uint32_t*)hcnt, heavy_cap, (const uint32_t*)hist,
                                   (const uint32_t*)offs, (const uint32_t*)keys, bases, seg, lc);
        } else {
            hipLaunchKernelGGL(k_fx_buckets_ordered<false>, dim3(bucket_grid), dim3(kBlock), 0, bst, (const uint32_t*)order, (uint32_t)n_buckets, (const uint32_t*)hist,
                               (const uint32_t*)offs, (const uint32_t*)keys, bases, heavy_threshold, buckets, lc);
            hipLaunchKernelGGL(k_fx_heavy_segments<false>, dim3(gh), dim3(kBlock), 0, bst, (const uint32_t*)heavy, (const uint32_t*)hcnt, heavy_cap, (const uint32_t*)hist, (const uint32_t*)offs,
                               (const uint32_t*)keys, bases, seg, lc);
        }
        hipLaunchKernelGGL(k_fx_heavy_combine, dim3(std::min<uint32_t>(gh, 2048)), dim3(kBlock), 0, bst, (const uint32_t*)heavy, (const uint32_t*)hcnt, heavy_cap, (const uint32_t*)hist,
                           (const G1Jac*)seg, buckets);
        JOLT_HIP_TRY(ctx, hipGetLastError());
        return JOLT_OK;
    };
    // ... and their reduction on `bst` (any stream ordered behind the sums), through the scratch at o_red / wsum; the result's partial sums go to msm_host[lane][slot ..]
    auto reduction = [&](const G1Jac* buckets, size_t o_red, G1Jac* wsum, int slot, hipStream_t bst) -> int32_t {
        if (grid_reduce) {
            G1Jac* colpart = (G1Jac*)(ws + o_red);
            G1Jac* rowpart = colpart + (size_t)red_chunks * kRedCols;
            G1Jac* cols = rowpart + (size_t)red_H * red_per_row;  // C_l, l < 2^S
            G1Jac* rows = cols + kRedCols;                         // R_h, h < H
            G1Jac* small_part = rows + red_H;                      // block partials of the two small reductions
            hipLaunchKernelGGL(k_fx_red_cols, dim3(kRedCols / kBlock, red_chunks), dim3(kBlock), 0, bst, (const G1Jac*)buckets, B, red_H, colpart);
            hipLaunchKernelGGL(k_fx_red_rows, dim3((red_H * red_per_row + kBlock - 1) / kBlock), dim3(kBlock), 0, bst, (const G1Jac*)buckets, B, red_H, rowpart);
            hipLaunchKernelGGL(k_fx_red_fold, dim3(kRedCols * 64 / kBlock), dim3(kBlock), 0, bst, (const G1Jac*)colpart, kRedCols, red_chunks, (size_t)kRedCols, (size_t)1, cols);
            hipLaunchKernelGGL(k_fx_red_fold, dim3((red_H * 64 + kBlock - 1) / kBlock), dim3(kBlock), 0, bst, (const G1Jac*)rowpart, red_H, red_per_row, (size_t)1, (size_t)red_per_row, rows);
            // sum_l l C_l (weights 1 .. 2^S - 1) and sum_h h R_h (weights 1 .. H - 1): index = weight, entry 0 unused -- the layout k_msm_window_reduce reads
            const uint32_t g_small = 8, nb_c = (kRedCols / g_small + kBlock - 1) / kBlock, nb_r = (red_H / g_small + kBlock) / kBlock;
            hipLaunchKernelGGL(k_msm_window_reduce, dim3(nb_c, 1), dim3(kBlock), 0, bst, (const G1Jac*)cols, kRedCols - 1, g_small, small_part);
            hipLaunchKernelGGL(k_msm_window_combine, dim3(1), dim3(64), 0, bst, (const G1Jac*)small_part, nb_c, wsum);
            if (red_H > 1) {
                hipLaunchKernelGGL(k_msm_window_reduce, dim3(nb_r, 1), dim3(kBlock), 0, bst, (const G1Jac*)rows, red_H - 1, g_small, small_part + 32);
                hipLaunchKernelGGL(k_msm_window_combine, dim3(1), dim3(64), 0, bst, (const G1Jac*)(small_part + 32), nb_r, wsum + 1);
            } else {
                JOLT_HIP_TRY(ctx, hipMemsetAsync(wsum + 1, 0, sizeof(G1Jac), bst));
            }
        } else {
            hipLaunchKernelGGL(k_msm_window_reduce, dim3(nb, 1), dim3(kBlock), 0, bst, (const G1Jac*)buckets, B, G, part);
            hipLaunchKernelGGL(k_msm_window_combine, dim3(1), dim3(64), 0, bst, (const G1Jac*)part, nb, wsum);
        }
        JOLT_HIP_TRY(ctx, hipGetLastError());
        JOLT_HIP_TRY(ctx, hipMemcpyAsync((G1Jac*)ctx->msm_host[lane] + slot, wsum, (size_t)per_result * sizeof(G1Jac), hipMemcpyDeviceToHost, bst));
        return JOLT_OK;
    };
    JOLT_TRY(bucket_sums((const G1Affine*)srs->pre, buckets));
    if (!pair_shift) {
        JOLT_TRY(reduction(buckets, o_red, wsum, 0, bst));
    } else if (overlap_reduction) {
        if (!ctx->msm_aux_stream) JOLT_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->msm_aux_stream, hipStreamNonBlocking));
        if (!ctx->ev_aux[lane][0]) for (hipEvent_t& ev : ctx->ev_aux[lane]) JOLT_HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        G1Jac* buckets2 = (G1Jac*)(ws + o_buckets2);
        JOLT_HIP_TRY(ctx, hipEventRecord(ctx->ev_aux[lane][0], bst));  // the first pass's buckets are complete
        JOLT_HIP_TRY(ctx, hipStreamWaitEvent(ctx->msm_aux_stream, ctx->ev_aux[lane][0], 0));
        JOLT_TRY(reduction(buckets, o_red, wsum, 0, ctx->msm_aux_stream));
        JOLT_HIP_TRY(ctx, hipEventRecord(ctx->ev_aux[lane][1], ctx->msm_aux_stream));
        JOLT_HIP_TRY(ctx, hipMemsetAsync(buckets2, 0, n_buckets * sizeof(G1Jac), bst));
        JOLT_TRY(bucket_sums((const G1Affine*)srs->pre + pair_shift, buckets2));
        JOLT_TRY(reduction(buckets2, o_red2, (G1Jac*)(ws + o_wsum2), per_result, bst));
        JOLT_HIP_TRY(ctx, hipStreamWaitEvent(bst, ctx->ev_aux[lane][1], 0));  // the lane's stream covers both results (jolt_internal_msm_collect waits for it alone)
    } else {
        JOLT_TRY(reduction(buckets, o_red, wsum, 0, bst));
        JOLT_HIP_TRY(ctx, hipMemsetAsync(buckets, 0, n_buckets * sizeof(G1Jac), bst));  // empty buckets rely on the identity the first pass overwrote nowhere; light / heavy ones are rewritten
        JOLT_TRY(bucket_sums((const G1Affine*)srs->pre + pair_shift, buckets));
        JOLT_TRY(reduction(buckets, o_red, wsum, per_result, bst));
    }
    if (split) { JOLT_HIP_TRY(ctx, hipEventRecord(ctx->ev_phase[lane][3], bst)); JOLT_HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_phase[lane][3], 0)); }
    job->n = n;
    job->lane = lane;
    job->c = grid_reduce ? kRedS : 0;  // the collect step's Horner: 2^S * (row-weighted sum) + (column-weighted sum); or a single "window"
    job->W = per_result;
    job->nb = nb;
    job->results = pair_shift ? 2 : 1;
    return JOLT_OK;
}

// The digit recoding of the fixed-base MSM for ONE scalar (Montgomery form), built for the host: keys_out[w] = |digit_w| | sign << 31 for the
// W = ceil(253 / c) windows, *buckets_out = the bucket count a table set with this c uses (largest possible |digit|).
extern "C" int32_t jolt_host_fx_digits(const jolt_fr_t* scalar, uint32_t window_bits, uint32_t* keys_out, uint32_t* n_windows_out, uint32_t* buckets_out) {
    if (!scalar || !keys_out || window_bits < 2 || window_bits > 26) return JOLT_ERR_INVALID_ARG;
    const int c = (int)window_bits, W = (253 + c - 1) / c;
    const int shift = c * (W - 1) + 1;
    if (254 - shift > 31) return JOLT_ERR_UNSUPPORTED;
    auto limb32 = [](int j) -> uint64_t { return j < 8 ? (uint64_t)(uint32_t)FrParams::P[j] : 0ull; };
    const uint64_t top_max = ((limb32(shift >> 5) | (limb32((shift >> 5) + 1) << 32)) >> (shift & 31)) + 1;
    Fr s = fr_from_abi(scalar);
    if (!fr_is_canonical(s)) return JOLT_ERR_INVALID_ARG;
    fx_digits_of(from_mont(s), c, W, keys_out, 1);
    if (n_windows_out) *n_windows_out = (uint32_t)W;
    if (buckets_out) *buckets_out = (uint32_t)std::max<uint64_t>((uint64_t)1 << (c - 1), top_max);
    return JOLT_OK;
}


// ---- profile of the dominant kernel (bench.py roofline_msm) ------------------------------------------------------------------------------------------------
// enable: every later fixed-base MSM of the context brackets its k_fx_buckets_ordered launch with HIP events on the launch's own stream.  _last: duration of the last
// bracketed launch and the mixed additions it performed (the non-zero signed digits of its scalars: light and heavy buckets together; heavy lists are a few
// thousandths for uniform scalars).  Call after the MSM was collected (the events must have completed).
extern "C" int32_t jolt_msm_profile_buckets(jolt_ctx* ctx, int32_t enable) {
    if (!ctx) return JOLT_ERR_INVALID_ARG;
    if (enable && !ctx->ev_fx[0]) {
        JOLT_HIP_TRY(ctx, hipEventCreate(&ctx->ev_fx[0]));
        JOLT_HIP_TRY(ctx, hipEventCreate(&ctx->ev_fx[1]));
    }
    ctx->fx_profile = enable != 0;
    ctx->fx_profile_valid = false;
    return JOLT_OK;
}
extern "C" int32_t jolt_msm_profile_buckets_last(jolt_ctx* ctx, float* ms, uint64_t* additions) {
    if (!ctx || !ms || !additions) return JOLT_ERR_INVALID_ARG;
    if (!ctx->fx_profile_valid) { ctx->last_error = "no profiled fixed-base MSM since jolt_msm_profile_buckets"; return JOLT_ERR_INVALID_ARG; }
    JOLT_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_fx[1]));
    JOLT_HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev_fx[0], ctx->ev_fx[1]));
    uint32_t info[4] = {0, 0, 0, 0};
    JOLT_HIP_TRY(ctx, hipMemcpy(info, ctx->fx_profile_info, sizeof(info), hipMemcpyDeviceToHost));
    *additions = info[3] ? info[3] : info[1];  // a capacity sort counts its entries in info[3]; the exact scan leaves the total in info[1]
    return JOLT_OK;
}


// ---- the bound roofline_msm divides by, measured in the process that divides by it ---------------------------------------------------------------------------
// Chip-wide v_mad_u64_u32 issue rate: 8 independent 64-bit accumulators per lane, `iters` x 8 multiply-adds per thread, no memory traffic (one store per thread at the
// end), 8 workgroups of 256 per CU -- the loop of jolt_amd/csrc/tools/microbench.hip's k_mad.  Launches are repeated until >= target_ms of kernel time have been timed
// with HIP events on the context's stream; the best launch's rate is reported (the bound is what the part can do, not its average under a cold clock).
namespace {
__global__ __launch_bounds__(256) void k_mad_peak(uint64_t* out, uint32_t a0, uint32_t b0, int iters) {
    uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
    uint64_t acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = (uint64_t)a * (uint32_t)(b + k) + acc[k];
        a += (uint32_t)acc[0];
    }
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s ^= acc[k];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
}  // namespace
extern "C" int32_t jolt_ctx_measure_mad_peak(jolt_ctx* ctx, float target_ms, double* mads_per_s, float* timed_ms, uint32_t* launches) {
    if (!ctx || !mads_per_s || !(target_ms > 0.f) || target_ms > 5000.f) return JOLT_ERR_INVALID_ARG;
    JOLT_TRY(jolt_internal_engine_quiesce(ctx));
    const int blocks = (int)ctx->num_cus * 8, iters = 4096;
    uint64_t* out = nullptr;
    JOLT_TRY(jolt_internal_dev_alloc(ctx, (size_t)blocks * 256 * sizeof(uint64_t), (void**)&out));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    double best = 0.0;
    float total = 0.f;
    uint32_t count = 0;
    const double mads = (double)blocks * 256.0 * (double)iters * 8.0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_mad_peak, dim3(blocks), dim3(256), 0, ctx->stream, out, 12345u, 67891u, iters);  // warm: code object load, clocks
        e = hipStreamSynchronize(ctx->stream);
    }
    while (e == hipSuccess && total < target_ms && count < 10000) {
        e = hipEventRecord(e0, ctx->stream);
        hipLaunchKernelGGL(k_mad_peak, dim3(blocks), dim3(256), 0, ctx->stream, out, 12345u + count, 67891u, iters);
        if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess || !(ms > 0.f)) break;
        total += ms;
        ++count;
        best = std::max(best, mads / ((double)ms * 1e-3));
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    jolt_internal_dev_free(ctx, out);
    if (e != hipSuccess || count == 0) {
        (void)hipGetLastError();
        ctx->last_error = std::string("mad peak measurement: ") + hipGetErrorString(e);
        return JOLT_ERR_HIP;
    }
    *mads_per_s = best;
    if (timed_ms) *timed_ms = total;
    if (launches) *launches = count;
    return JOLT_OK;
}
