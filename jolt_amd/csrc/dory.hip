// jolt_amd/csrc/dory.hip -- Dory tier-1 (G1) streaming commitments on gfx950: SURVEY.md section 8(f) row 2.
//
// Replaces the row-commitment work of DoryScheme's streaming interface (crates/jolt-dory/src/streaming.rs):
//   feed_u64 / feed_i128 / feed_i128_rows_with (:115-205)  -> jolt_dory_commit_rows: one G1 MSM per row_width window of
//       small integers, every row over the same first row_width bases (ark msm_u64 / msm_i128: sum_j v_j * G_j, negative
//       values as v = -|v|);
//   process_one_hot_chunk(s_with) (:230-275) -> one_hot_chunk_commitments (:366-419) -> jolt_dory_commit_onehot:
//       commitment[k] = sum of the bases of the columns whose hot row is k (no scalar multiplications at all).
// Tier 2 (pairings into GT, commit_rows_tier_2) is outside SURVEY.md section 8.
//
// Both are the bucket method of msm.hip with more bucket sets: a "window" is (row, digit window) for the integer rows and
// one chunk for the one-hot columns, so a whole batch is ONE pass of digits -> scan -> scatter -> bucket sums, followed by
// one workgroup per row that folds its windows (running-sum reduction + Horner).  Only the windows the batch's largest
// magnitude needs are processed (an OR-reduction over the batch decides).  Integer VALU work, no MFMA.
#include <algorithm>

#include "ctx.hpp"
#include "msm_kernels.hip.h"
#include "onehot.hpp"
#include "srs.hpp"

using namespace jolt;
using namespace jolt::msmk;

#include "ints.hpp"

namespace {

constexpr int kMaxRowWindows = 48;  // LDS window sums of the row fold: 129 bits / c with c >= 3

__host__ __device__ inline size_t int_bytes(int kind) { return kind == JOLT_INT_I128 ? 16 : 8; }

// |v| as a 128-bit magnitude in four u32 (LE) and the sign
template <int KIND>
__device__ __forceinline__ void load_magnitude(const void* __restrict__ data, size_t i, uint32_t m[4], uint32_t& negative) {
    uint64_t lo, hi;
    if (KIND == JOLT_INT_I128) {
        const uint64_t* p = reinterpret_cast<const uint64_t*>(data) + 2 * i;
        lo = p[0];
        hi = p[1];
        negative = (uint32_t)(hi >> 63);
        if (negative) {  // two's complement negate; |i128::MIN| = 2^127 still fits
            lo = ~lo + 1;
            hi = ~hi + (lo == 0 ? 1 : 0);
        }
    } else {
        lo = reinterpret_cast<const uint64_t*>(data)[i];
        hi = 0;
        negative = KIND == JOLT_INT_I64 ? (uint32_t)(lo >> 63) : 0u;
        if (negative) lo = ~lo + 1;
    }
    m[0] = (uint32_t)lo; m[1] = (uint32_t)(lo >> 32); m[2] = (uint32_t)hi; m[3] = (uint32_t)(hi >> 32);
}

// OR of all magnitudes of the batch -> its bit length bounds the windows worth processing
template <int KIND>
__global__ __launch_bounds__(kBlock) void k_ints_or(const void* __restrict__ data, size_t n, uint32_t* __restrict__ out4) {
    uint32_t acc[4] = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        uint32_t m[4], negf;
        load_magnitude<KIND>(data, i, m, negf);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] |= m[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        for (int off = 32; off >= 1; off >>= 1) acc[k] |= (uint32_t)__shfl_xor((int)acc[k], off, 64);
        if ((threadIdx.x & 63) == 0 && acc[k]) atomicOr(&out4[k], acc[k]);
    }
}

// Signed c-bit digits of the values of rows [row0, row0 + rows): window (r, w) = r * W + w, keys[window * width + col].
template <int KIND>
__global__ __launch_bounds__(kBlock) void k_rows_digits(const void* __restrict__ data, size_t first, size_t n, uint32_t width_log, int c, int W,
                                                       uint32_t* __restrict__ keys, uint32_t* __restrict__ hist) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < n;
    uint32_t m[4] = {0, 0, 0, 0}, negv = 0;
    if (live) load_magnitude<KIND>(data, first + i, m, negv);
    const size_t row = i >> width_log, col = i & (((size_t)1 << width_log) - 1);
    const uint32_t B = 1u << (c - 1);
    uint32_t carry = 0;
    for (int w = 0; w < W; ++w) {
        int bit = w * c;
        uint32_t raw = 0;
        if (bit < 128) {
            int limb = bit >> 5, off = bit & 31;
            uint64_t two = (uint64_t)m[limb] | (limb + 1 < 4 ? (uint64_t)m[limb + 1] << 32 : 0ull);
            raw = (uint32_t)(two >> off) & ((1u << c) - 1);
        }
        raw += carry;
        uint32_t mag, negf;
        if (raw > B) { mag = (1u << c) - raw; negf = 1; carry = 1; }
        else { mag = raw; negf = 0; carry = 0; }
        const size_t win = row * (size_t)W + w;
        if (live) keys[(win << width_log) + col] = mag | ((negf ^ negv) << 31);
        const uint32_t slot = (uint32_t)(win * (B + 1) + mag);  // < 2^32 (checked by the host); also the aggregation key, rows may share a wavefront
        WaveAgg ag = wave_aggregate(slot, live && mag != 0);
        if (ag.do_atomic) atomicAdd(&hist[slot], ag.count);
    }
}

// The signed digit of window w of a 128-bit magnitude (the carry chain is replayed from window 0: W <= 48 cheap steps)
__device__ __forceinline__ void digit_at(const uint32_t m[4], int c, int w, uint32_t& mag, uint32_t& negf) {
    const uint32_t B = 1u << (c - 1);
    uint32_t carry = 0;
    mag = 0;
    negf = 0;
    for (int v = 0; v <= w; ++v) {
        int bit = v * c;
        uint32_t raw = 0;
        if (bit < 128) {
            int limb = bit >> 5, off = bit & 31;
            uint64_t two = (uint64_t)m[limb] | (limb + 1 < 4 ? (uint64_t)m[limb + 1] << 32 : 0ull);
            raw = (uint32_t)(two >> off) & ((1u << c) - 1);
        }
        raw += carry;
        if (raw > B) { mag = (1u << c) - raw; negf = 1; carry = 1; }
        else { mag = raw; negf = 0; carry = 0; }
    }
}

// Counting sort of one row's digits, window after window, entirely in LDS: blockIdx.x = row.  Replaces digits -> scan -> scatter
// with their two global atomics per key (device-scope atomics cost ~90 ps each on this part: 1.7 of the 11 ms of a
// 2048 x 2048 u64 batch); writes hist / offsets / sorted for the row's W windows and lists the over-full buckets.
template <int KIND>
__global__ __launch_bounds__(kBlock) void k_rows_sort_lds(const void* __restrict__ data, size_t first, uint32_t width_log, int c, int W,
                                                         uint32_t* __restrict__ hist, uint32_t* __restrict__ offsets, uint32_t* __restrict__ sorted,
                                                         uint32_t heavy_threshold, uint32_t* __restrict__ heavy_list, uint32_t* __restrict__ heavy_count,
                                                         uint32_t heavy_cap) {
    extern __shared__ uint32_t rows_sh[];  // B + 1 counters, then kBlock scan cells
    const uint32_t B = 1u << (c - 1), width = 1u << width_log;
    uint32_t* cnt = rows_sh;
    uint32_t* cell = rows_sh + B + 1;
    const size_t row = blockIdx.x;
    const size_t base = first + (row << width_log);
    const uint32_t per = (B + kBlock) / kBlock;  // bins per thread in the scan, bins 0..B
    for (int w = 0; w < W; ++w) {
        for (uint32_t b = threadIdx.x; b <= B; b += kBlock) cnt[b] = 0;
        __syncthreads();
        for (uint32_t col0 = 0; col0 < width; col0 += kBlock) {  // whole wavefronts walk the loop together (ballots inside)
            const uint32_t col = col0 + threadIdx.x;
            uint32_t mag = 0, negf = 0;
            if (col < width) {
                uint32_t m[4], negv;
                load_magnitude<KIND>(data, base + col, m, negv);
                digit_at(m, c, w, mag, negf);
            }
            WaveAgg ag = wave_aggregate(mag, mag != 0);
            if (ag.do_atomic) atomicAdd(&cnt[mag], ag.count);
        }
        __syncthreads();
        // exclusive scan of the counters (bin 0 stays empty) -> global hist / offsets; the counters become the scatter cursors
        const uint32_t lo = threadIdx.x * per, hi = min(lo + per, B + 1);
        uint32_t local = 0;
        for (uint32_t k = lo; k < hi; ++k) local += cnt[k];
        cell[threadIdx.x] = local;
        __syncthreads();
        for (int off = 1; off < kBlock; off <<= 1) {
            uint32_t v = (int)threadIdx.x >= off ? cell[threadIdx.x - off] : 0;
            __syncthreads();
            cell[threadIdx.x] += v;
            __syncthreads();
        }
        uint32_t run = cell[threadIdx.x] - local;
        const size_t win = row * (size_t)W + w;
        for (uint32_t k = lo; k < hi; ++k) {
            const uint32_t n_k = cnt[k];
            const uint32_t slot = (uint32_t)(win * (B + 1) + k);
            hist[slot] = n_k;
            offsets[slot] = run;
            cnt[k] = run;
            if (n_k > heavy_threshold) {
                uint32_t nseg = (n_k + kHeavySeg - 1) / kHeavySeg;
                uint32_t f0 = atomicAdd(heavy_count, nseg);
                for (uint32_t sgi = 0; sgi < nseg && f0 + sgi < heavy_cap; ++sgi) {
                    heavy_list[2 * (f0 + sgi)] = slot;
                    heavy_list[2 * (f0 + sgi) + 1] = sgi;
                }
            }
            run += n_k;
        }
        __syncthreads();
        for (uint32_t col0 = 0; col0 < width; col0 += kBlock) {
            const uint32_t col = col0 + threadIdx.x;
            uint32_t mag = 0, negf = 0, negv = 0;
            if (col < width) {
                uint32_t m[4];
                load_magnitude<KIND>(data, base + col, m, negv);
                digit_at(m, c, w, mag, negf);
            }
            WaveAgg ag = wave_aggregate(mag, mag != 0);
            uint32_t f0 = 0;
            if (ag.do_atomic) f0 = atomicAdd(&cnt[mag], ag.count);
            uint32_t pos = (uint32_t)__shfl((int)f0, ag.src, 64) + ag.rank;
            if (mag) sorted[(win << width_log) + pos] = col | ((negf ^ negv) << 31);
        }
        __syncthreads();
    }
}

// One workgroup per row: each wavefront folds whole windows (running sums over its lanes' bucket ranges, butterfly over the
// lanes), then one lane runs the Horner recombination acc = 2^c acc + S_w over the row's W window sums.
__global__ __launch_bounds__(kBlock) void k_rows_fold(const G1Jac* __restrict__ buckets, uint32_t B, int c, int W, G1Jac* __restrict__ out) {
    __shared__ G1Jac sm[kMaxRowWindows];
    const size_t row = blockIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t G = (B + 63) / 64;
    for (int w = (int)wave; w < W; w += kBlock / 64) {
        const G1Jac* bk = buckets + (row * (size_t)W + w) * (B + 1);
        uint32_t lo = lane * G + 1, hi = min(lo + G - 1, B);
        G1Jac contrib = g1_identity();
        if (lo <= B) {
            G1Jac running = g1_identity(), acc = g1_identity();
            for (uint32_t b = hi; b >= lo; --b) {
                running = g1_add(running, bk[b]);
                acc = g1_add(acc, running);
            }
            contrib = g1_add(acc, g1_mul_small(running, lo - 1));  // sum_b b B_b over the lane's range
        }
        contrib = wave_sum_g1(contrib, 64);
        if (lane == 0) sm[w] = contrib;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        G1Jac acc = g1_identity();
        for (int w = W - 1; w >= 0; --w) {
            for (int k = 0; k < c; ++k) acc = g1_double(acc);
            acc = g1_add(acc, sm[w]);
        }
        out[row] = g1_is_identity(acc) ? g1_identity() : acc;
    }
}

// The same fold for small bucket sets (B <= 128, row widths up to 4096): sum_w 2^(cw) sum_b b B_(w,b) = sum_b b H_b with
// H_b = sum_w 2^(cw) B_(w,b).  Lane b runs the Horner recombination over the windows for ITS bucket -- all lanes in
// lockstep, so the W*c doublings cost one wavefront pass instead of one idle-lane pass per row on top of W butterfly
// reductions -- then a single weighted reduction over the lanes.  Measured 5.3 -> see DESIGN.md on 2048 rows x 2048 u64.
__global__ __launch_bounds__(128) void k_rows_fold_lanes(const G1Jac* __restrict__ buckets, uint32_t B, int c, int W, G1Jac* __restrict__ out) {
    __shared__ G1Jac sm[2];
    const size_t row = blockIdx.x;
    const uint32_t b = threadIdx.x + 1;
    G1Jac h = g1_identity();
    if (b <= B) {
        const G1Jac* bk = buckets + row * (size_t)W * (B + 1) + b;
        for (int w = W - 1; w >= 0; --w) {
            for (int k = 0; k < c; ++k) h = g1_double(h);
            h = g1_add(h, bk[(size_t)w * (B + 1)]);
        }
        h = g1_mul_small(h, b);
    }
    h = wave_sum_g1(h, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) {
        G1Jac acc = blockDim.x > 64 ? g1_add(sm[0], sm[1]) : sm[0];
        out[row] = g1_is_identity(acc) ? g1_identity() : acc;
    }
}

// One-hot chunks: window = chunk, key = hot row + 1 (0 = cold cycle, skipped), no signs.
__global__ __launch_bounds__(kBlock) void k_onehot_keys(const uint8_t* __restrict__ idx, uint32_t wide, size_t n, uint32_t width_log, uint32_t K,
                                                       uint32_t* __restrict__ keys, uint32_t* __restrict__ hist) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < n;
    const uint32_t v = live ? hot_load(idx, i, wide) : kColdIdx;
    uint32_t mag = v == kColdIdx ? 0u : v + 1;
    if (live) keys[i] = mag;
    const uint32_t slot = (uint32_t)((i >> width_log) * (K + 1) + mag);
    WaveAgg ag = wave_aggregate(slot, mag != 0);
    if (ag.do_atomic) atomicAdd(&hist[slot], ag.count);
}
__global__ __launch_bounds__(kBlock) void k_onehot_emit(const G1Jac* __restrict__ buckets, uint32_t K, size_t total, G1Jac* __restrict__ out) {
    size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= total) return;
    G1Jac b = buckets[(t / K) * (K + 1) + (t % K) + 1];
    out[t] = g1_is_identity(b) ? g1_identity() : b;
}

struct BucketPlan {
    int L;
    uint32_t heavy_threshold, heavy_cap;
};
// V bucket sets of B buckets over n points each
BucketPlan plan_buckets(size_t V, uint32_t B, size_t n) {
    BucketPlan p;
    p.L = 1;
    while (p.L < 64 && V * B * (size_t)(2 * p.L) <= 524288) p.L *= 2;  // enough lanes to fill the chip when there are few buckets
    size_t avg = (n + B - 1) / B;
    p.heavy_threshold = (uint32_t)std::min<size_t>((size_t)p.L * std::max<size_t>(kLaneCap, 2 * avg), 0x7FFFFFFFu);
    size_t pts = V * n;
    p.heavy_cap = (uint32_t)std::min<size_t>(pts / kHeavySeg + pts / p.heavy_threshold + 16, 0x7FFFFFFFu);
    return p;  // callers keep V * (B + 1) < 2^32 (bucket slots are u32) by batching at 2^26 points
}

struct Workspace {
    uint32_t *keys, *sorted, *hist, *offs, *cur, *heavy, *hcnt;
    G1Jac *buckets, *seg, *out;
};
// Carve lane 0's grow-only MSM workspace for V windows of n points, B buckets each, `outs` result points.
int32_t carve(jolt_ctx* ctx, size_t V, size_t n, uint32_t B, uint32_t heavy_cap, size_t outs, Workspace* w) {
    const size_t VB = V * (B + 1);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    size_t o_keys = take(V * n * 4), o_sorted = take(V * n * 4), o_hist = take(VB * 4), o_offs = take(VB * 4), o_cur = take(VB * 4),
           o_heavy = take((size_t)heavy_cap * 8), o_hcnt = take(256), o_buckets = take(VB * sizeof(G1Jac)),
           o_seg = take((size_t)heavy_cap * sizeof(G1Jac)), o_out = take(outs * sizeof(G1Jac));
    if (off > ctx->msm_ws_cap[0]) {
        if (ctx->msm_ws[0]) {
            JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            JOLT_HIP_TRY(ctx, hipFree(ctx->msm_ws[0]));
            ctx->msm_ws[0] = nullptr;
            ctx->msm_ws_cap[0] = 0;
        }
        JOLT_HIP_TRY(ctx, hipMalloc(&ctx->msm_ws[0], off));
        ctx->msm_ws_cap[0] = off;
    }
    char* ws = (char*)ctx->msm_ws[0];
    w->keys = (uint32_t*)(ws + o_keys);
    w->sorted = (uint32_t*)(ws + o_sorted);
    w->hist = (uint32_t*)(ws + o_hist);
    w->offs = (uint32_t*)(ws + o_offs);
    w->cur = (uint32_t*)(ws + o_cur);
    w->heavy = (uint32_t*)(ws + o_heavy);
    w->hcnt = (uint32_t*)(ws + o_hcnt);
    w->buckets = (G1Jac*)(ws + o_buckets);
    w->seg = (G1Jac*)(ws + o_seg);
    w->out = (G1Jac*)(ws + o_out);
    return JOLT_OK;
}

// scan -> scatter -> light / heavy bucket sums for V windows whose keys and histogram are in place (already_sorted: hist / offsets /
// sorted / heavy list were produced by k_rows_sort_lds)
void launch_bucket_sums(jolt_ctx* ctx, const Workspace& w, const G1Affine* bases, size_t V, size_t n, uint32_t B, const BucketPlan& p,
                        bool already_sorted = false) {
    hipStream_t st = ctx->stream;
    const unsigned gn = (unsigned)((n + kBlock - 1) / kBlock);
    const unsigned gy = (unsigned)std::min<size_t>(V, 32768), gz = (unsigned)((V + gy - 1) / gy);
    const unsigned gh = std::min<uint32_t>((p.heavy_cap + 3) / 4, 4096);
    if (!already_sorted) {
        hipLaunchKernelGGL(k_msm_scan, dim3((unsigned)V), dim3(kBlock), 0, st, (const uint32_t*)w.hist, w.offs, w.cur, B, p.heavy_threshold, w.heavy, w.hcnt,
                           p.heavy_cap);
        hipLaunchKernelGGL(k_msm_scatter, dim3(gn, gy, gz), dim3(kBlock), 0, st, (const uint32_t*)w.keys, n, B, w.cur, w.sorted, V);
    }
    hipLaunchKernelGGL(k_msm_buckets_light<false>, dim3((unsigned)(((size_t)B * p.L + kBlock - 1) / kBlock), gy, gz), dim3(kBlock), 0, st, (const uint32_t*)w.hist,
                       (const uint32_t*)w.offs, (const uint32_t*)w.sorted, bases, n, B, p.L, p.heavy_threshold, w.buckets, V);
    hipLaunchKernelGGL(k_msm_buckets_heavy<false>, dim3(gh), dim3(kBlock), 0, st, (const uint32_t*)w.heavy, (const uint32_t*)w.hcnt, (const uint32_t*)w.hist,
                       (const uint32_t*)w.offs, (const uint32_t*)w.sorted, bases, n, B, w.seg, LformConsts{});
    hipLaunchKernelGGL(k_msm_heavy_combine, dim3(gh), dim3(kBlock), 0, st, (const uint32_t*)w.heavy, (const uint32_t*)w.hcnt, (const uint32_t*)w.hist,
                       (const G1Jac*)w.seg, w.buckets);
}

int log2_exact(size_t v) {
    if (v == 0 || (v & (v - 1))) return -1;
    int l = 0;
    while (((size_t)1 << l) < v) ++l;
    return l;
}

int32_t hip_fail(jolt_ctx* ctx, const char* what, hipError_t e) {
    ctx->last_error = std::string(what) + ": " + hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? JOLT_ERR_OOM : JOLT_ERR_HIP;
}

}  // namespace

extern "C" int32_t jolt_ints_upload(jolt_ctx* ctx, const void* host, int32_t kind, size_t count, jolt_ints** out) {
    if (!ctx || !out || (!host && count)) return JOLT_ERR_INVALID_ARG;
    if (kind != JOLT_INT_U64 && kind != JOLT_INT_I64 && kind != JOLT_INT_I128) return JOLT_ERR_INVALID_ARG;
    jolt_ints* v = new (std::nothrow) jolt_ints();
    if (!v) return JOLT_ERR_OOM;
    v->ctx = ctx;
    v->count = count;
    v->kind = kind;
    hipError_t e = hipMalloc(&v->data, std::max<size_t>(count, 1) * int_bytes(kind));
    if (e == hipSuccess && count) e = hipMemcpyAsync(v->data, host, count * int_bytes(kind), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        if (v->data) (void)hipFree(v->data);
        delete v;
        return hip_fail(ctx, "ints upload", e);
    }
    *out = v;
    return JOLT_OK;
}

extern "C" int32_t jolt_ints_free(jolt_ctx* ctx, jolt_ints* v) {
    if (!v) return JOLT_OK;
    jolt_ctx* c = v->ctx ? v->ctx : ctx;
    const bool pooled = c && v->data && c->pool_live.count(v->data);  // jolt_ints_from_rows: back to the pool, reused in stream order; uploads: the runtime's block
    if (c && !pooled) (void)hipStreamSynchronize(c->stream);
    if (v->data) { if (c) jolt_internal_dev_free(c, v->data); else (void)hipFree(v->data); }
    delete v;
    return JOLT_OK;
}

extern "C" int32_t jolt_dory_commit_rows(jolt_ctx* ctx, const jolt_srs* srs, const jolt_ints* values, size_t row_width, jolt_g1_t* out) {
    if (!ctx || !srs || !values || (!out && values->count)) return JOLT_ERR_INVALID_ARG;
    const int wl = log2_exact(row_width);
    JOLT_REQUIRE(ctx, wl >= 0, "streaming: row width must be a power of two");  // streaming.rs:99-102
    if (row_width > srs->n) return JOLT_ERR_SRS_TOO_SMALL;                        // :103-108
    if (values->count % row_width) return JOLT_ERR_SIZE_MISMATCH;                 // :192-195
    const size_t rows = values->count / row_width;
    if (rows == 0) return JOLT_OK;
    hipStream_t st = ctx->stream;
    const int kind = values->kind;

    // ---- bit length of the largest magnitude (pinned scratch: the lane-0 MSM result buffer)
    if (!ctx->msm_host[0]) JOLT_HIP_TRY(ctx, hipHostMalloc(&ctx->msm_host[0], 128 * sizeof(G1Jac), hipHostMallocDefault));
    Workspace probe;
    JOLT_TRY(carve(ctx, 1, 1, 1, 16, 1, &probe));
    JOLT_HIP_TRY(ctx, hipMemsetAsync(probe.hcnt, 0, 256, st));
    {
        unsigned g = (unsigned)std::min<size_t>((values->count + kBlock - 1) / kBlock, 4096);
        if (kind == JOLT_INT_U64) hipLaunchKernelGGL(k_ints_or<JOLT_INT_U64>, dim3(g), dim3(kBlock), 0, st, (const void*)values->data, values->count, probe.hcnt);
        else if (kind == JOLT_INT_I64) hipLaunchKernelGGL(k_ints_or<JOLT_INT_I64>, dim3(g), dim3(kBlock), 0, st, (const void*)values->data, values->count, probe.hcnt);
        else hipLaunchKernelGGL(k_ints_or<JOLT_INT_I128>, dim3(g), dim3(kBlock), 0, st, (const void*)values->data, values->count, probe.hcnt);
    }
    uint32_t* h_or = (uint32_t*)ctx->msm_host[0];
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(h_or, probe.hcnt, 16, hipMemcpyDeviceToHost, st));
    JOLT_HIP_TRY(ctx, hipStreamSynchronize(st));
    int bits = 0;
    for (int k = 3; k >= 0 && !bits; --k)
        if (h_or[k]) bits = 32 * k + 32 - __builtin_clz(h_or[k]);
    if (bits == 0) {  // all-zero batch: every row commitment is the identity (Bn254G1::default())
        G1Jac id = g1_identity();
        for (size_t r = 0; r < rows; ++r) std::memcpy(&out[r], &id, sizeof(id));
        return JOLT_OK;
    }

    // ---- plan: ~16 points per bucket; the top window keeps one spare bit for the signed-digit carry
    const int c = std::max(3, std::min(13, wl - 4));
    const int W = (bits + 1 + c - 1) / c;
    if (W > kMaxRowWindows) return JOLT_ERR_UNSUPPORTED;
    const uint32_t B = 1u << (c - 1);
    const size_t batch_rows = std::max<size_t>(1, std::min<size_t>(rows, ((size_t)1 << 26) / ((size_t)W * row_width)));
    for (size_t r0 = 0; r0 < rows; r0 += batch_rows) {
        const size_t nr = std::min(batch_rows, rows - r0), V = nr * (size_t)W, nvals = nr * row_width;
        BucketPlan p = plan_buckets(V, B, row_width);
        Workspace w;
        JOLT_TRY(carve(ctx, V, row_width, B, p.heavy_cap, nr, &w));
        const size_t VB = V * (B + 1);
        hipError_t e = hipMemsetAsync(w.hist, 0, VB * 4, st);
        if (e == hipSuccess) e = hipMemsetAsync(w.hcnt, 0, 256, st);
        if (e == hipSuccess) e = hipMemsetAsync(w.buckets, 0, VB * sizeof(G1Jac), st);  // z = 0: identity
        if (e != hipSuccess) return hip_fail(ctx, "dory rows", e);
        const unsigned gv = (unsigned)((nvals + kBlock - 1) / kBlock);
        const size_t first = r0 * row_width;
        const size_t sort_lds = ((size_t)B + 1 + kBlock) * sizeof(uint32_t);
        const bool lds_sort = ctx->msm_lds_sort && sort_lds <= 64 * 1024;
        if (lds_sort) {  // one workgroup per row sorts its W windows in LDS
            const void* vd = (const void*)values->data;
            if (kind == JOLT_INT_U64)
                hipLaunchKernelGGL(k_rows_sort_lds<JOLT_INT_U64>, dim3((unsigned)nr), dim3(kBlock), sort_lds, st, vd, first, (uint32_t)wl, c, W, w.hist, w.offs, w.sorted,
                                   p.heavy_threshold, w.heavy, w.hcnt, p.heavy_cap);
            else if (kind == JOLT_INT_I64)
                hipLaunchKernelGGL(k_rows_sort_lds<JOLT_INT_I64>, dim3((unsigned)nr), dim3(kBlock), sort_lds, st, vd, first, (uint32_t)wl, c, W, w.hist, w.offs, w.sorted,
                                   p.heavy_threshold, w.heavy, w.hcnt, p.heavy_cap);
            else
                hipLaunchKernelGGL(k_rows_sort_lds<JOLT_INT_I128>, dim3((unsigned)nr), dim3(kBlock), sort_lds, st, vd, first, (uint32_t)wl, c, W, w.hist, w.offs, w.sorted,
                                   p.heavy_threshold, w.heavy, w.hcnt, p.heavy_cap);
        } else if (kind == JOLT_INT_U64)
            hipLaunchKernelGGL(k_rows_digits<JOLT_INT_U64>, dim3(gv), dim3(kBlock), 0, st, (const void*)values->data, first, nvals, (uint32_t)wl, c, W, w.keys, w.hist);
        else if (kind == JOLT_INT_I64)
            hipLaunchKernelGGL(k_rows_digits<JOLT_INT_I64>, dim3(gv), dim3(kBlock), 0, st, (const void*)values->data, first, nvals, (uint32_t)wl, c, W, w.keys, w.hist);
        else
            hipLaunchKernelGGL(k_rows_digits<JOLT_INT_I128>, dim3(gv), dim3(kBlock), 0, st, (const void*)values->data, first, nvals, (uint32_t)wl, c, W, w.keys, w.hist);
        launch_bucket_sums(ctx, w, srs->pts, V, row_width, B, p, lds_sort);
        if (B <= 128)
            hipLaunchKernelGGL(k_rows_fold_lanes, dim3((unsigned)nr), dim3(B <= 64 ? 64 : 128), 0, st, (const G1Jac*)w.buckets, B, c, W, w.out);
        else
            hipLaunchKernelGGL(k_rows_fold, dim3((unsigned)nr), dim3(kBlock), 0, st, (const G1Jac*)w.buckets, B, c, W, w.out);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(out + r0, w.out, nr * sizeof(G1Jac), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return hip_fail(ctx, "dory rows", e);
    }
    return JOLT_OK;
}

extern "C" int32_t jolt_dory_commit_onehot(jolt_ctx* ctx, const jolt_srs* srs, const jolt_onehot* source, size_t poly, size_t chunk_width, jolt_g1_t* out) {
    if (!ctx || !srs || !source || !out) return JOLT_ERR_INVALID_ARG;
    if (poly >= source->n_polys) return JOLT_ERR_INVALID_ARG;
    const int wl = log2_exact(chunk_width);
    JOLT_REQUIRE(ctx, wl >= 0, "streaming one-hot: chunk length must be a power of two");  // streaming.rs:376-380
    if (chunk_width > srs->n) return JOLT_ERR_SRS_TOO_SMALL;                                 // :381-392
    if (source->cycles % chunk_width) return JOLT_ERR_SIZE_MISMATCH;
    const size_t chunks = source->cycles / chunk_width;
    if (chunks == 0) return JOLT_OK;
    const uint32_t K = source->k;
    hipStream_t st = ctx->stream;
    const size_t batch = std::max<size_t>(1, std::min<size_t>(chunks, ((size_t)1 << 26) / chunk_width));
    for (size_t c0 = 0; c0 < chunks; c0 += batch) {
        const size_t V = std::min(batch, chunks - c0), nvals = V * chunk_width;
        BucketPlan p = plan_buckets(V, K, chunk_width);
        Workspace w;
        JOLT_TRY(carve(ctx, V, chunk_width, K, p.heavy_cap, V * K, &w));
        const size_t VB = V * (K + 1);
        hipError_t e = hipMemsetAsync(w.hist, 0, VB * 4, st);
        if (e == hipSuccess) e = hipMemsetAsync(w.hcnt, 0, 256, st);
        if (e == hipSuccess) e = hipMemsetAsync(w.buckets, 0, VB * sizeof(G1Jac), st);
        if (e != hipSuccess) return hip_fail(ctx, "dory one-hot", e);
        const uint8_t* idx = source->idx + ((poly * source->cycles + c0 * chunk_width) << source->wide);
        hipLaunchKernelGGL(k_onehot_keys, dim3((unsigned)((nvals + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, idx, source->wide, nvals, (uint32_t)wl, K, w.keys, w.hist);
        launch_bucket_sums(ctx, w, srs->pts, V, chunk_width, K, p);
        hipLaunchKernelGGL(k_onehot_emit, dim3((unsigned)((V * K + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, (const G1Jac*)w.buckets, K, V * K, w.out);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(out + c0 * K, w.out, V * K * sizeof(G1Jac), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return hip_fail(ctx, "dory one-hot", e);
    }
    return JOLT_OK;
}
