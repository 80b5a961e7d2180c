// jolt_amd/csrc/msm.hip -- BN254 G1 multi-scalar multiplication (bucket method) and SRS handling on gfx950.
//
// Replaces JoltGroup::msm for Bn254G1 (crates/jolt-crypto/src/ec/group.rs:63-70, ec/bn254/mod.rs:195-212 ->
// ark_ec::VariableBaseMSM::msm_bigint) and the per-call affine conversion of every base (mod.rs:205): bases are
// converted once at upload and stay resident as 64-byte affine points.
//
// Pipeline (all integer VALU work, v_mad_u64_u32 bound; no MFMA):
//   1. digits   : scalar -> canonical integer -> W signed c-bit digits; per-(window,|digit|) histogram (atomics)
//   2. scan     : exclusive prefix sums per window (bucket start offsets); over-full buckets are listed per segment
//   3. scatter  : counting-sort the point indices into bucket order
//   4. buckets  : L lanes per light bucket / one wavefront per heavy-bucket segment sum the points (mixed Jacobian+affine adds)
//   5. windows  : sum_b b*B_b per window by running sums over bucket ranges, tree-reduced in LDS
//   6. host     : Horner over the W window sums (c doublings each) -- W*c ~ 256 doublings on the host (64-bit limbs there)
// Corner cases (P+P, P+(-P), infinity) are handled inside the group law (g1.hip.h); the result is the same POINT as
// the reference's, in some Jacobian representation.
#include <algorithm>
#include <cstring>
#include <vector>

#include "ctx.hpp"
#include "msm_kernels.hip.h"
#include "srs.hpp"
#include "term_map.hip.h"

using namespace jolt;
using namespace jolt::msmk;

static_assert(sizeof(G1Jac) == sizeof(jolt_g1_t), "G1Jac must be layout-compatible with jolt_g1_t");

namespace {

__global__ __launch_bounds__(kBlock) void k_jac_to_affine(const G1Jac* __restrict__ in, G1Affine* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = g1_to_affine(in[i]);
}
__global__ __launch_bounds__(kBlock) void k_affine_to_jac(const G1Affine* __restrict__ in, G1Jac* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = g1_from_affine(in[i]);
}

// g1_powers[i] = beta^i * g  (HyperKZGScheme::setup_from_secret, crates/jolt-hyperkzg/src/scheme.rs:54-73)
// out[j] = beta^(the index of the rank's j-th term) * g: the rank's compact SRS under a sharded term assignment (term_map.hip.h)
__global__ __launch_bounds__(kBlock) void k_srs_powers(Fr beta, G1Jac g, G1Affine* __restrict__ out, size_t n, TermMap map) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const size_t exponent = term_global(map, i);
    Fr acc = Fr::one(), base = beta;  // beta^exponent by square-and-multiply on the index
    for (size_t e = exponent; e; e >>= 1) {
        if (e & 1) acc = mul(acc, base);
        base = sqr(base);
    }
    Fr k = from_mont(acc);
    out[i] = g1_to_affine(g1_mul_canonical(g, k.l));
}

}  // namespace
struct MsmJob {
    size_t n = 0;
    int lane = 0, c = 0, W = 0;
    uint32_t nb = 0;
    int results = 1;  // 2: a pair of MSMs over the same scalars (jolt_internal_msm_fixed_enqueue with pair_shift): the second result follows the first in msm_host
};
int32_t jolt_internal_msm_fixed_enqueue(jolt_ctx* ctx, const jolt_srs* srs, const Fr* d_scalars, size_t n, int lane, MsmJob* job, size_t pair_shift = 0);  // msm_fixed.hip
namespace {
struct MsmPlan {
    int c, W, L;  // window bits, windows, lanes per light bucket
    uint32_t B, G, nb, heavy_threshold;
};
MsmPlan plan_for(size_t n) {
    int lg = 0;
    while (((size_t)2 << lg) <= n) lg++;
    MsmPlan p;
    p.c = std::max(2, std::min(16, lg - 4));  // ~16 points per bucket up to 2^20 terms; wider windows measured no faster at 2^22 (19.7 ms either way)
    p.W = (255 + p.c - 1) / p.c;
    p.B = 1u << (p.c - 1);
    // window reduction: nb blocks of 256 threads per window, G buckets per thread
    uint32_t threads = std::min<uint32_t>(p.B, 4096);
    p.nb = (threads + kBlock - 1) / kBlock;
    p.G = (p.B + p.nb * kBlock - 1) / (p.nb * kBlock);
    // lanes per light bucket: enough threads to fill 256 CUs x 4 SIMDs x 8 waves when W*B alone is too small
    size_t wb = (size_t)p.W * p.B;
    p.L = 1;
    while (p.L < 64 && wb * (size_t)(2 * p.L) <= 524288) p.L *= 2;
    // n points over B magnitudes per window; the top window of 254-bit scalars only has ~12.4 k distinct digits (2.6x the
    // average per bucket) and must stay on the light path: one wavefront per 340-point bucket spends its time in the butterfly
    size_t avg = (n + (size_t)p.B - 1) / (size_t)p.B;
    p.heavy_threshold = (uint32_t)std::min<size_t>((size_t)p.L * std::max<size_t>(kLaneCap, 4 * avg), 0x7FFFFFFFu);
    return p;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// SRS
// ------------------------------------------------------------------------------------------------------------------
extern "C" int32_t jolt_srs_upload_g1(jolt_ctx* ctx, const jolt_g1_t* bases, size_t n, jolt_srs** out) {
    if (!ctx || !out || (!bases && n)) return JOLT_ERR_INVALID_ARG;
    jolt_srs* s = new (std::nothrow) jolt_srs();
    if (!s) return JOLT_ERR_OOM;
    s->ctx = ctx;
    s->n = n;
    G1Jac* tmp = nullptr;
    hipError_t e = hipMalloc((void**)&s->pts, std::max<size_t>(n, 1) * sizeof(G1Affine));
    if (e == hipSuccess && n) e = hipMalloc((void**)&tmp, n * sizeof(G1Jac));
    if (e == hipSuccess && n) e = hipMemcpyAsync(tmp, bases, n * sizeof(G1Jac), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && n) {
        hipLaunchKernelGGL(k_jac_to_affine, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, (const G1Jac*)tmp, s->pts, n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (tmp) (void)hipFree(tmp);
    if (e != hipSuccess) {
        ctx->last_error = std::string("srs upload: ") + hipGetErrorString(e);
        if (s->pts) (void)hipFree(s->pts);
        delete s;
        return e == hipErrorOutOfMemory ? JOLT_ERR_OOM : JOLT_ERR_HIP;
    }
    *out = s;
    return JOLT_OK;
}

static int32_t srs_setup_impl(jolt_ctx* ctx, const jolt_fr_t* beta, size_t count, const jolt_g1_t* g1, const TermMap& map, jolt_srs** out);
extern "C" int32_t jolt_srs_setup_from_secret(jolt_ctx* ctx, const jolt_fr_t* beta, size_t count, const jolt_g1_t* g1, jolt_srs** out) {
    return srs_setup_impl(ctx, beta, count, g1, TermMap{}, out);
}
// One rank's share of the same powers under the block-cyclic term assignment: the count_global / world powers
// beta^i with (i / block) % world == rank, compacted in index order.
extern "C" int32_t jolt_srs_setup_from_secret_blocks(jolt_ctx* ctx, const jolt_fr_t* beta, size_t count_global, const jolt_g1_t* g1, size_t block, int32_t rank,
                                                     int32_t world, jolt_srs** out) {
    TermMap m;
    if (!make_block_map(block, rank, world, &m)) return JOLT_ERR_INVALID_ARG;
    if (count_global % (block * (size_t)world) != 0) return JOLT_ERR_SIZE_MISMATCH;
    return srs_setup_impl(ctx, beta, count_global / (size_t)world, g1, m, out);
}
// ... and under the subtree assignment (term_map.hip.h): count_global a power of two >= world, world a power of two
extern "C" int32_t jolt_srs_setup_from_secret_subtree(jolt_ctx* ctx, const jolt_fr_t* beta, size_t count_global, const jolt_g1_t* g1, int32_t rank, int32_t world,
                                                      jolt_srs** out) {
    TermMap m;
    if (!make_subtree_map(rank, world, &m)) return JOLT_ERR_INVALID_ARG;
    if (count_global < (size_t)world || (count_global & (count_global - 1)) != 0) return JOLT_ERR_SIZE_MISMATCH;
    return srs_setup_impl(ctx, beta, count_global / (size_t)world, g1, m, out);
}
static int32_t srs_setup_impl(jolt_ctx* ctx, const jolt_fr_t* beta, size_t count, const jolt_g1_t* g1, const TermMap& map, jolt_srs** out) {
    if (!ctx || !beta || !g1 || !out) return JOLT_ERR_INVALID_ARG;
    Fr b = fr_from_abi(beta);
    JOLT_REQUIRE(ctx, fr_is_canonical(b), "beta is not a canonical Fr");
    G1Jac g;
    std::memcpy(&g, g1, sizeof(g));
    jolt_srs* s = new (std::nothrow) jolt_srs();
    if (!s) return JOLT_ERR_OOM;
    s->ctx = ctx;
    s->n = count;
    hipError_t e = hipMalloc((void**)&s->pts, std::max<size_t>(count, 1) * sizeof(G1Affine));
    if (e == hipSuccess && count) {
        hipLaunchKernelGGL(k_srs_powers, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, b, g, s->pts, count, map);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        ctx->last_error = std::string("srs setup: ") + hipGetErrorString(e);
        if (s->pts) (void)hipFree(s->pts);
        delete s;
        return e == hipErrorOutOfMemory ? JOLT_ERR_OOM : JOLT_ERR_HIP;
    }
    *out = s;
    return JOLT_OK;
}

extern "C" int32_t jolt_srs_len(const jolt_srs* srs, size_t* n) {
    if (!srs || !n) return JOLT_ERR_INVALID_ARG;
    *n = srs->n;
    return JOLT_OK;
}

extern "C" int32_t jolt_srs_download(jolt_ctx* ctx, const jolt_srs* srs, size_t offset, size_t n, jolt_g1_t* out) {
    if (!ctx || !srs || (!out && n)) return JOLT_ERR_INVALID_ARG;
    if (n > srs->n || offset > srs->n - n) return JOLT_ERR_SIZE_MISMATCH;
    if (!n) return JOLT_OK;
    G1Jac* tmp = nullptr;
    JOLT_HIP_TRY(ctx, hipMalloc((void**)&tmp, n * sizeof(G1Jac)));
    hipLaunchKernelGGL(k_affine_to_jac, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, (const G1Affine*)(srs->pts + offset), tmp, n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, tmp, n * sizeof(G1Jac), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return JOLT_ERR_HIP; }
    return JOLT_OK;
}

extern "C" int32_t jolt_srs_free(jolt_ctx* ctx, jolt_srs* srs) {
    if (!srs) return JOLT_OK;
    jolt_ctx* c = ctx ? ctx : srs->ctx;
    if (c) (void)hipStreamSynchronize(c->stream);
    if (srs->pts) (void)hipFree(srs->pts);
    if (srs->pre) (void)hipFree(srs->pre);
    if (srs->mid_tables) {
        if (srs->mid_tables->pre) (void)hipFree(srs->mid_tables->pre);
        delete srs->mid_tables;
    }
    delete srs;
    return JOLT_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// MSM
// ------------------------------------------------------------------------------------------------------------------
constexpr size_t kMsmHostEntries = 128;  // >= W for every plan (c = 2: 128 windows)

// Enqueue one MSM on lane `lane` (its stream, workspace and pinned result buffer); nothing blocks the host.
int32_t jolt_internal_msm_enqueue(jolt_ctx* ctx, const jolt_srs* srs, const Fr* d_scalars, size_t n, int lane, MsmJob* job) {
    job->n = n;
    job->lane = lane;
    job->results = 1;
    if (n > srs->n) return JOLT_ERR_SRS_TOO_SMALL;
    if (n == 0) return JOLT_OK;
    if (n >= ((size_t)1 << 31)) return JOLT_ERR_UNSUPPORTED;
    // mid-length MSMs of FULL-WIDTH scalars: the mid table set (srs.hpp).  The caller says so (hyperkzg.hip around the level commitments): 64-bit witness scalars would leave
    // its 20-bit windows a 4-bit top window whose n digits pile onto 16 buckets (measured: the commit leg 21.9 -> 29.5 ms)
    const jolt_srs* tables = (srs->mid_tables && ctx->msm_full_width_scalars && n <= srs->mid_tables->n && n >= srs->mid_tables->pre_min_n) ? srs->mid_tables : srs;
    if (tables->pre && ctx->msm_fixed && n >= tables->pre_min_n) {  // window-precomputed bases: one bucket set for all windows (msm_fixed.hip)
        int32_t fs = jolt_internal_msm_fixed_enqueue(ctx, tables, d_scalars, n, lane, job);
        if (fs != JOLT_ERR_UNSUPPORTED) return fs;  // skewed scalars fall through to the per-window method and its heavy-bucket kernels
    }
    MsmPlan p = plan_for(n);
    job->c = p.c;
    job->W = p.W;
    job->nb = p.nb;
    const size_t WB = (size_t)p.W * (p.B + 1);
    // heavy buckets hold > heavy_threshold points each: at most W*n/threshold of them, W*n/kHeavySeg + one entry per bucket
    const uint32_t heavy_cap = (uint32_t)((size_t)p.W * n / kHeavySeg + (size_t)p.W * n / p.heavy_threshold + 16);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    size_t o_keys = take((size_t)p.W * n * 4), o_sorted = take((size_t)p.W * n * 4), o_hist = take(WB * 4), o_offs = take(WB * 4),
           o_cur = take(WB * 4), o_heavy = take((size_t)heavy_cap * 8), o_hcnt = take(256), o_buckets = take(WB * sizeof(G1Jac)),
           o_part = take((size_t)p.W * p.nb * sizeof(G1Jac)), o_seg = take((size_t)heavy_cap * sizeof(G1Jac)), o_wsum = take((size_t)p.W * sizeof(G1Jac));
    hipStream_t st = lane == 0 ? ctx->stream : ctx->side[lane - 1];
    if (off > ctx->msm_ws_cap[lane]) {  // grow-only (hipMalloc / hipFree per MSM cost more than a small MSM itself)
        if (ctx->msm_ws[lane]) {
            JOLT_HIP_TRY(ctx, hipStreamSynchronize(st));
            JOLT_HIP_TRY(ctx, hipFree(ctx->msm_ws[lane]));
            ctx->msm_ws[lane] = nullptr;
            ctx->msm_ws_cap[lane] = 0;
        }
        JOLT_HIP_TRY(ctx, hipMalloc(&ctx->msm_ws[lane], off));
        ctx->msm_ws_cap[lane] = off;
    }
    if (!ctx->msm_host[lane]) JOLT_HIP_TRY(ctx, hipHostMalloc(&ctx->msm_host[lane], kMsmHostEntries * sizeof(G1Jac), hipHostMallocDefault));
    if ((size_t)p.W > kMsmHostEntries) return JOLT_ERR_UNSUPPORTED;
    char* ws = (char*)ctx->msm_ws[lane];
    uint32_t* keys = (uint32_t*)(ws + o_keys);
    uint32_t* sorted = (uint32_t*)(ws + o_sorted);
    uint32_t* hist = (uint32_t*)(ws + o_hist);
    uint32_t* offs = (uint32_t*)(ws + o_offs);
    uint32_t* cur = (uint32_t*)(ws + o_cur);
    uint32_t* heavy = (uint32_t*)(ws + o_heavy);
    uint32_t* hcnt = (uint32_t*)(ws + o_hcnt);
    G1Jac* buckets = (G1Jac*)(ws + o_buckets);
    G1Jac* part = (G1Jac*)(ws + o_part);
    G1Jac* seg = (G1Jac*)(ws + o_seg);
    G1Jac* wsum = (G1Jac*)(ws + o_wsum);
    hipError_t e = hipMemsetAsync(hist, 0, WB * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(hcnt, 0, 256, st);
    if (e == hipSuccess) e = hipMemsetAsync(buckets, 0, WB * sizeof(G1Jac), st);  // z = 0: identity
    if (e == hipSuccess) {
        unsigned gn = (unsigned)((n + kBlock - 1) / kBlock);
        unsigned gh = std::min<uint32_t>((heavy_cap + 3) / 4, 4096);
        // counting sort: per-workgroup LDS histograms when a window's counters fit (they do on gfx950 up to c = 16) and the MSM
        // is large enough for the slices to amortise their flush; per-key global atomics otherwise
        const size_t lds_bytes = ((size_t)p.B + 1) * sizeof(uint32_t);
        const bool lds_sort = ctx->msm_lds_sort && n >= ((size_t)1 << 16) && lds_bytes <= ctx->max_lds_per_block;
        const unsigned slices = (unsigned)std::max<size_t>(1, std::min<size_t>((size_t)ctx->num_cus / p.W, n / 8192));
        if (lds_sort && !ctx->msm_lds_attr_set) {  // once per context (= per device): allow more than 64 KiB of dynamic LDS
            hipError_t a1 = hipFuncSetAttribute((const void*)k_msm_hist_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->max_lds_per_block);
            hipError_t a2 = hipFuncSetAttribute((const void*)k_msm_scatter_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->max_lds_per_block);
            if (a1 != hipSuccess || a2 != hipSuccess) {
                (void)hipGetLastError();
                ctx->msm_lds_sort = false;  // this runtime refuses the large allocation: keep the per-key path
            }
            ctx->msm_lds_attr_set = true;
        }
        const bool use_lds_sort = lds_sort && ctx->msm_lds_sort;
        if (use_lds_sort) {
            hipLaunchKernelGGL(k_msm_digits, dim3(gn), dim3(kBlock), 0, st, d_scalars, n, p.c, p.W, keys, (uint32_t*)nullptr);
            hipLaunchKernelGGL(k_msm_hist_lds, dim3(slices, p.W), dim3(kSortBlock), lds_bytes, st, (const uint32_t*)keys, n, p.B, hist);
        } else {
            hipLaunchKernelGGL(k_msm_digits, dim3(gn), dim3(kBlock), 0, st, d_scalars, n, p.c, p.W, keys, hist);
        }
        hipLaunchKernelGGL(k_msm_scan, dim3(p.W), dim3(kBlock), 0, st, (const uint32_t*)hist, offs, cur, p.B, p.heavy_threshold, heavy, hcnt, heavy_cap);
        if (use_lds_sort) hipLaunchKernelGGL(k_msm_scatter_lds, dim3(slices, p.W), dim3(kSortBlock), lds_bytes, st, (const uint32_t*)keys, n, p.B, cur, sorted);
        else hipLaunchKernelGGL(k_msm_scatter, dim3(gn, p.W), dim3(kBlock), 0, st, (const uint32_t*)keys, n, p.B, cur, sorted, (size_t)p.W);
        hipLaunchKernelGGL(k_msm_buckets_light<true>, dim3((unsigned)(((size_t)p.B * p.L + kBlock - 1) / kBlock), p.W), dim3(kBlock), 0, st, (const uint32_t*)hist,
                           (const uint32_t*)offs, (const uint32_t*)sorted, (const G1Affine*)srs->pts, n, p.B, p.L, p.heavy_threshold, buckets, (size_t)p.W);
        hipLaunchKernelGGL(k_msm_buckets_heavy<false>, dim3(gh), dim3(kBlock), 0, st, (const uint32_t*)heavy, (const uint32_t*)hcnt, (const uint32_t*)hist,
                           (const uint32_t*)offs, (const uint32_t*)sorted, (const G1Affine*)srs->pts, n, p.B, seg, LformConsts{});
        hipLaunchKernelGGL(k_msm_heavy_combine, dim3(gh), dim3(kBlock), 0, st, (const uint32_t*)heavy, (const uint32_t*)hcnt, (const uint32_t*)hist,
                           (const G1Jac*)seg, buckets);
        hipLaunchKernelGGL(k_msm_window_reduce, dim3(p.nb, p.W), dim3(kBlock), 0, st, (const G1Jac*)buckets, p.B, p.G, part);
        hipLaunchKernelGGL(k_msm_window_combine, dim3(p.W), dim3(64), 0, st, (const G1Jac*)part, p.nb, wsum);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->msm_host[lane], wsum, (size_t)p.W * sizeof(G1Jac), hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) {
        ctx->last_error = std::string("msm: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? JOLT_ERR_OOM : JOLT_ERR_HIP;
    }
    return JOLT_OK;
}

// Wait for the lane and finish on the host: Horner over the windows, acc = 2^c * acc + S_w.
int32_t jolt_internal_msm_collect(jolt_ctx* ctx, const MsmJob* job, G1Jac* out) {  // out: job->results points
    if (job->n == 0) {
        for (int r = 0; r < job->results; ++r) out[r] = g1_identity();
        return JOLT_OK;
    }
    hipStream_t st = job->lane == 0 ? ctx->stream : ctx->side[job->lane - 1];
    JOLT_HIP_TRY(ctx, hipStreamSynchronize(st));
    for (int r = 0; r < job->results; ++r) {
        const G1Jac* wsum = (const G1Jac*)ctx->msm_host[job->lane] + (size_t)r * job->W;
        G1Jac acc = g1_identity();
        for (int w = job->W - 1; w >= 0; --w) {
            for (int k = 0; k < job->c; ++k) acc = g1_double(acc);
            acc = g1_add(acc, wsum[w]);
        }
        out[r] = acc;
    }
    return JOLT_OK;
}
// sum_i s_i P_i and sum_i s_i P_(i + shift) on `lane` with ONE sort of the scalars' digits (msm_fixed.hip); JOLT_ERR_UNSUPPORTED when this SRS / length does not take the
// fixed-base method (the caller then runs two MSMs)
int32_t jolt_internal_msm_enqueue_pair(jolt_ctx* ctx, const jolt_srs* srs, const Fr* d_scalars, size_t n, size_t shift, int lane, MsmJob* job) {
    job->n = n;
    job->lane = lane;
    job->results = 2;
    if (shift > srs->n || n > srs->n - shift) return JOLT_ERR_SRS_TOO_SMALL;
    if (n == 0) return JOLT_OK;
    if (n >= ((size_t)1 << 31)) return JOLT_ERR_UNSUPPORTED;
    if (!(srs->pre && ctx->msm_fixed && n >= srs->pre_min_n)) return JOLT_ERR_UNSUPPORTED;
    return jolt_internal_msm_fixed_enqueue(ctx, srs, d_scalars, n, lane, job, shift);
}

// Sharded term assignments (term_map.hip.h): a rank's SRS object holds exactly its terms' bases, compacted in index order, so the
// terms a rank owns of ANY prefix [0, n) are a prefix of its compact arrays.
// host hooks (no GPU): the owned-term count of a prefix and the index of a compact slot, for callers sizing per-rank buffers and
// building a rank's compact SRS from a full one, and for the CPU tests
extern "C" int32_t jolt_host_owned_terms(size_t n, size_t block, int32_t rank, int32_t world, size_t* out) {
    TermMap m;
    if (!out || !make_block_map(block, rank, world, &m)) return JOLT_ERR_INVALID_ARG;
    *out = term_owned(m, n);
    return JOLT_OK;
}
extern "C" int32_t jolt_host_subtree_owned_terms(size_t n, int32_t rank, int32_t world, size_t* out) {
    TermMap m;
    if (!out || !make_subtree_map(rank, world, &m)) return JOLT_ERR_INVALID_ARG;
    *out = term_owned(m, n);
    return JOLT_OK;
}
extern "C" int32_t jolt_host_subtree_term_index(size_t slot, int32_t rank, int32_t world, size_t* out) {
    TermMap m;
    if (!out || !make_subtree_map(rank, world, &m)) return JOLT_ERR_INVALID_ARG;
    *out = term_global(m, slot);
    return JOLT_OK;
}
namespace {
__global__ __launch_bounds__(kBlock) void k_gather_owned_terms(const Fr* __restrict__ src, Fr* __restrict__ dst, size_t len, TermMap map) {
    const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= len) return;
    dst[j] = src[term_global(map, j)];
}
}  // namespace
// dst[0 .. owned) = the scalars of the rank's terms of src[0 .. n), on the main stream
int32_t jolt_internal_gather_owned_terms(jolt_ctx* ctx, const Fr* src, size_t n, const TermMap& map, Fr* dst) {
    const size_t len = term_owned(map, n);
    if (!len) return JOLT_OK;
    hipLaunchKernelGGL(k_gather_owned_terms, dim3((unsigned)((len + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, src, dst, len, map);
    JOLT_HIP_TRY(ctx, hipGetLastError());
    return JOLT_OK;
}

int32_t jolt_internal_msm(jolt_ctx* ctx, const jolt_srs* srs, const Fr* d_scalars, size_t n, G1Jac* out) {
    MsmJob job;
    if (ctx->msm_tables_pending) { ctx->last_error = "an MSM while jolt_msm_g1_tables_begin's MSMs are in flight (finish them first)"; return JOLT_ERR_INVALID_ARG; }
    JOLT_TRY(jolt_internal_msm_enqueue(ctx, srs, d_scalars, n, 0, &job));
    return jolt_internal_msm_collect(ctx, &job, out);
}

// ------------------------------------------------------------------------------------------------------------------
// A BATCH of short MSMs over prefixes of the same bases in one pass of the per-window kernels (the level commitments of a HyperKZG opening below the table sets'
// crossover: 18 MSMs of 2^18 ... 2 terms, each of which cost ~13 launches, a stream synchronisation and a host Horner on its own: 8 ms of an opening with no bucket
// sum running, profiles/r04_open_exposed.txt).  MSM j is padded with zero scalars to the longest length N and every (MSM, window) pair becomes one WINDOW of the
// per-window method: keys[(j W + w) N + i], W' = count * W windows of B + 1 buckets -- histogram, scan, scatter, bucket sums and window reductions are the kernels
// of jolt_internal_msm_enqueue with W' windows, the sorted entries are base indices i as before, and the host's Horner runs per MSM over its own W window sums.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kBatchMax = 32;                       // MSMs per batch
constexpr size_t kBatchMaxLen = (size_t)1 << 18;    // longest member: below the mid table set's crossover (msm_fixed.hip: kMidMin = 2^19)
struct BatchScalars {
    const Fr* ptr[kBatchMax];
    uint32_t len[kBatchMax];
};
struct MsmBatchJob {
    int count = 0, c = 0, W = 0;
    bool queued = false;
};
namespace {
__global__ __launch_bounds__(kBlock) void k_msm_digits_batch(BatchScalars b, size_t N, int c, int W, uint32_t* __restrict__ keys) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const uint32_t j = blockIdx.y;
    if (i >= N) return;
    const bool live = i < b.len[j];
    Fr s = live ? from_mont(ld_fr(b.ptr[j] + i)) : Fr::zero();
    const uint32_t B = 1u << (c - 1);
    uint32_t carry = 0;
    for (int w = 0; w < W; ++w) {
        const int bit = w * c;
        uint32_t raw = 0;
        if (bit < 256) {
            const int limb = bit >> 5, off = bit & 31;
            const uint64_t two = (uint64_t)s.l[limb] | (limb + 1 < 8 ? (uint64_t)s.l[limb + 1] << 32 : 0ull);
            raw = (uint32_t)(two >> off) & ((1u << c) - 1);
        }
        raw += carry;
        uint32_t mag, negf;
        if (raw > B) { mag = (1u << c) - raw; negf = 1; carry = 1; }
        else { mag = raw; negf = 0; carry = 0; }
        keys[((size_t)j * W + w) * N + i] = mag | (negf << 31);  // padding: zero keys (skipped by the sort)
    }
}
}  // namespace
static int32_t msm_batch_enqueue(jolt_ctx* ctx, const jolt_srs* srs, const Fr* const* d_scalars, const size_t* n, const int* members, int count, MsmBatchJob* job) {
    job->count = count;
    job->queued = false;
    if (count == 0) return JOLT_OK;
    size_t N = 0;
    BatchScalars bs;
    for (int k = 0; k < count; ++k) {
        bs.ptr[k] = d_scalars[members[k]];
        bs.len[k] = (uint32_t)n[members[k]];
        N = std::max(N, n[members[k]]);
    }
    for (int k = count; k < kBatchMax; ++k) { bs.ptr[k] = nullptr; bs.len[k] = 0; }
    if (N > srs->n) return JOLT_ERR_SRS_TOO_SMALL;
    int lg = 0;
    while (((size_t)1 << lg) < N) lg++;
    // windows of ~64 points per bucket in the longest member: the shorter ones leave most of their buckets empty, which the reduction passes over quickly
    const int c = std::max(3, std::min(12, lg - 6)), W = (255 + c - 1) / c, Wv = count * W;
    const uint32_t B = 1u << (c - 1);
    const uint32_t G = 8, nb = (uint32_t)(((B + G - 1) / G + kBlock - 1) / kBlock);  // reduction: chains of 8 buckets
    const size_t WB = (size_t)Wv * (B + 1);
    const size_t avg = (N + B - 1) / B;
    const uint32_t heavy_threshold = (uint32_t)std::max<size_t>(kLaneCap, 4 * avg);
    const uint32_t heavy_cap = (uint32_t)((size_t)Wv * N / kHeavySeg + (size_t)Wv * N / heavy_threshold + 16);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_keys = take((size_t)Wv * N * 4), o_sorted = take((size_t)Wv * N * 4), o_hist = take(WB * 4), o_offs = take(WB * 4), o_cur = take(WB * 4),
                 o_heavy = take((size_t)heavy_cap * 8), o_hcnt = take(256), o_buckets = take(WB * sizeof(G1Jac)), o_part = take((size_t)Wv * nb * sizeof(G1Jac)),
                 o_seg = take((size_t)heavy_cap * sizeof(G1Jac)), o_wsum = take((size_t)Wv * sizeof(G1Jac));
    if (!ctx->msm_batch_stream) JOLT_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->msm_batch_stream, hipStreamNonBlocking));
    hipStream_t st = ctx->msm_batch_stream;
    if (off > ctx->msm_batch_ws_cap) {
        if (ctx->msm_batch_ws) { JOLT_HIP_TRY(ctx, hipStreamSynchronize(st)); JOLT_HIP_TRY(ctx, hipFree(ctx->msm_batch_ws)); ctx->msm_batch_ws = nullptr; ctx->msm_batch_ws_cap = 0; }
        JOLT_HIP_TRY(ctx, hipMalloc(&ctx->msm_batch_ws, off));
        ctx->msm_batch_ws_cap = off;
    }
    if ((size_t)Wv * sizeof(G1Jac) > ctx->msm_batch_host_cap) {
        if (ctx->msm_batch_host) { JOLT_HIP_TRY(ctx, hipStreamSynchronize(st)); JOLT_HIP_TRY(ctx, hipHostFree(ctx->msm_batch_host)); ctx->msm_batch_host = nullptr; ctx->msm_batch_host_cap = 0; }
        JOLT_HIP_TRY(ctx, hipHostMalloc(&ctx->msm_batch_host, (size_t)Wv * sizeof(G1Jac), hipHostMallocDefault));
        ctx->msm_batch_host_cap = (size_t)Wv * sizeof(G1Jac);
    }
    char* ws = (char*)ctx->msm_batch_ws;
    uint32_t *keys = (uint32_t*)(ws + o_keys), *sorted = (uint32_t*)(ws + o_sorted), *hist = (uint32_t*)(ws + o_hist), *offs = (uint32_t*)(ws + o_offs), *cur = (uint32_t*)(ws + o_cur),
             *heavy = (uint32_t*)(ws + o_heavy), *hcnt = (uint32_t*)(ws + o_hcnt);
    G1Jac *buckets = (G1Jac*)(ws + o_buckets), *part = (G1Jac*)(ws + o_part), *seg = (G1Jac*)(ws + o_seg), *wsum = (G1Jac*)(ws + o_wsum);
    // the scalars are produced on the main stream (ev_fork is recorded there by the caller)
    JOLT_HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_fork, 0));
    JOLT_HIP_TRY(ctx, hipMemsetAsync(hist, 0, WB * 4, st));
    JOLT_HIP_TRY(ctx, hipMemsetAsync(hcnt, 0, 256, st));
    JOLT_HIP_TRY(ctx, hipMemsetAsync(buckets, 0, WB * sizeof(G1Jac), st));
    const unsigned gn = (unsigned)((N + kBlock - 1) / kBlock), gh = std::min<uint32_t>((heavy_cap + 3) / 4, 4096);
    hipLaunchKernelGGL(k_msm_digits_batch, dim3(gn, count), dim3(kBlock), 0, st, bs, N, c, W, keys);
    // counting sort through LDS, ONE workgroup per window (<= 2^18 keys, (B + 1) * 4 <= 8 KiB of counters): hundreds of windows fill the chip by themselves
    const size_t lds_bytes = ((size_t)B + 1) * sizeof(uint32_t);
    hipLaunchKernelGGL(k_msm_hist_lds, dim3(1, Wv), dim3(kSortBlock), lds_bytes, st, (const uint32_t*)keys, N, B, hist);
    hipLaunchKernelGGL(k_msm_scan, dim3(Wv), dim3(kBlock), 0, st, (const uint32_t*)hist, offs, cur, B, heavy_threshold, heavy, hcnt, heavy_cap);
    hipLaunchKernelGGL(k_msm_scatter_lds, dim3(1, Wv), dim3(kSortBlock), lds_bytes, st, (const uint32_t*)keys, N, B, cur, sorted);
    hipLaunchKernelGGL(k_msm_buckets_light<true>, dim3((unsigned)(((size_t)B + kBlock - 1) / kBlock), Wv), dim3(kBlock), 0, st, (const uint32_t*)hist, (const uint32_t*)offs,
                       (const uint32_t*)sorted, (const G1Affine*)srs->pts, N, B, 1, heavy_threshold, buckets, (size_t)Wv);
    hipLaunchKernelGGL(k_msm_buckets_heavy<false>, dim3(gh), dim3(kBlock), 0, st, (const uint32_t*)heavy, (const uint32_t*)hcnt, (const uint32_t*)hist, (const uint32_t*)offs,
                       (const uint32_t*)sorted, (const G1Affine*)srs->pts, N, B, seg, LformConsts{});
    hipLaunchKernelGGL(k_msm_heavy_combine, dim3(gh), dim3(kBlock), 0, st, (const uint32_t*)heavy, (const uint32_t*)hcnt, (const uint32_t*)hist, (const G1Jac*)seg, buckets);
    hipLaunchKernelGGL(k_msm_window_reduce, dim3(nb, Wv), dim3(kBlock), 0, st, (const G1Jac*)buckets, B, G, part);
    hipLaunchKernelGGL(k_msm_window_combine, dim3(Wv), dim3(64), 0, st, (const G1Jac*)part, nb, wsum);
    JOLT_HIP_TRY(ctx, hipGetLastError());
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(ctx->msm_batch_host, wsum, (size_t)Wv * sizeof(G1Jac), hipMemcpyDeviceToHost, st));
    job->c = c;
    job->W = W;
    job->queued = true;
    return JOLT_OK;
}
// wait for the batch and finish on the host: per member the Horner over its W window sums
static int32_t msm_batch_collect(jolt_ctx* ctx, const MsmBatchJob* job, const int* members, G1Jac* out) {
    if (!job->queued) return JOLT_OK;
    JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->msm_batch_stream));
    for (int k = 0; k < job->count; ++k) {
        const G1Jac* wsum = (const G1Jac*)ctx->msm_batch_host + (size_t)k * job->W;
        G1Jac acc = g1_identity();
        for (int w = job->W - 1; w >= 0; --w) {
            for (int d = 0; d < job->c; ++d) acc = g1_double(acc);
            acc = g1_add(acc, wsum[w]);
        }
        out[members[k]] = acc;
    }
    return JOLT_OK;
}

// `count` independent MSMs over the same bases, pipelined over the four lanes (results in order).  The side lanes wait
// for the work already queued on the main stream (the tables being committed are produced there).
int32_t jolt_internal_msm_many(jolt_ctx* ctx, const jolt_srs* srs, const Fr* const* d_scalars, const size_t* n, size_t count, G1Jac* out,
                               const size_t* base_offsets /* per MSM, or nullptr: every MSM multiplies the prefix */) {
    if (count == 0) return JOLT_OK;
    if (ctx->msm_tables_pending) { ctx->last_error = "MSMs while jolt_msm_g1_tables_begin's MSMs are in flight (finish them first)"; return JOLT_ERR_INVALID_ARG; }
    JOLT_HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    for (int k = 0; k < 3; ++k) JOLT_HIP_TRY(ctx, hipStreamWaitEvent(ctx->side[k], ctx->ev_fork, 0));
    MsmJob jobs[4];
    int32_t status = JOLT_OK;
    const size_t L = (size_t)ctx->msm_lanes;
    // the short prefix MSMs go as ONE batch on their own stream, queued first and collected (host Horner included) while the lanes still hold long MSMs
    int members[kBatchMax], n_members = 0;
    std::vector<size_t> rest;
    for (size_t i = 0; i < count; ++i) {
        const bool prefix = !base_offsets || base_offsets[i] == 0;
        if (ctx->msm_batch && prefix && n[i] >= 2 && n[i] <= kBatchMaxLen && n[i] <= srs->n && n_members < kBatchMax) members[n_members++] = (int)i;
        else rest.push_back(i);
    }
    if (n_members < 2) {  // nothing to share
        rest.clear();
        for (size_t i = 0; i < count; ++i) rest.push_back(i);
        n_members = 0;
    }
    MsmBatchJob batch;
    if (n_members) status = msm_batch_enqueue(ctx, srs, d_scalars, n, members, n_members, &batch);
    const size_t R = rest.size();
    for (size_t k = 0; k < R + L; ++k) {
        int lane = (int)(k % L);
        if (k == R && status == JOLT_OK && n_members) status = msm_batch_collect(ctx, &batch, members, out);  // every long MSM is queued: the lanes are busy
        if (k >= L && k - L < R && status == JOLT_OK) status = jolt_internal_msm_collect(ctx, &jobs[lane], &out[rest[k - L]]);
        if (k < R && status == JOLT_OK) {
            const size_t i = rest[k];
            if (base_offsets && base_offsets[i] + n[i] > srs->n) status = JOLT_ERR_SRS_TOO_SMALL;
            const jolt_srs view = jolt_srs_range_view(*srs, base_offsets && status == JOLT_OK ? base_offsets[i] : 0);  // consumed by the enqueue itself
            if (status == JOLT_OK) status = jolt_internal_msm_enqueue(ctx, &view, d_scalars[i], n[i], lane, &jobs[lane]);
        }
    }
    if (status != JOLT_OK) {
        for (int k = 0; k < 3; ++k) (void)hipStreamSynchronize(ctx->side[k]);
        if (ctx->msm_batch_stream) (void)hipStreamSynchronize(ctx->msm_batch_stream);
    }
    return status;
}

// out[0] = sum_i a_i P_i, out[1] = sum_i a_i P_(i + shift) (one sort of a's digits for both: msm_fixed.hip) and out[2] = sum_i b_i P_i on a second lane beside them.
// JOLT_ERR_UNSUPPORTED, with nothing enqueued, when the pair cannot take the fixed-base method.
int32_t jolt_internal_msm_pair_and_one(jolt_ctx* ctx, const jolt_srs* srs, const Fr* d_a, size_t n_a, size_t shift, const Fr* d_b, size_t n_b, G1Jac* out) {
    if (shift > srs->n || n_a > srs->n - shift) return JOLT_ERR_SRS_TOO_SMALL;
    if (n_a == 0 || !(srs->pre && ctx->msm_fixed && n_a >= srs->pre_min_n) || ctx->msm_tables_pending) return JOLT_ERR_UNSUPPORTED;
    JOLT_HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    for (int k = 0; k < 3; ++k) JOLT_HIP_TRY(ctx, hipStreamWaitEvent(ctx->side[k], ctx->ev_fork, 0));
    MsmJob pair, one;
    int32_t status = jolt_internal_msm_enqueue_pair(ctx, srs, d_a, n_a, shift, 0, &pair);
    if (status == JOLT_ERR_UNSUPPORTED) return status;  // (decided before any launch: window count / LDS limits)
    const int lane_b = ctx->msm_lanes > 1 ? 1 : 0;
    if (status == JOLT_OK && lane_b == 0) status = jolt_internal_msm_collect(ctx, &pair, out);  // one lane: its workspace serves one MSM at a time
    if (status == JOLT_OK) status = jolt_internal_msm_enqueue(ctx, srs, d_b, n_b, lane_b, &one);
    if (status == JOLT_OK && lane_b != 0) status = jolt_internal_msm_collect(ctx, &pair, out);
    if (status == JOLT_OK) status = jolt_internal_msm_collect(ctx, &one, out + 2);
    if (status != JOLT_OK)
        for (int k = 0; k < 3; ++k) (void)hipStreamSynchronize(ctx->side[k]);
    return status;
}

// The same three results with the SINGLE MSM started early: jolt_internal_msm_one_begin enqueues sum_i b_i P_i on lane 1 as soon as b exists (its sort then runs under
// whatever the main stream does next -- in an opening, the scan that produces the pair's scalars), jolt_internal_msm_pair_finish enqueues the pair on lane 0 and
// collects both.  begin returns JOLT_ERR_UNSUPPORTED, with nothing enqueued, when the pair could not take the fixed-base method anyway or there is only one lane
// (the caller then uses jolt_internal_msm_pair_and_one or three MSMs).
struct MsmPendingOne {
    MsmJob job;
};
int32_t jolt_internal_msm_one_begin(jolt_ctx* ctx, const jolt_srs* srs, size_t n_a, size_t shift, const Fr* d_b, size_t n_b) {
    if (shift > srs->n || n_a > srs->n - shift) return JOLT_ERR_SRS_TOO_SMALL;
    if (n_a == 0 || !(srs->pre && ctx->msm_fixed && n_a >= srs->pre_min_n) || ctx->msm_lanes < 2 || ctx->msm_pending_one || ctx->msm_tables_pending) return JOLT_ERR_UNSUPPORTED;
    if ((size_t)srs->pre_W * n_a >= ((size_t)1 << 32)) return JOLT_ERR_UNSUPPORTED;  // what jolt_internal_msm_fixed_enqueue would refuse for the pair
    MsmPendingOne* p = new (std::nothrow) MsmPendingOne();
    if (!p) return JOLT_ERR_OOM;
    int32_t status = JOLT_OK;
    hipError_t e = hipEventRecord(ctx->ev_fork, ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->side[0], ctx->ev_fork, 0);
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); status = JOLT_ERR_HIP; }
    if (status == JOLT_OK) status = jolt_internal_msm_enqueue(ctx, srs, d_b, n_b, 1, &p->job);
    if (status != JOLT_OK) { (void)hipStreamSynchronize(ctx->side[0]); delete p; return status; }
    ctx->msm_pending_one = p;
    return JOLT_OK;
}
// out[0], out[1]: the pair over d_a (shift as in jolt_internal_msm_pair_and_one); out[2]: the MSM begun above
int32_t jolt_internal_msm_pair_finish(jolt_ctx* ctx, const jolt_srs* srs, const Fr* d_a, size_t n_a, size_t shift, G1Jac* out) {
    MsmPendingOne* p = static_cast<MsmPendingOne*>(ctx->msm_pending_one);
    if (!p) return JOLT_ERR_INVALID_ARG;
    ctx->msm_pending_one = nullptr;
    MsmJob pair;
    int32_t status = jolt_internal_msm_enqueue_pair(ctx, srs, d_a, n_a, shift, 0, &pair);  // lane 0 = the main stream: ordered behind d_a's producer
    if (status == JOLT_OK) status = jolt_internal_msm_collect(ctx, &pair, out);
    else if (status == JOLT_ERR_UNSUPPORTED) {  // (not expected after begin's checks) two plain MSMs, the second against the shifted bases
        status = jolt_internal_msm(ctx, srs, d_a, n_a, &out[0]);
        const jolt_srs view = jolt_srs_range_view(*srs, shift);
        if (status == JOLT_OK) status = jolt_internal_msm(ctx, &view, d_a, n_a, &out[1]);
    }
    const int32_t one_status = jolt_internal_msm_collect(ctx, &p->job, out + 2);  // always: the lane must be drained before its workspace is used again
    delete p;
    if (status != JOLT_OK) (void)hipStreamSynchronize(ctx->side[0]);
    return status != JOLT_OK ? status : one_status;
}
// a begun MSM whose opening failed in between: drain the lane and drop the state
void jolt_internal_msm_one_abandon(jolt_ctx* ctx) {
    MsmPendingOne* p = static_cast<MsmPendingOne*>(ctx->msm_pending_one);
    if (!p) return;
    ctx->msm_pending_one = nullptr;
    (void)hipStreamSynchronize(ctx->side[0]);
    delete p;
}

// Up to three MSMs over prefixes of one SRS begun on the side lanes and collected later: the dense columns of a commitment (CommitWitness::commit_witness walks the
// committed polynomials one by one, crates/jolt-kernels/src/commitment.rs:137-160; here they are in flight together, and the caller commits its one-hot columns on the
// main stream in between -- their sums are bound by multiplications, these short MSMs by the latency of their sort and reduction chains).
struct jolt_msm_pending {
    int count = 0;
    MsmJob jobs[3];
};
extern "C" int32_t jolt_msm_g1_tables_begin(jolt_ctx* ctx, const jolt_srs* srs, const jolt_table* const* scalars, const size_t* n, size_t count, jolt_msm_pending** out) {
    if (!ctx || !srs || !scalars || !n || !out || count == 0) return JOLT_ERR_INVALID_ARG;
    *out = nullptr;
    if (count > 3 || (size_t)ctx->msm_lanes < count + 1 || ctx->msm_tables_pending || ctx->msm_pending_one) return JOLT_ERR_UNSUPPORTED;
    for (size_t i = 0; i < count; ++i) {
        if (!scalars[i]) return JOLT_ERR_INVALID_ARG;
        if (n[i] > scalars[i]->len) return JOLT_ERR_SIZE_MISMATCH;
        if (n[i] > srs->n) return JOLT_ERR_SRS_TOO_SMALL;
    }
    JOLT_TRY(jolt_internal_engine_quiesce(ctx));
    jolt_msm_pending* p = new (std::nothrow) jolt_msm_pending();
    if (!p) return JOLT_ERR_OOM;
    int32_t status = JOLT_OK;
    hipError_t e = hipEventRecord(ctx->ev_fork, ctx->stream);
    for (size_t i = 0; i < count && e == hipSuccess; ++i) e = hipStreamWaitEvent(ctx->side[i], ctx->ev_fork, 0);
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); status = JOLT_ERR_HIP; }
    for (size_t i = 0; i < count && status == JOLT_OK; ++i) {
        status = jolt_internal_msm_enqueue(ctx, srs, scalars[i]->data(), n[i], (int)i + 1, &p->jobs[i]);
        if (status == JOLT_OK) p->count = (int)i + 1;
    }
    if (status != JOLT_OK) {
        for (int k = 0; k < 3; ++k) (void)hipStreamSynchronize(ctx->side[k]);
        delete p;
        return status;
    }
    ctx->msm_tables_pending = true;  // the side lanes' workspaces are taken: no other MSM of this context until the finish
    *out = p;
    return JOLT_OK;
}
extern "C" int32_t jolt_msm_g1_tables_finish(jolt_ctx* ctx, jolt_msm_pending* p, jolt_g1_t* out) {
    if (!ctx || !p) return JOLT_ERR_INVALID_ARG;
    int32_t status = JOLT_OK;
    for (int i = 0; i < p->count; ++i) {
        G1Jac r;
        const int32_t s = jolt_internal_msm_collect(ctx, &p->jobs[i], &r);  // always: the lane is drained even after an earlier failure
        if (s == JOLT_OK && out) std::memcpy(&out[i], &r, sizeof(r));
        if (status == JOLT_OK) status = s;
    }
    ctx->msm_tables_pending = false;
    delete p;
    if (status == JOLT_OK && !out) status = JOLT_ERR_INVALID_ARG;
    return status;
}

extern "C" int32_t jolt_msm_g1_table(jolt_ctx* ctx, const jolt_srs* srs, const jolt_table* scalars, size_t n, jolt_g1_t* out) {
    if (!ctx || !srs || !scalars || !out) return JOLT_ERR_INVALID_ARG;
    if (n > scalars->len) return JOLT_ERR_SIZE_MISMATCH;
    G1Jac r;
    JOLT_TRY(jolt_internal_msm(ctx, srs, scalars->data(), n, &r));
    std::memcpy(out, &r, sizeof(r));
    return JOLT_OK;
}

// the same for scalars the caller KNOWS to be full-width field elements (a polynomial folded by a challenge): mid-length MSMs then take the mid table set
// (msm_fixed.hip; 64-bit witness scalars must not: their top window would pile onto a few buckets)
extern "C" int32_t jolt_msm_g1_table_full_width(jolt_ctx* ctx, const jolt_srs* srs, const jolt_table* scalars, size_t n, jolt_g1_t* out) {
    if (!ctx) return JOLT_ERR_INVALID_ARG;
    ctx->msm_full_width_scalars = ctx->msm_uniform_scalars = true;  // the caller vouches for uniform field elements: mid tables AND the capacity sort
    const int32_t s = jolt_msm_g1_table(ctx, srs, scalars, n, out);
    ctx->msm_full_width_scalars = ctx->msm_uniform_scalars = false;
    return s;
}

// sum_i scalars[scalar_offset + i] * srs[base_offset + i], i < n: one rank's term range of a sharded MSM
extern "C" int32_t jolt_msm_g1_table_range(jolt_ctx* ctx, const jolt_srs* srs, size_t base_offset, const jolt_table* scalars, size_t scalar_offset, size_t n,
                                           jolt_g1_t* out) {
    if (!ctx || !srs || !scalars || !out) return JOLT_ERR_INVALID_ARG;
    if (n > scalars->len || scalar_offset > scalars->len - n) return JOLT_ERR_SIZE_MISMATCH;
    if (n > srs->n || base_offset > srs->n - n) return JOLT_ERR_SRS_TOO_SMALL;
    const jolt_srs view = jolt_srs_range_view(*srs, base_offset);
    G1Jac r;
    JOLT_TRY(jolt_internal_msm(ctx, &view, scalars->data() + scalar_offset, n, &r));
    std::memcpy(out, &r, sizeof(r));
    return JOLT_OK;
}

// One window of a STREAMED commitment (StreamingCommitment::{feed, feed_u64, feed_i128}, crates/jolt-openings/src/schemes.rs:167-222): the polynomial arrives in
// coefficient order as host slices of field elements or machine integers; the window's share is sum_i values[i] * srs[base_offset + i], added to the running
// partial commitment.  Stateless on purpose: the caller's PartialCommitment is (point, next coefficient index), a plain value it can clone.
extern "C" int32_t jolt_msm_g1_window(jolt_ctx* ctx, const jolt_srs* srs, size_t base_offset, int32_t kind, const void* host, size_t n, const jolt_g1_t* acc, jolt_g1_t* out) {
    if (!ctx || !srs || !out || (!host && n)) return JOLT_ERR_INVALID_ARG;
    if (kind != JOLT_SCALAR_FR && kind != JOLT_INT_U64 && kind != JOLT_INT_I64 && kind != JOLT_INT_I128) return JOLT_ERR_INVALID_ARG;
    if (n > srs->n || base_offset > srs->n - n) return JOLT_ERR_SRS_TOO_SMALL;
    G1Jac sum = g1_identity();
    if (acc) std::memcpy(&sum, acc, sizeof(sum));
    if (n) {
        jolt_table* t = nullptr;
        if (kind == JOLT_SCALAR_FR) {
            JOLT_TRY(jolt_table_upload(ctx, (const jolt_fr_t*)host, n, &t));
        } else {
            jolt_ints* v = nullptr;
            JOLT_TRY(jolt_ints_upload(ctx, host, kind, n, &v));
            const int32_t s = jolt_table_from_ints(ctx, v, 0, n, &t);
            jolt_ints_free(ctx, v);
            if (s != JOLT_OK) return s;
        }
        jolt_g1_t part;
        const int32_t s = jolt_msm_g1_table_range(ctx, srs, base_offset, t, 0, n, &part);
        jolt_table_free(ctx, t);
        if (s != JOLT_OK) return s;
        G1Jac p;
        std::memcpy(&p, &part, sizeof(p));
        sum = g1_add(sum, p);
    }
    std::memcpy(out, &sum, sizeof(sum));
    return JOLT_OK;
}

// rank's share of sum_{i < n} scalars[i] * SRS[i] under a sharded term assignment: `srs` is the rank's compact SRS
// (jolt_srs_setup_from_secret_blocks / _subtree, or an upload of the same points); the shares of all ranks add up to the MSM
static int32_t msm_share(jolt_ctx* ctx, const jolt_srs* srs, const jolt_table* scalars, size_t n, const TermMap& map, jolt_g1_t* out) {
    if (!ctx || !srs || !scalars || !out) return JOLT_ERR_INVALID_ARG;
    if (n > scalars->len) return JOLT_ERR_SIZE_MISMATCH;
    const size_t len = term_owned(map, n);
    if (len > srs->n) return JOLT_ERR_SRS_TOO_SMALL;
    Fr* compact = nullptr;
    JOLT_TRY(jolt_internal_dev_alloc(ctx, std::max<size_t>(len, 1) * sizeof(Fr), (void**)&compact));
    int32_t s = jolt_internal_gather_owned_terms(ctx, scalars->data(), n, map, compact);
    G1Jac r;
    if (s == JOLT_OK) s = jolt_internal_msm(ctx, srs, compact, len, &r);
    jolt_internal_dev_free(ctx, compact);
    if (s == JOLT_OK) std::memcpy(out, &r, sizeof(r));
    return s;
}
extern "C" int32_t jolt_msm_g1_table_blocks(jolt_ctx* ctx, const jolt_srs* srs, const jolt_table* scalars, size_t n, size_t block, int32_t rank, int32_t world,
                                            jolt_g1_t* out) {
    TermMap m;
    if (!make_block_map(block, rank, world, &m)) return JOLT_ERR_INVALID_ARG;
    return msm_share(ctx, srs, scalars, n, m, out);
}
extern "C" int32_t jolt_msm_g1_table_subtree(jolt_ctx* ctx, const jolt_srs* srs, const jolt_table* scalars, size_t n, int32_t rank, int32_t world, jolt_g1_t* out) {
    TermMap m;
    if (!make_subtree_map(rank, world, &m)) return JOLT_ERR_INVALID_ARG;
    return msm_share(ctx, srs, scalars, n, m, out);
}

extern "C" int32_t jolt_msm_g1(jolt_ctx* ctx, const jolt_srs* srs, const jolt_fr_t* scalars, size_t n, jolt_g1_t* out) {
    if (!ctx || !srs || (!scalars && n) || !out) return JOLT_ERR_INVALID_ARG;
    if (n > srs->n) return JOLT_ERR_SRS_TOO_SMALL;
    jolt_table* t = nullptr;
    JOLT_TRY(jolt_table_upload(ctx, scalars, n, &t));
    int32_t s = jolt_msm_g1_table(ctx, srs, t, n, out);
    jolt_table_free(ctx, t);
    return s;
}

// ------------------------------------------------------------------------------------------------------------------
// host-side G1 helpers (host mirror; no GPU needed)
// ------------------------------------------------------------------------------------------------------------------
extern "C" int32_t jolt_host_g1_add(const jolt_g1_t* p, const jolt_g1_t* q, jolt_g1_t* out) {
    if (!p || !q || !out) return JOLT_ERR_INVALID_ARG;
    G1Jac a, b;
    std::memcpy(&a, p, sizeof(a));
    std::memcpy(&b, q, sizeof(b));
    G1Jac r = g1_add(a, b);
    std::memcpy(out, &r, sizeof(r));
    return JOLT_OK;
}
// is `p` a point of BN254 G1 in the ABI's representation: canonical Montgomery coordinates with Y^2 = X^3 + 3 Z^6, or the identity (Z = 0)?  What every point a caller
// hands in must satisfy before it is absorbed (jolt_host_hyperkzg_open_with_levels); G1 has cofactor 1, so on the curve is in the group.
extern "C" int32_t jolt_host_g1_is_on_curve(const jolt_g1_t* p, int32_t* on_curve) {
    if (!p || !on_curve) return JOLT_ERR_INVALID_ARG;
    G1Jac a;
    std::memcpy(&a, p, sizeof(a));
    *on_curve = g1_is_on_curve(a) ? 1 : 0;
    return JOLT_OK;
}
extern "C" int32_t jolt_host_g1_eq(const jolt_g1_t* p, const jolt_g1_t* q, int32_t* equal) {
    if (!p || !q || !equal) return JOLT_ERR_INVALID_ARG;
    G1Jac a, b;
    std::memcpy(&a, p, sizeof(a));
    std::memcpy(&b, q, sizeof(b));
    *equal = g1_eq(a, b) ? 1 : 0;
    return JOLT_OK;
}
// ark-serialize compressed short-Weierstrass form, as Bn254G1 is appended to transcripts and proofs
// (crates/jolt-crypto/src/ec/bn254/mod.rs:139-171): 32-byte LE x, bit 7 of the last byte = y > -y, bit 6 = infinity.
extern "C" int32_t jolt_host_g1_serialize_compressed(const jolt_g1_t* p, uint8_t out[32]) {
    if (!p || !out) return JOLT_ERR_INVALID_ARG;
    G1Jac a;
    std::memcpy(&a, p, sizeof(a));
    std::memset(out, 0, 32);
    if (g1_is_identity(a)) { out[31] |= 0x40; return JOLT_OK; }
    G1Affine af = g1_to_affine(a);
    Fq xc = from_mont(af.x), yc = from_mont(af.y), nyc = from_mont(neg(af.y));
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(xc.l[i] >> (8 * j));
    bool y_gt_neg = false;  // y > -y as canonical integers
    for (int i = 7; i >= 0; --i) {
        if (yc.l[i] != nyc.l[i]) { y_gt_neg = yc.l[i] > nyc.l[i]; break; }
    }
    if (y_gt_neg) out[31] |= 0x80;
    return JOLT_OK;
}
