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
// Corner cases (P+P, P+(-P), infinity) are handled inside the group law (g1.cuh); the result is the same POINT as
// the reference's, in some Jacobian representation.
#include <algorithm>

#include "ctx.hpp"
#include "g1.cuh"
#include "poly_kernels.cuh"

using namespace jolt;

static_assert(sizeof(G1Jac) == sizeof(jolt_g1_t), "G1Jac must be layout-compatible with jolt_g1_t");

struct jolt_srs {
    jolt_ctx* ctx = nullptr;
    G1Affine* pts = nullptr;
    size_t n = 0;
};

namespace {

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

__global__ __launch_bounds__(kBlock) void k_jac_to_affine(const G1Jac* __restrict__ in, G1Affine* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = g1_to_affine(in[i]);
}
__global__ __launch_bounds__(kBlock) void k_affine_to_jac(const G1Affine* __restrict__ in, G1Jac* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = g1_from_affine(in[i]);
}

// g1_powers[i] = beta^i * g  (HyperKZGScheme::setup_from_secret, crates/jolt-hyperkzg/src/scheme.rs:54-73)
__global__ __launch_bounds__(kBlock) void k_srs_powers(Fr beta, G1Jac g, G1Affine* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Fr acc = Fr::one(), base = beta;  // beta^i by square-and-multiply on the index
    for (size_t e = i; e; e >>= 1) {
        if (e & 1) acc = mul(acc, base);
        base = sqr(base);
    }
    Fr k = from_mont(acc);
    out[i] = g1_to_affine(g1_mul_canonical(g, k.l));
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
        WaveAgg ag = wave_aggregate(mag, live && mag != 0);
        if (ag.do_atomic) atomicAdd(&hist[(size_t)w * (B + 1) + mag], ag.count);
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
__global__ __launch_bounds__(kBlock) void k_msm_scatter(const uint32_t* __restrict__ keys, size_t n, uint32_t B, uint32_t* __restrict__ cursor,
                                                       uint32_t* __restrict__ sorted) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const int w = blockIdx.y;
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

__device__ __forceinline__ G1Jac sum_bucket_points(const uint32_t* __restrict__ src, const G1Affine* __restrict__ bases, uint32_t lo, uint32_t hi,
                                                   uint32_t stride) {
    G1Jac acc = g1_identity();
    for (uint32_t k = lo; k < hi; k += stride) {
        uint32_t v = src[k];
        G1Affine p = ld_aff(bases + (v & 0x7FFFFFFFu));
        if (v >> 31) p.y = neg(p.y);
        acc = g1_add_mixed(acc, p);
    }
    return acc;
}

// ---- 4a. light buckets: L adjacent lanes per bucket (L = 1 when there are enough buckets to fill the chip) ----------
__global__ __launch_bounds__(kBlock) void k_msm_buckets_light(const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offsets,
                                                             const uint32_t* __restrict__ sorted, const G1Affine* __restrict__ bases, size_t n,
                                                             uint32_t B, int L, uint32_t heavy_threshold, G1Jac* __restrict__ buckets) {
    uint32_t gt = blockIdx.x * kBlock + threadIdx.x;
    uint32_t b = gt / L + 1, sub = gt % L;  // bucket magnitude 1..B
    const int w = gridDim.y - 1 - blockIdx.y;  // top window first: its buckets are the fullest for 254-bit scalars
    bool mine = b <= B;
    size_t slot = (size_t)w * (B + 1) + (mine ? b : 0);
    uint32_t cnt = mine ? hist[slot] : 0;
    if (cnt > heavy_threshold) { cnt = 0; mine = false; }  // the heavy kernels own it
    G1Jac acc = sum_bucket_points(sorted + (size_t)w * n + offsets[slot], bases, sub, cnt, (uint32_t)L);
    acc = wave_sum_g1(acc, L);
    if (mine && sub == 0) buckets[slot] = acc;
}

// ---- 4b. heavy buckets: one wavefront per kHeavySeg-point segment, then one wavefront per bucket adds its segment sums ----
__global__ __launch_bounds__(kBlock) void k_msm_buckets_heavy(const uint32_t* __restrict__ heavy_list, const uint32_t* __restrict__ heavy_count,
                                                             const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offsets,
                                                             const uint32_t* __restrict__ sorted, const G1Affine* __restrict__ bases, size_t n,
                                                             uint32_t B, G1Jac* __restrict__ seg_sums) {
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, n_waves = gridDim.x * (kBlock / 64);
    const uint32_t total = *heavy_count;
    for (uint32_t h = wave; h < total; h += n_waves) {
        uint32_t slot = heavy_list[2 * h], sgi = heavy_list[2 * h + 1];
        uint32_t w = slot / (B + 1);
        uint32_t cnt = hist[slot];
        uint32_t lo = sgi * kHeavySeg, hi = min(lo + (uint32_t)kHeavySeg, cnt);
        G1Jac acc = sum_bucket_points(sorted + (size_t)w * n + offsets[slot], bases, lo + lane, hi, 64u);
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

}  // namespace
struct MsmJob {
    size_t n = 0;
    int lane = 0, c = 0, W = 0;
    uint32_t nb = 0;
};
namespace {
// ---- 5b. one wavefront per window adds that window's nb block partials ---------------------------------------------------
__global__ __launch_bounds__(64) void k_msm_window_combine(const G1Jac* __restrict__ partial, uint32_t nb, G1Jac* __restrict__ window_sums) {
    const int w = blockIdx.x;
    G1Jac acc = g1_identity();
    for (uint32_t k = threadIdx.x; k < nb; k += 64) acc = g1_add(acc, partial[(size_t)w * nb + k]);
    acc = wave_sum_g1(acc, 64);
    if (threadIdx.x == 0) window_sums[w] = acc;
}

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
    uint32_t threads = std::min<uint32_t>(p.B, 8192);
    p.nb = (threads + kBlock - 1) / kBlock;
    p.G = (p.B + p.nb * kBlock - 1) / (p.nb * kBlock);
    // lanes per light bucket: enough threads to fill 256 CUs x 4 SIMDs x 8 waves when W*B alone is too small
    size_t wb = (size_t)p.W * p.B;
    p.L = 1;
    while (p.L < 64 && wb * (size_t)(2 * p.L) <= 524288) p.L *= 2;
    size_t avg = (n + 2 * (size_t)p.B - 1) / (2 * (size_t)p.B);
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

extern "C" int32_t jolt_srs_setup_from_secret(jolt_ctx* ctx, const jolt_fr_t* beta, size_t count, const jolt_g1_t* g1, jolt_srs** out) {
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
        hipLaunchKernelGGL(k_srs_powers, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, b, g, s->pts, count);
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
    if (offset + n > srs->n) return JOLT_ERR_SIZE_MISMATCH;
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
    if (n > srs->n) return JOLT_ERR_SRS_TOO_SMALL;
    if (n == 0) return JOLT_OK;
    if (n >= ((size_t)1 << 31)) return JOLT_ERR_UNSUPPORTED;
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
        hipLaunchKernelGGL(k_msm_digits, dim3(gn), dim3(kBlock), 0, st, d_scalars, n, p.c, p.W, keys, hist);
        hipLaunchKernelGGL(k_msm_scan, dim3(p.W), dim3(kBlock), 0, st, (const uint32_t*)hist, offs, cur, p.B, p.heavy_threshold, heavy, hcnt, heavy_cap);
        hipLaunchKernelGGL(k_msm_scatter, dim3(gn, p.W), dim3(kBlock), 0, st, (const uint32_t*)keys, n, p.B, cur, sorted);
        hipLaunchKernelGGL(k_msm_buckets_light, dim3((unsigned)(((size_t)p.B * p.L + kBlock - 1) / kBlock), p.W), dim3(kBlock), 0, st, (const uint32_t*)hist,
                           (const uint32_t*)offs, (const uint32_t*)sorted, (const G1Affine*)srs->pts, n, p.B, p.L, p.heavy_threshold, buckets);
        hipLaunchKernelGGL(k_msm_buckets_heavy, dim3(gh), dim3(kBlock), 0, st, (const uint32_t*)heavy, (const uint32_t*)hcnt, (const uint32_t*)hist,
                           (const uint32_t*)offs, (const uint32_t*)sorted, (const G1Affine*)srs->pts, n, p.B, seg);
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
int32_t jolt_internal_msm_collect(jolt_ctx* ctx, const MsmJob* job, G1Jac* out) {
    if (job->n == 0) { *out = g1_identity(); return JOLT_OK; }
    hipStream_t st = job->lane == 0 ? ctx->stream : ctx->side[job->lane - 1];
    JOLT_HIP_TRY(ctx, hipStreamSynchronize(st));
    const G1Jac* wsum = (const G1Jac*)ctx->msm_host[job->lane];
    G1Jac acc = g1_identity();
    for (int w = job->W - 1; w >= 0; --w) {
        for (int k = 0; k < job->c; ++k) acc = g1_double(acc);
        acc = g1_add(acc, wsum[w]);
    }
    *out = acc;
    return JOLT_OK;
}

int32_t jolt_internal_msm(jolt_ctx* ctx, const jolt_srs* srs, const Fr* d_scalars, size_t n, G1Jac* out) {
    MsmJob job;
    JOLT_TRY(jolt_internal_msm_enqueue(ctx, srs, d_scalars, n, 0, &job));
    return jolt_internal_msm_collect(ctx, &job, out);
}

// `count` independent MSMs over the same bases, pipelined over the four lanes (results in order).  The side lanes wait
// for the work already queued on the main stream (the tables being committed are produced there).
int32_t jolt_internal_msm_many(jolt_ctx* ctx, const jolt_srs* srs, const Fr* const* d_scalars, const size_t* n, size_t count, G1Jac* out) {
    if (count == 0) return JOLT_OK;
    JOLT_HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    for (int k = 0; k < 3; ++k) JOLT_HIP_TRY(ctx, hipStreamWaitEvent(ctx->side[k], ctx->ev_fork, 0));
    MsmJob jobs[4];
    int32_t status = JOLT_OK;
    for (size_t i = 0; i < count + 4; ++i) {
        int lane = (int)(i % 4);
        if (i >= 4 && i - 4 < count && status == JOLT_OK) status = jolt_internal_msm_collect(ctx, &jobs[lane], &out[i - 4]);
        if (i < count && status == JOLT_OK) status = jolt_internal_msm_enqueue(ctx, srs, d_scalars[i], n[i], lane, &jobs[lane]);
    }
    if (status != JOLT_OK)
        for (int k = 0; k < 3; ++k) (void)hipStreamSynchronize(ctx->side[k]);
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
