// jolt_amd/csrc/tools/batched_affine_bench.hip -- measurement tool (not part of the shipped library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 batched_affine_bench.hip -o build/batched_affine_bench && build/batched_affine_bench [log2 pairs]
//
// The question: would BATCHED-AFFINE additions (one shared inversion per batch, Montgomery's trick: 5M + 1S
// per addition against the 8M + 2S of the XYZZ mixed addition) make the bucket sums of the fixed-base MSM faster?  Measured here in the form
// that is MOST favourable to batched-affine: one level of the pairwise tree over bucket lists, R_i = P_i + Q_i for N independent pairs, with
// every input and output COALESCED (in the real pipeline the first level gathers 64-byte points by index, twice) and the same limb-form
// field arithmetic as the production kernel (fq_limb.hip.h).  Each lane owns a batch of M pairs:
//   forward : d_i = x2_i - x1_i, prefix_i = prefix_(i-1) d_i, prefix_i stored to scratch                      (1 M, reads 64 B, writes 36 B)
//   invert  : run = prefix_M^-1 by Fermat in limb form (253 S + ~120 M, amortised over the M pairs)
//   backward: inv_i = run prefix_(i-1), run = run d_i, lambda = (y2 - y1) inv_i, x3 = lambda^2 - x1 - x2, y3 = lambda (x1 - x3) - y1
//                                                                                                        (4 M + 1 S, reads 36 + 128 B, writes 64 B)
// Against it, the XYZZ loop of the production bucket kernel on the same coalesced stream (k_xyzz_stream: a lane accumulates a chain of
// points; the production kernel does the same behind a 4-byte index gather).  Output: additions per second and registers for both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../fq_limb.hip.h"
#include "../g1.hip.h"

using namespace jolt;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

namespace {
constexpr int kBlock = 256;

__device__ __forceinline__ uint32_t mix(uint64_t v) {
    v ^= v >> 33; v *= 0xff51afd7ed558ccdull; v ^= v >> 33; v *= 0xc4ceb9fe1a85ec53ull; v ^= v >> 33;
    return (uint32_t)v;
}
// pseudo-random field elements below 2^253 (< p) as the words of L-form values.  Chord addition is a rational map of the coordinates: the
// points need not lie on the curve for the two formulas to be compared or timed.
__global__ void k_fill(G1Affine* pts, size_t n, uint64_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine p;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        p.x.l[k] = mix(seed + 16 * i + k);
        p.y.l[k] = mix(seed + 16 * i + 8 + k);
    }
    p.x.l[7] &= 0x1FFFFFFFu;
    p.y.l[7] &= 0x1FFFFFFFu;
    pts[i] = p;
}

// random pair of DISTINCT table indices per pair (the table is the 2n-point array)
__global__ void k_fill_idx(uint32_t* idx, size_t pairs, uint32_t table, uint64_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pairs) return;
    const uint32_t a = mix(seed + 2 * i) % table;
    uint32_t b = mix(seed + 2 * i + 1) % table;
    if (b == a) b = (a + 1) % table;
    idx[2 * i] = a;
    idx[2 * i + 1] = b;
}

__device__ __forceinline__ void ld_words(const Fq* p, Fq& out) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1];
    out.l[0] = a.x; out.l[1] = a.y; out.l[2] = a.z; out.l[3] = a.w; out.l[4] = b.x; out.l[5] = b.y; out.l[6] = b.z; out.l[7] = b.w;
}
__device__ __forceinline__ FqL ld_fql(const Fq* p) {
    Fq w;
    ld_words(p, w);
    return fql_from_words(w);
}
// nine normalised limbs of a value below 2^256 -> eight words (no reduction: the value stays whatever representative it was)
__device__ __forceinline__ Fq fql_pack(const FqL& v) {
    Fq out;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = (32 * j) / 29, s = 32 * j - 29 * k;
        uint32_t w = v.l[k] >> s;
        w |= v.l[k + 1] << (29 - s);
        if (58 - s < 32 && k + 2 < 9) w |= v.l[k + 2] << (58 - s);
        out.l[j] = w;
    }
    return out;
}
__device__ __forceinline__ void st_words(Fq* p, const Fq& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// a^(p - 2) in limb form (values carry the factor 2^261: fql_mul keeps it).  p - 2 as 29-bit limbs at compile time.
__device__ __noinline__ FqL fql_inv(const FqL& a, const FqL& one) {
    constexpr uint32_t PL[9] = JOLT_FQL_P;
    FqL r = one;
    // p is odd and its low limb is >= 2: p - 2 only changes limb 0
    for (int k = 8; k >= 0; --k) {
        const uint32_t e = k == 0 ? PL[0] - 2u : PL[k];
        const int top = k == 8 ? 21 : 28;  // p < 2^254 = 2^(8 * 29 + 22)
        for (int b = top; b >= 0; --b) {
            r = fql_sqr(r);
            if ((e >> b) & 1u) r = fql_mul(r, a);  // wave-uniform: the exponent is a constant
        }
    }
    return r;
}

// ---- batched affine: one tree level ------------------------------------------------------------------------------------------------
// pair (t, j) of thread t is element j * threads + t of the arrays: adjacent lanes touch adjacent records in every step
// GATHER: the first level of the real pipeline -- both operands are table points fetched by index (twice: forward and backward pass)
template <int M, bool GATHER>
__global__ __launch_bounds__(kBlock) void k_batched_affine_level(const G1Affine* __restrict__ Pa, const G1Affine* __restrict__ Qa, G1Affine* __restrict__ R,
                                                                 uint32_t* __restrict__ scratch /* 9 words per pair, limb-major */, size_t threads, Fq one_words,
                                                                 const uint32_t* __restrict__ idx) {
    const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= threads) return;
    const size_t n = threads * M;
    const FqL one = fql_from_words(one_words);
    FqL pref = one;
    for (int j = 0; j < M; ++j) {
        const size_t i = (size_t)j * threads + t;
        const G1Affine* P = GATHER ? Pa + idx[2 * i] - i : Pa;  // P[i] below is the gathered record
        const G1Affine* Q = GATHER ? Pa + idx[2 * i + 1] - i : Qa;
        const FqL d = fql_sub(ld_fql(&Q[i].x), ld_fql(&P[i].x));
        pref = fql_mul(pref, d);
#pragma unroll
        for (int k = 0; k < 9; ++k) scratch[(size_t)k * n + i] = pref.l[k];
    }
    FqL run = fql_inv(pref, one);
    for (int j = M - 1; j >= 0; --j) {
        const size_t i = (size_t)j * threads + t;
        FqL before = one;
        if (j > 0) {
            const size_t ip = i - threads;
#pragma unroll
            for (int k = 0; k < 9; ++k) before.l[k] = scratch[(size_t)k * n + ip];
        }
        const G1Affine* P = GATHER ? Pa + idx[2 * i] - i : Pa;
        const G1Affine* Q = GATHER ? Pa + idx[2 * i + 1] - i : Qa;
        const FqL x1 = ld_fql(&P[i].x), y1 = ld_fql(&P[i].y), x2 = ld_fql(&Q[i].x), y2 = ld_fql(&Q[i].y);
        const FqL inv_d = fql_mul(run, before);
        run = fql_mul(run, fql_sub(x2, x1));
        const FqL lam = fql_mul(fql_sub(y2, y1), inv_d);
        const FqL x3 = fql_sub(fql_sub(fql_sqr(lam), x1), x2);
        const FqL y3 = fql_sub(fql_mul(lam, fql_sub(x1, x3)), y1);
        st_words(&R[i].x, fql_pack(x3));
        st_words(&R[i].y, fql_pack(y3));
    }
}

// ---- the production loop on the same stream: a lane accumulates a chain of LEN points in XYZZ ----------------------------------------
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_xyzz_stream(const G1Affine* __restrict__ pts, size_t threads, int len,
                                                                                               G1XyzzL* __restrict__ out, Fq one_words) {
    const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= threads) return;
    const FqL one = fql_from_words(one_words);
    G1XyzzL acc = g1xl_identity();
    Fq nx, ny;
    ld_words(&pts[t].x, nx);
    ld_words(&pts[t].y, ny);
    for (int j = 0; j < len; ++j) {
        const Fq cx = nx, cy = ny;
        if (j + 1 < len) {  // the next point is in flight during the addition, as in sum_bucket_points_lform
            ld_words(&pts[(size_t)(j + 1) * threads + t].x, nx);
            ld_words(&pts[(size_t)(j + 1) * threads + t].y, ny);
        }
        acc = g1xl_add_mixed(acc, fql_from_words(cx), fql_from_words(cy), one);
    }
    out[t] = acc;
}

// ---- check: R_i against the XYZZ mixed addition of the same pair, compared projectively ------------------------------------------------
__global__ void k_check(const G1Affine* __restrict__ Pa, const G1Affine* __restrict__ Qa, const G1Affine* __restrict__ R, size_t n, size_t step, Fq one_words,
                        unsigned int* __restrict__ bad, const uint32_t* __restrict__ idx) {
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i = s * step;
    if (i >= n) return;
    const G1Affine* P = idx ? Pa + idx[2 * i] - i : Pa;
    const G1Affine* Q = idx ? Pa + idx[2 * i + 1] - i : Qa;
    const FqL one = fql_from_words(one_words);
    G1XyzzL acc;
    acc.x = ld_fql(&P[i].x);
    acc.y = ld_fql(&P[i].y);
    acc.zz = one;
    acc.zzz = one;
    const G1XyzzL r = g1xl_add_mixed(acc, ld_fql(&Q[i].x), ld_fql(&Q[i].y), one);
    // X = x3 ZZ, Y = y3 ZZZ (all values carry 2^261 once: fql_mul(x3, zz) is the L-form of x3 zz, r.x * one likewise)
    const FqL lx = fql_mul(ld_fql(&R[i].x), r.zz), ly = fql_mul(ld_fql(&R[i].y), r.zzz);
    const FqL rx = fql_mul(r.x, one), ry = fql_mul(r.y, one);
    if (!fql_is_zero(fql_sub(lx, rx)) || !fql_is_zero(fql_sub(ly, ry))) atomicAdd(bad, 1u);
}

template <int M, bool GATHER>
void run_batched(const G1Affine* P, const G1Affine* Q, G1Affine* R, uint32_t* scratch, size_t n, Fq one_words, unsigned int* d_bad, const uint32_t* idx) {
    const size_t threads = n / M;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((threads + kBlock - 1) / kBlock);
    hipLaunchKernelGGL((k_batched_affine_level<M, GATHER>), dim3(grid), dim3(kBlock), 0, 0, P, Q, R, scratch, threads, one_words, idx);  // warm-up
    CK(hipDeviceSynchronize());
    const int reps = 3;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_batched_affine_level<M, GATHER>), dim3(grid), dim3(kBlock), 0, 0, P, Q, R, scratch, threads, one_words, idx);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    CK(hipMemset(d_bad, 0, 4));
    const size_t step = 4099, checks = (n + step - 1) / step;
    hipLaunchKernelGGL(k_check, dim3((unsigned)((checks + 255) / 256)), dim3(256), 0, 0, P, Q, (const G1Affine*)R, n, step, one_words, d_bad, GATHER ? idx : (const uint32_t*)nullptr);
    unsigned int bad = 0;
    CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_batched_affine_level<M, GATHER>)));
    const double bytes = (double)n * (64 + 36 + 36 + 128 + 64);
    printf("{\"what\": \"batched-affine level, %s\", \"pairs\": %zu, \"batch_per_lane\": %d, \"lanes\": %zu, \"ms\": %.3f, \"G_adds_per_s\": %.2f, \"TB_per_s\": %.2f, \"vgprs\": %d, "
           "\"scratch_bytes_per_thread\": %zu, \"checked\": %zu, \"mismatches\": %u}\n",
           GATHER ? "operands gathered by index from a 2^27-point table" : "coalesced operands", n, M, threads, ms, n / ms * 1e-6, bytes / ms * 1e-9, fa.numRegs, (size_t)fa.localSizeBytes, checks, bad);
    fflush(stdout);
}
}  // namespace

int main(int argc, char** argv) {
    const int log_n = argc > 1 ? atoi(argv[1]) : 26;
    const size_t n = (size_t)1 << log_n;
    G1Affine *P, *Q, *R;
    uint32_t* scratch;
    unsigned int* d_bad;
    CK(hipMalloc(&P, 2 * n * sizeof(G1Affine)));  // P and Q adjacent: the XYZZ stream runs over both
    Q = P + n;
    CK(hipMalloc(&R, n * sizeof(G1Affine)));
    CK(hipMalloc(&scratch, n * 9 * sizeof(uint32_t)));
    CK(hipMalloc(&d_bad, 4));
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0, 0, P, 2 * n, 0x243F6A8885A308D3ull);
    CK(hipDeviceSynchronize());
    Fq thirty_two = Fq::zero();
    thirty_two.l[0] = 32;
    const Fq one_words = to_mont(thirty_two);  // the L-form of 1

    uint32_t* idx;
    CK(hipMalloc(&idx, 2 * n * sizeof(uint32_t)));
    hipLaunchKernelGGL(k_fill_idx, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, idx, n, (uint32_t)(2 * n), 0x13198A2E03707344ull);
    CK(hipDeviceSynchronize());

    run_batched<32, false>(P, Q, R, scratch, n, one_words, d_bad, idx);
    run_batched<64, false>(P, Q, R, scratch, n, one_words, d_bad, idx);
    run_batched<128, false>(P, Q, R, scratch, n, one_words, d_bad, idx);
    run_batched<256, false>(P, Q, R, scratch, n, one_words, d_bad, idx);
    run_batched<512, false>(P, Q, R, scratch, n, one_words, d_bad, idx);
    run_batched<1024, false>(P, Q, R, scratch, n, one_words, d_bad, idx);
    run_batched<128, true>(P, Q, R, scratch, n, one_words, d_bad, idx);
    run_batched<256, true>(P, Q, R, scratch, n, one_words, d_bad, idx);
    run_batched<512, true>(P, Q, R, scratch, n, one_words, d_bad, idx);

    // XYZZ chains over the same 2n points: chain lengths as in the MSM (the average bucket of the 2^26-term MSM holds 116 points)
    G1XyzzL* out;
    CK(hipMalloc(&out, (2 * n / 32) * sizeof(G1XyzzL)));
    for (int len : {116, 512}) {
        const size_t threads = 2 * n / len;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const unsigned grid = (unsigned)((threads + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(k_xyzz_stream, dim3(grid), dim3(kBlock), 0, 0, (const G1Affine*)P, threads, len, out, one_words);
        CK(hipDeviceSynchronize());
        const int reps = 3;
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_xyzz_stream, dim3(grid), dim3(kBlock), 0, 0, (const G1Affine*)P, threads, len, out, one_words);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        hipFuncAttributes fa;
        CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_xyzz_stream)));
        const double adds = (double)threads * len;
        printf("{\"what\": \"xyzz chains, coalesced stream\", \"points\": %.0f, \"chain\": %d, \"lanes\": %zu, \"ms\": %.3f, \"G_adds_per_s\": %.2f, \"TB_per_s\": %.2f, \"vgprs\": %d, "
               "\"scratch_bytes_per_thread\": %zu}\n",
               adds, len, threads, ms, adds / ms * 1e-6, adds * 64 / ms * 1e-9, fa.numRegs, (size_t)fa.localSizeBytes);
        fflush(stdout);
    }
    return 0;
}
