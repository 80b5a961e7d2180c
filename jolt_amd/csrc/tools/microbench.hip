// jolt_amd/csrc/tools/microbench.hip -- measurement tool (not part of the shipped library).
// Run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 microbench.hip -o microbench && ./microbench
// Measures: HBM copy bandwidth, v_mad_u64_u32 issue rate, register-resident Fr mul rates (full / shifted challenge),
// and the LowToHigh bind kernel variants, so that design choices in DESIGN.md are backed by numbers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../field.hip.h"

using namespace jolt;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

// dependent-free mad throughput: 8 independent accumulators, ITER x 8 mads per thread
__global__ void k_mad(uint64_t* out, uint32_t a0, uint32_t b0, int iters) {
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

template <int MODE>  // 0 = full mul, 1 = shifted challenge mul, 2 = add only
__global__ void k_frmul(Fr* out, Fr seed, Fr c, int iters) {
    Fr x = seed;
    x.l[0] ^= threadIdx.x + blockIdx.x * 977;
    Fr y = x;
    y.l[1] ^= 0x1234;
    uint32_t chi[4] = {c.l[4], c.l[5], c.l[6], c.l[7]};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { x = mul(x, c); y = mul(y, c); }
        if (MODE == 1) { x = mul_shifted(x, chi); y = mul_shifted(y, chi); }
        if (MODE == 2) { x = add(x, c); y = sub(y, c); }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = add(x, y);
}

// multiply throughput against occupancy and instruction-level parallelism: CHAINS independent multiply chains per thread,
// occupancy limited through the dynamic LDS request (blocks per CU = 160 KB / lds bytes)
template <int CHAINS>
__global__ __launch_bounds__(256) void k_frmul_ilp(Fr* out, Fr seed, Fr c, int iters) {
    extern __shared__ unsigned char lds_pad[];
    Fr x[CHAINS];
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) {
        x[k] = seed;
        x[k].l[0] ^= threadIdx.x + blockIdx.x * 977 + k * 131;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < CHAINS; ++k) x[k] = mul(x[k], c);
    }
    Fr s = x[0];
#pragma unroll
    for (int k = 1; k < CHAINS; ++k) s = add(s, x[k]);
    if (lds_pad[threadIdx.x] == 77) s.l[0] ^= 1;  // keep the LDS request alive
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// code-size sensitivity: the loop body holds N fully inlined multiplies (~3.7 KB of code each); the instruction cache is 64 KB
// per CU pair, so bodies beyond ~16 multiplies no longer fit
template <int N>
__global__ __launch_bounds__(256) void k_frmul_body(Fr* out, Fr seed, Fr c, int iters) {
    Fr x = seed, y = c;
    x.l[0] ^= threadIdx.x + blockIdx.x * 977;
    y.l[1] ^= threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            x = mul(x, y);
            y.l[0] ^= k + 1;  // keeps the N multiplies distinct instruction sequences
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = add(x, y);
}
__device__ __noinline__ Fr mul_call(const Fr& a, const Fr& b) { return mul(a, b); }
template <int N>
__global__ __launch_bounds__(256) void k_frmul_body_call(Fr* out, Fr seed, Fr c, int iters) {
    Fr x = seed, y = c;
    x.l[0] ^= threadIdx.x + blockIdx.x * 977;
    y.l[1] ^= threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            x = mul_call(x, y);
            y.l[0] ^= k + 1;
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = add(x, y);
}

// ---- bind variants -----------------------------------------------------------------------------------------
template <bool SHIFTED>
__device__ __forceinline__ Fr bind_one(const Fr& lo, const Fr& hi, const Fr& r) {
    Fr d = sub(hi, lo);
    Fr m;
    if (SHIFTED) { uint32_t chi[4] = {r.l[4], r.l[5], r.l[6], r.l[7]}; m = mul_shifted(d, chi); }
    else m = mul(d, r);
    return add(lo, m);
}
__device__ __forceinline__ Fr load_fr(const uint4* p) {
    uint4 a = p[0], b = p[1];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
__device__ __forceinline__ void store_fr(uint4* p, const Fr& r) {
    p[0] = make_uint4(r.l[0], r.l[1], r.l[2], r.l[3]);
    p[1] = make_uint4(r.l[4], r.l[5], r.l[6], r.l[7]);
}

// A: one output per thread, direct 64-byte pair load per lane
template <bool SHIFTED, int MULS>
__global__ __launch_bounds__(256) void k_bind_a(const uint4* __restrict__ in, uint4* __restrict__ out, size_t half, Fr r) {
    size_t y = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; y < half; y += stride) {
        Fr lo = load_fr(in + 4 * y), hi = load_fr(in + 4 * y + 2);
        Fr o = bind_one<SHIFTED>(lo, hi, r);
        if (MULS == 0) o = add(lo, hi);
        store_fr(out + 2 * y, o);
    }
}

// B: coalesced 16-byte-per-lane loads of a 4 KiB chunk per wave + 4x4 transpose inside lane quads (DPP)
__device__ __forceinline__ uint32_t dpp_xor1(uint32_t v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }  // quad_perm [1,0,3,2]
__device__ __forceinline__ uint32_t dpp_xor2(uint32_t v) { return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true); }  // quad_perm [2,3,0,1]
__device__ __forceinline__ uint4 sel4(bool c, uint4 a, uint4 b) { return c ? a : b; }
__device__ __forceinline__ uint4 x1(uint4 v) { return make_uint4(dpp_xor1(v.x), dpp_xor1(v.y), dpp_xor1(v.z), dpp_xor1(v.w)); }
__device__ __forceinline__ uint4 x2(uint4 v) { return make_uint4(dpp_xor2(v.x), dpp_xor2(v.y), dpp_xor2(v.z), dpp_xor2(v.w)); }

template <bool SHIFTED>
__global__ __launch_bounds__(256) void k_bind_b(const uint4* __restrict__ in, uint4* __restrict__ out, size_t half, Fr r) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 3;
    size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    size_t nchunks = half >> 6;  // 64 pairs per chunk
    for (size_t ch = wave; ch < nchunks; ch += nwaves) {
        const uint4* base = in + ch * 256;  // 4 KiB = 256 uint4
        uint4 R0 = base[lane], R1 = base[64 + lane], R2 = base[128 + lane], R3 = base[192 + lane];
        // 4x4 transpose of (R0..R3) across the quad: after it lane j holds load j's four quarters
        bool odd = j & 1;
        uint4 a0 = sel4(odd, x1(R1), R0), a1 = sel4(odd, R1, x1(R0));
        uint4 a2 = sel4(odd, x1(R3), R2), a3 = sel4(odd, R3, x1(R2));
        bool hi2 = j & 2;
        uint4 b0 = sel4(hi2, x2(a2), a0), b2 = sel4(hi2, a2, x2(a0));
        uint4 b1 = sel4(hi2, x2(a3), a1), b3 = sel4(hi2, a3, x2(a1));
        Fr lo, hi;
        lo.l[0] = b0.x; lo.l[1] = b0.y; lo.l[2] = b0.z; lo.l[3] = b0.w; lo.l[4] = b1.x; lo.l[5] = b1.y; lo.l[6] = b1.z; lo.l[7] = b1.w;
        hi.l[0] = b2.x; hi.l[1] = b2.y; hi.l[2] = b2.z; hi.l[3] = b2.w; hi.l[4] = b3.x; hi.l[5] = b3.y; hi.l[6] = b3.z; hi.l[7] = b3.w;
        Fr o = bind_one<SHIFTED>(lo, hi, r);
        size_t oidx = ch * 64 + 16 * j + (lane >> 2);
        store_fr(out + 2 * oidx, o);
    }
}

static double time_ms(hipStream_t s, int reps, const std::function<void()>& f) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

#include <functional>

int main(int argc, char** argv) {
    int logn = argc > 1 ? atoi(argv[1]) : 24;
    size_t n = (size_t)1 << logn;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    uint4 *in, *out;
    CK(hipMalloc(&in, n * 32));
    CK(hipMalloc(&out, n * 32));
    // fill with canonical-looking random limbs (top limb < 0x30000000 keeps values < p)
    std::vector<uint32_t> h(n * 8);
    uint64_t st = 88172645463325252ull;
    for (size_t i = 0; i < n * 8; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; h[i] = (uint32_t)st; if ((i & 7) == 7) h[i] &= 0x1FFFFFFF; }
    CK(hipMemcpy(in, h.data(), n * 32, hipMemcpyHostToDevice));
    Fr r; for (int i = 0; i < 8; ++i) r.l[i] = h[i]; r.l[7] &= 0x1FFFFFFF;
    Fr rs = r; rs.l[0] = rs.l[1] = rs.l[2] = rs.l[3] = 0;

    printf("== table 2^%d Fr = %.1f MiB ==\n", logn, n * 32 / 1048576.0);
    for (int blocks : {2048, 8192}) {
        double ms = time_ms(s, 10, [&] { k_copy<<<blocks, 256, 0, s>>>(in, out, n * 2); });
        printf("copy   grid %5d: %.3f ms  %.2f TB/s (r+w)\n", blocks, ms, 2.0 * n * 32 / ms / 1e9);
    }
    {
        uint64_t* o; CK(hipMalloc(&o, (size_t)256 * 32 * 256 * sizeof(Fr)));  // room for the largest grid below (8192 blocks x 256 threads)
        int iters = 4096;
        int blocks = 256 * 8;
        double ms = time_ms(s, 5, [&] { k_mad<<<blocks, 256, 0, s>>>(o, 12345, 67891, iters); });
        double mads = (double)blocks * 256 * iters * 8;
        printf("v_mad_u64_u32: %.3f ms, %.2f Tmad/s  => %.2f cycles/wave-instr/SIMD @2.4GHz\n", ms, mads / ms / 1e9,
               (256.0 * 4 * 2.4e9) / (mads / (ms * 1e-3) / 64));
        Fr* fo = (Fr*)o;
        int it2 = 512;
        double m0 = time_ms(s, 5, [&] { k_frmul<0><<<blocks, 256, 0, s>>>(fo, r, r, it2); });
        double m1 = time_ms(s, 5, [&] { k_frmul<1><<<blocks, 256, 0, s>>>(fo, r, rs, it2); });
        double m2 = time_ms(s, 5, [&] { k_frmul<2><<<blocks, 256, 0, s>>>(fo, r, rs, it2); });
        double muls = (double)blocks * 256 * it2 * 2;
        printf("Fr mul full:    %.3f ms  %.1f Gmul/s\n", m0, muls / m0 / 1e6);
        printf("Fr mul shifted: %.3f ms  %.1f Gmul/s\n", m1, muls / m1 / 1e6);
        printf("Fr add/sub:     %.3f ms  %.1f Gop/s\n", m2, muls / m2 / 1e6);
        // occupancy x ILP matrix (waves per SIMD = blocks per CU for 256-thread blocks)
        for (int wps : {1, 2, 4, 8}) {
            size_t lds = wps == 8 ? 0 : (size_t)(160 * 1024 / wps) - 1024;
            int grid = 256 * wps * 4;
            auto run = [&](auto kern, int chains) {
                (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                double ms = time_ms(s, 3, [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, fo, r, r, 256); });
                printf("  waves/SIMD %d chains %d: %.1f Gmul/s\n", wps, chains, (double)grid * 256 * 256 * chains / ms / 1e6);
            };
            run(k_frmul_ilp<1>, 1);
            run(k_frmul_ilp<2>, 2);
            run(k_frmul_ilp<4>, 4);
        }
        {
            int grid = 256 * 16;
            auto run = [&](auto kern, int nmul, const char* what) {
                int iters = 2048 / nmul;
                double ms = time_ms(s, 3, [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s, fo, r, r, iters); });
                printf("  loop body of %2d %s multiplies: %.1f Gmul/s\n", nmul, what, (double)grid * 256 * iters * nmul / ms / 1e6);
            };
            run(k_frmul_body<2>, 2, "inlined");
            run(k_frmul_body<8>, 8, "inlined");
            run(k_frmul_body<16>, 16, "inlined");
            run(k_frmul_body<32>, 32, "inlined");
            run(k_frmul_body<64>, 64, "inlined");
            run(k_frmul_body_call<2>, 2, "called ");
            run(k_frmul_body_call<16>, 16, "called ");
            run(k_frmul_body_call<64>, 64, "called ");
        }
        CK(hipFree(o));
    }
    size_t half = n / 2;
    double bytes = 48.0 * n;  // algorithmic: 32N read + 16N write
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
        double a0 = time_ms(s, 10, [&] { k_bind_a<false, 1><<<blocks, 256, 0, s>>>(in, out, half, r); });
        double a1 = time_ms(s, 10, [&] { k_bind_a<true, 1><<<blocks, 256, 0, s>>>(in, out, half, rs); });
        double a2 = time_ms(s, 10, [&] { k_bind_a<true, 0><<<blocks, 256, 0, s>>>(in, out, half, rs); });
        double b0 = time_ms(s, 10, [&] { k_bind_b<false><<<blocks, 256, 0, s>>>(in, out, half, r); });
        double b1 = time_ms(s, 10, [&] { k_bind_b<true><<<blocks, 256, 0, s>>>(in, out, half, rs); });
        printf("bind grid %5d: A full %.3f ms %.2f TB/s | A shifted %.3f ms %.2f TB/s | A nomul %.3f ms %.2f TB/s | B full %.3f ms %.2f TB/s | B shifted %.3f ms %.2f TB/s\n",
               blocks, a0, bytes / a0 / 1e9, a1, bytes / a1 / 1e9, a2, bytes / a2 / 1e9, b0, bytes / b0 / 1e9, b1, bytes / b1 / 1e9);
    }
    // correctness cross-check A vs B on the device (host check lives in the pytest suite)
    {
        uint4* out2; CK(hipMalloc(&out2, n * 16));
        k_bind_a<false, 1><<<4096, 256, 0, s>>>(in, out, half, r);
        k_bind_b<false><<<4096, 256, 0, s>>>(in, out2, half, r);
        CK(hipStreamSynchronize(s));
        std::vector<uint32_t> ha(1 << 16), hb(1 << 16);
        CK(hipMemcpy(ha.data(), out, ha.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb.data(), out2, hb.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < ha.size(); ++i) bad += ha[i] != hb[i];
        printf("A vs B mismatching words in first 256 KiB: %zu\n", bad);
    }
    return 0;
}
