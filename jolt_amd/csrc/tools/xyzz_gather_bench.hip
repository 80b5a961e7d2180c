// jolt_amd/csrc/tools/xyzz_gather_bench.hip -- measurement tool (not part of the shipped library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 xyzz_gather_bench.hip -o build/xyzz_gather_bench && build/xyzz_gather_bench
// What does the index gather cost the bucket sums?  The production loop (sum_bucket_points_lform: one lane per bucket, the next index and point in
// flight during the addition) against the same loop over a coalesced stream, for table sizes from cache-resident to the 44 GiB of the window
// tables, with one or two points in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

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
__global__ void k_fill_idx(uint32_t* idx, size_t n, uint64_t table, uint64_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t r = ((uint64_t)mix(seed + 2 * i) << 32) | mix(seed + 2 * i + 1);
    idx[i] = (uint32_t)(r % table);
}
__device__ __forceinline__ G1Affine ld_aff(const G1Affine* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    G1Affine r;
    r.x.l[0] = a.x; r.x.l[1] = a.y; r.x.l[2] = a.z; r.x.l[3] = a.w; r.x.l[4] = b.x; r.x.l[5] = b.y; r.x.l[6] = b.z; r.x.l[7] = b.w;
    r.y.l[0] = c.x; r.y.l[1] = c.y; r.y.l[2] = c.z; r.y.l[3] = c.w; r.y.l[4] = d.x; r.y.l[5] = d.y; r.y.l[6] = d.z; r.y.l[7] = d.w;
    return r;
}

// lane t sums the points idx[t * len .. (t + 1) * len) (a bucket's contiguous list, as in the MSM); DEPTH points in flight; MODE 0 = gather by index,
// 1 = coalesced stream (element j * threads + t, no index)
template <int DEPTH, int MODE>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_chain(const uint32_t* __restrict__ idx, const G1Affine* __restrict__ table, size_t threads,
                                                                                         int len, G1XyzzL* __restrict__ out, Fq one_words) {
    const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= threads) return;
    const FqL one = fql_from_words(one_words);
    const uint32_t* src = idx + t * (size_t)len;
    G1XyzzL acc = g1xl_identity();
    G1Affine q[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) q[d] = MODE ? ld_aff(table + (size_t)d * threads + t) : ld_aff(table + src[d]);
    for (int j = 0; j < len; j += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const G1Affine cur = q[d];
            const int jn = j + d + DEPTH;
            if (jn < len) q[d] = MODE ? ld_aff(table + (size_t)jn * threads + t) : ld_aff(table + src[jn]);
            if (j + d < len) acc = g1xl_add_mixed(acc, fql_from_words(cur.x), fql_from_words(cur.y), one);
        }
    }
    out[t] = acc;
}

// rotating queue: ONE addition in the loop body, two points in flight, the index of the next point load fetched an iteration ahead (the point
// load never waits for an index load issued in the same iteration)
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_chain_rot(const uint32_t* __restrict__ idx, const G1Affine* __restrict__ table, size_t threads,
                                                                                             int len, G1XyzzL* __restrict__ out, Fq one_words) {
    const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= threads) return;
    const FqL one = fql_from_words(one_words);
    const uint32_t* src = idx + t * (size_t)len;
    G1XyzzL acc = g1xl_identity();
    G1Affine q0 = ld_aff(table + src[0]);
    G1Affine q1 = len > 1 ? ld_aff(table + src[1]) : q0;
    uint32_t v_ahead = len > 2 ? src[2] : 0;
    for (int j = 0; j < len; ++j) {
        const G1Affine cur = q0;
        q0 = q1;
        if (j + 2 < len) q1 = ld_aff(table + v_ahead);
        if (j + 3 < len) v_ahead = src[j + 3];
        acc = g1xl_add_mixed(acc, fql_from_words(cur.x), fql_from_words(cur.y), one);
    }
    out[t] = acc;
}

// two accumulators per lane over ONE index list: table[idx] into the first, table[idx + 1] into the second (two MSMs over the same scalars and shifted bases:
// the witness commitments at r and -r of an opening) -- one index load and one 128-byte gather for two additions
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_chain_pair(const uint32_t* __restrict__ idx, const G1Affine* __restrict__ table, size_t threads,
                                                                                              int len, G1XyzzL* __restrict__ out, Fq one_words) {
    const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= threads) return;
    const FqL one = fql_from_words(one_words);
    const uint32_t* src = idx + t * (size_t)len;
    G1XyzzL acc0 = g1xl_identity(), acc1 = g1xl_identity();
    G1Affine a = ld_aff(table + src[0]), b = ld_aff(table + src[0] + 1);
    for (int j = 0; j < len; ++j) {
        const G1Affine ca = a, cb = b;
        if (j + 1 < len) {
            const uint32_t v = src[j + 1];
            a = ld_aff(table + v);
            b = ld_aff(table + v + 1);
        }
        acc0 = g1xl_add_mixed(acc0, fql_from_words(ca.x), fql_from_words(ca.y), one);
        acc1 = g1xl_add_mixed(acc1, fql_from_words(cb.x), fql_from_words(cb.y), one);
    }
    out[2 * t] = acc0;
    out[2 * t + 1] = acc1;
}

template <int DEPTH, int MODE>
void run(const char* what, const uint32_t* idx, const G1Affine* table, size_t threads, int len, G1XyzzL* out, Fq one_words, double table_gib) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((threads + kBlock - 1) / kBlock);
    auto launch = [&]() {
        if (DEPTH == 0) hipLaunchKernelGGL(k_chain_rot, dim3(grid), dim3(kBlock), 0, 0, idx, table, threads, len, out, one_words);
        else hipLaunchKernelGGL((k_chain<(DEPTH ? DEPTH : 1), MODE>), dim3(grid), dim3(kBlock), 0, 0, idx, table, threads, len, out, one_words);
    };
    launch();
    CK(hipDeviceSynchronize());
    const int reps = 3;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    hipFuncAttributes fa;
    if (DEPTH == 0) CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_chain_rot)));
    else CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_chain<(DEPTH ? DEPTH : 1), MODE>)));
    const double adds = (double)threads * len;
    printf("{\"what\": \"%s\", \"table_GiB\": %.2f, \"in_flight\": %d, \"chain\": %d, \"lanes\": %zu, \"ms\": %.3f, \"G_adds_per_s\": %.2f, \"vgprs\": %d, \"scratch_bytes_per_thread\": %zu}\n", what,
           table_gib, DEPTH, len, threads, ms, adds / ms * 1e-6, fa.numRegs, (size_t)fa.localSizeBytes);
    fflush(stdout);
}
}  // namespace

int main() {
    const size_t n_idx = (size_t)1 << 28;  // additions per launch
    const int len = 116;
    const size_t threads = n_idx / len;
    const size_t max_table = (size_t)11 << 26;  // the window tables of a 2^26-point SRS: 44 GiB
    G1Affine* table;
    uint32_t* idx;
    G1XyzzL* out;
    CK(hipMalloc(&table, max_table * sizeof(G1Affine)));
    CK(hipMalloc(&idx, n_idx * sizeof(uint32_t)));
    CK(hipMalloc(&out, threads * sizeof(G1XyzzL)));
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((max_table + 255) / 256)), dim3(256), 0, 0, table, max_table, 0x243F6A8885A308D3ull);
    CK(hipDeviceSynchronize());
    Fq thirty_two = Fq::zero();
    thirty_two.l[0] = 32;
    const Fq one_words = to_mont(thirty_two);
    run<1, 1>("coalesced stream", idx, table, threads, len, out, one_words, n_idx * 64.0 / (1 << 30));
    run<2, 1>("coalesced stream", idx, table, threads, len, out, one_words, n_idx * 64.0 / (1 << 30));
    for (size_t tab : {(size_t)1 << 20, (size_t)1 << 24, max_table}) {
        hipLaunchKernelGGL(k_fill_idx, dim3((unsigned)((n_idx + 255) / 256)), dim3(256), 0, 0, idx, n_idx, (uint64_t)tab, 0x13198A2E03707344ull + tab);
        CK(hipDeviceSynchronize());
        run<1, 0>("gather by index", idx, table, threads, len, out, one_words, tab * 64.0 / (1 << 30));
        run<2, 0>("gather by index", idx, table, threads, len, out, one_words, tab * 64.0 / (1 << 30));
        run<0, 0>("gather by index, rotating queue (in_flight 0 = two points + index ahead)", idx, table, threads, len, out, one_words, tab * 64.0 / (1 << 30));
    }
    {
        const size_t tab = max_table - 1;
        hipLaunchKernelGGL(k_fill_idx, dim3((unsigned)((n_idx + 255) / 256)), dim3(256), 0, 0, idx, n_idx, (uint64_t)tab, 0x5555ull);
        CK(hipDeviceSynchronize());
        G1XyzzL* out2;
        const size_t th = threads / 2;  // the same number of additions per launch as the other runs
        CK(hipMalloc(&out2, 2 * th * sizeof(G1XyzzL)));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const unsigned grid = (unsigned)((th + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(k_chain_pair, dim3(grid), dim3(kBlock), 0, 0, (const uint32_t*)idx, (const G1Affine*)table, th, len, out2, one_words);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_chain_pair, dim3(grid), dim3(kBlock), 0, 0, (const uint32_t*)idx, (const G1Affine*)table, th, len, out2, one_words);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 3;
        hipFuncAttributes fa;
        CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_chain_pair)));
        const double adds = 2.0 * th * len;
        printf("{\"what\": \"gather by index, two accumulators over one list (table[i], table[i + 1])\", \"table_GiB\": %.2f, \"chain\": %d, \"lanes\": %zu, \"ms\": %.3f, \"G_adds_per_s\": %.2f, \"vgprs\": %d, \"scratch_bytes_per_thread\": %zu}\n",
               tab * 64.0 / (1 << 30), len, th, ms, adds / ms * 1e-6, fa.numRegs, (size_t)fa.localSizeBytes);
    }
    return 0;
}
