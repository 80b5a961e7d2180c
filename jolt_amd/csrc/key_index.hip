// jolt_amd/csrc/key_index.hip -- pushforwards of cycle weights onto a LARGE address domain (bytecode PCs, RAM words): the T-scale half of the
// joint-domain relations whose rounds run over K-sized tables (last row).
//
//   out_s[k] = sum_{j : key(j) = k} w_s[j]        for S weight tables over the cycles at once
//
// is what the reference computes in
//   * stage_pushforwards (crates/jolt-kernels/src/optimized/bytecode_read_raf.rs:152-237): the five per-stage tables F_s(k) = sum_{j: pc(j) = k}
//     eq(r_cycle_s, j) of the bytecode read+RAF address phase, all stages in one trace walk (its split-eq two-table form is an evaluation
//     order: the sums are those of the full eq tables, which its own test asserts);
//   * RamAccessColumns::fold_cycles (optimized/ram_trace.rs:150-162): ra_folded(k) = sum_{j: address(j) = k} eq(tau_low, j) of RAM RAF
//     evaluation (optimized/ram_raf_evaluation.rs:44-48).
// jolt_onehot_pushforward (onehot.hip) serves K <= 256 chunk domains with per-lane bins; here K is 2^10 .. 2^24 and field elements have no
// atomics, so the rows are SORTED BY KEY once per trace column (jolt_key_index: what the reference shares through its ProofSession as the
// packed PC rows / RamAccessColumns) and every pushforward is a segmented sum over that order:
//   index   : counting sort through LDS (the MSM's k_msm_hist_lds / k_msm_scatter_lds, 32768 bins per pass; keys outside a pass's group of
//             32768 addresses are masked out, K / 32768 passes), bin offsets into `start`;
//   sum     : a bin's rows are cut into work items of <= 1024 rows (skew: a loop body's PCs hold most cycles of a trace); one wavefront per
//             item adds w_s[row] per lane and folds the lanes by shuffles, for each s; one thread per (s, bin) adds the items of a bin.
// No atomics on field elements; results do not depend on the order the sort happened to produce (field addition is exact).
#include <algorithm>
#include <vector>

#include "ctx.hpp"
#include "ints.hpp"
#include "msm_kernels.hip.h"
#include "poly_kernels.hip.h"

using namespace jolt;
using namespace jolt::msmk;

int32_t jolt_internal_table_new(jolt_ctx* ctx, size_t len, jolt_table** out);

struct jolt_key_index {
    jolt_ctx* ctx = nullptr;
    size_t cycles = 0;
    uint64_t K = 0;
    uint32_t* sorted = nullptr;     // row ids grouped by key (rows with a key >= K -- cold cycles -- are absent)
    uint32_t* start = nullptr;      // [K + 1]: bin k owns sorted[start[k] .. start[k + 1])
    uint32_t* item_start = nullptr; // [K + 1]: first work item of bin k; item_start[K] = number of items
    uint32_t n_items = 0;
};

namespace {
constexpr uint32_t kGroupBits = 15, kGroup = 1u << kGroupBits;  // addresses per sort pass: kGroup + 1 counters of 4 bytes fit the LDS of a CU
constexpr uint32_t kItemRows = 1024;
constexpr uint32_t kMaxWeights = 8;

// keys32[j] = (key(j) in this pass's group) ? low bits + 1 : 0   (0 = "no digit": the counting sort skips it)
__global__ __launch_bounds__(kBlock) void k_ki_keys(const uint64_t* __restrict__ keys, size_t cycles, uint64_t K, uint64_t group, uint32_t* __restrict__ keys32) {
    const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= cycles) return;
    const uint64_t k = keys[j];
    keys32[j] = (k < K && (k >> kGroupBits) == group) ? (uint32_t)(k & (kGroup - 1)) + 1u : 0u;
}
// One workgroup: exclusive scan of a pass's histogram on top of the rows placed by the earlier passes (*base): cursor[b] for the scatter,
// start[group * kGroup + b - 1] for the index; *base moves on by the pass's row count.
__global__ __launch_bounds__(1024) void k_ki_offsets(const uint32_t* __restrict__ hist, uint32_t bins /* keys 1 .. bins */, uint32_t* __restrict__ cursor,
                                                     uint32_t* __restrict__ start, uint32_t* __restrict__ base) {
    __shared__ uint32_t sm[1024];
    const uint32_t b0 = *base;
    const uint32_t per = (bins + 1023) / 1024, lo = min(threadIdx.x * per, bins), hi = min(lo + per, bins);
    uint32_t local = 0;
    for (uint32_t b = lo; b < hi; ++b) local += hist[b + 1];
    sm[threadIdx.x] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t v = (int)threadIdx.x >= off ? sm[threadIdx.x - off] : 0;
        __syncthreads();
        sm[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = b0 + sm[threadIdx.x] - local;
    for (uint32_t b = lo; b < hi; ++b) {
        cursor[b + 1] = run;
        start[b] = run;
        run += hist[b + 1];
    }
    __syncthreads();  // every thread has read *base
    if (threadIdx.x == 1023) *base = b0 + sm[1023];
}
// start[K] = total rows; item_start = exclusive scan of ceil(rows / kItemRows) over the K bins (grid-stride over bins in ONE workgroup: K <= 2^24)
__global__ __launch_bounds__(1024) void k_ki_items(uint32_t* __restrict__ start, const uint32_t* __restrict__ base, uint64_t K, uint32_t* __restrict__ item_start) {
    __shared__ uint32_t sm[1024];
    if (threadIdx.x == 0) start[K] = *base;
    __syncthreads();
    const uint64_t per = (K + 1023) / 1024, lo = min((uint64_t)threadIdx.x * per, K), hi = min(lo + per, K);
    uint32_t local = 0;
    for (uint64_t b = lo; b < hi; ++b) local += (start[b + 1] - start[b] + kItemRows - 1) / kItemRows;
    sm[threadIdx.x] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t v = (int)threadIdx.x >= off ? sm[threadIdx.x - off] : 0;
        __syncthreads();
        sm[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = sm[threadIdx.x] - local;
    for (uint64_t b = lo; b < hi; ++b) {
        item_start[b] = run;
        run += (start[b + 1] - start[b] + kItemRows - 1) / kItemRows;
    }
    if (threadIdx.x == 1023) item_start[K] = sm[1023];
}

__device__ __forceinline__ Fr wave_sum_fr(Fr v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        Fr o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o.l[k] = __shfl_xor(v.l[k], off, 64);
        v = add(v, o);
    }
    return v;
}
struct WeightPtrs {
    const Fr* w[kMaxWeights];
};
// part[item * S + s] = sum of w_s over the item's rows
__global__ __launch_bounds__(kBlock) void k_ki_accumulate(const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ start, const uint32_t* __restrict__ item_start,
                                                          uint64_t K, WeightPtrs wp, uint32_t S, Fr* __restrict__ part) {
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, n_waves = gridDim.x * (kBlock / 64);
    const uint32_t n_items = item_start[K];
    for (uint32_t item = wave; item < n_items; item += n_waves) {
        uint64_t b_lo = 0, b_hi = K;  // the bin with item_start[bin] <= item < item_start[bin + 1] (empty bins own no item)
        while (b_hi - b_lo > 1) {
            const uint64_t mid = (b_lo + b_hi) >> 1;
            if (item_start[mid] <= item) b_lo = mid; else b_hi = mid;
        }
        const uint32_t first = start[b_lo] + (item - item_start[b_lo]) * kItemRows;
        const uint32_t cnt = min(start[b_lo + 1] - first, kItemRows);
        for (uint32_t s = 0; s < S; ++s) {
            const Fr* __restrict__ w = wp.w[s];
            Fr acc = Fr::zero();
            for (uint32_t k = lane; k < cnt; k += 64) acc = add(acc, ld_fr(w + sorted[first + k]));
            acc = wave_sum_fr(acc);
            if (lane == 0) st_fr(part + (size_t)item * S + s, acc);
        }
    }
}
struct OutPtrs {
    Fr* o[kMaxWeights];
};
__global__ __launch_bounds__(kBlock) void k_ki_fold(const Fr* __restrict__ part, const uint32_t* __restrict__ item_start, uint64_t K, uint32_t S, OutPtrs op) {
    const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t bin = t / S;
    const uint32_t s = (uint32_t)(t % S);
    if (bin >= K) return;
    Fr acc = Fr::zero();
    for (uint32_t item = item_start[bin]; item < item_start[bin + 1]; ++item) acc = add(acc, ld_fr(part + (size_t)item * S + s));
    st_fr(op.o[s] + bin, acc);
}
// last[k] = values[the LATEST row of bin k] as a field element, init[k] for a bin without rows (the final state of a word that is only ever
// overwritten: RamAccessColumns / ram_val_final of the witness oracle)
__global__ __launch_bounds__(kBlock) void k_ki_last_value(const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ start, uint64_t K, const uint64_t* __restrict__ values,
                                                          const Fr* __restrict__ init, Fr* __restrict__ out) {
    const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= K) return;
    const uint32_t lo = start[k], hi = start[k + 1];
    if (lo == hi) { st_fr(out + k, ld_fr(init + k)); return; }
    uint32_t latest = sorted[lo];
    for (uint32_t i = lo + 1; i < hi; ++i) latest = max(latest, sorted[i]);
    Fr v = Fr::zero();
    const uint64_t x = values[latest];
    v.l[0] = (uint32_t)x;
    v.l[1] = (uint32_t)(x >> 32);
    st_fr(out + k, to_mont(v));
}
}  // namespace

extern "C" int32_t jolt_key_index_destroy(jolt_ctx* ctx, jolt_key_index* ix) {
    if (!ix) return JOLT_OK;
    if (!ctx) ctx = ix->ctx;
    if (ix->sorted) jolt_internal_dev_free(ctx, ix->sorted);
    if (ix->start) jolt_internal_dev_free(ctx, ix->start);
    if (ix->item_start) jolt_internal_dev_free(ctx, ix->item_start);
    delete ix;
    return JOLT_OK;
}

extern "C" int32_t jolt_key_index_create(jolt_ctx* ctx, const jolt_ints* keys, uint64_t K, jolt_key_index** out) {
    if (!ctx || !keys || !out) return JOLT_ERR_INVALID_ARG;
    if (keys->kind != JOLT_INT_U64) return JOLT_ERR_INVALID_ARG;
    const size_t T = keys->count;
    if (T == 0 || T >= ((size_t)1 << 31)) return JOLT_ERR_SIZE_MISMATCH;
    if (K == 0 || K > ((uint64_t)1 << 24)) return JOLT_ERR_UNSUPPORTED;
    const size_t lds = ((size_t)kGroup + 1) * 4;
    if (lds > ctx->max_lds_per_block) return JOLT_ERR_UNSUPPORTED;
    jolt_key_index* ix = new (std::nothrow) jolt_key_index();
    if (!ix) return JOLT_ERR_OOM;
    ix->ctx = ctx;
    ix->cycles = T;
    ix->K = K;
    uint32_t *keys32 = nullptr, *hist = nullptr, *cursor = nullptr, *base = nullptr;
    int32_t rc = jolt_internal_dev_alloc(ctx, T * 4, (void**)&ix->sorted);
    if (rc == JOLT_OK) rc = jolt_internal_dev_alloc(ctx, (K + 1) * 4, (void**)&ix->start);
    if (rc == JOLT_OK) rc = jolt_internal_dev_alloc(ctx, (K + 1) * 4, (void**)&ix->item_start);
    if (rc == JOLT_OK) rc = jolt_internal_dev_alloc(ctx, T * 4, (void**)&keys32);
    if (rc == JOLT_OK) rc = jolt_internal_dev_alloc(ctx, lds, (void**)&hist);
    if (rc == JOLT_OK) rc = jolt_internal_dev_alloc(ctx, lds, (void**)&cursor);
    if (rc == JOLT_OK) rc = jolt_internal_dev_alloc(ctx, 256, (void**)&base);
    hipStream_t st = ctx->stream;
    hipError_t e = hipSuccess;
    if (rc == JOLT_OK) {
        (void)hipFuncSetAttribute((const void*)k_msm_hist_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->max_lds_per_block);
        (void)hipFuncSetAttribute((const void*)k_msm_scatter_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->max_lds_per_block);
        (void)hipGetLastError();
        e = hipMemsetAsync(base, 0, 4, st);
        const uint64_t groups = (K + kGroup - 1) >> kGroupBits;
        const unsigned slices = (unsigned)std::max<size_t>(1, std::min<size_t>((size_t)ctx->num_cus, T / 16384 + 1));
        for (uint64_t g = 0; g < groups && e == hipSuccess; ++g) {
            const uint32_t bins = (uint32_t)std::min<uint64_t>(kGroup, K - (g << kGroupBits));
            e = hipMemsetAsync(hist, 0, ((size_t)bins + 1) * 4, st);
            if (e != hipSuccess) break;
            hipLaunchKernelGGL(k_ki_keys, dim3((unsigned)((T + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, (const uint64_t*)keys->data, T, K, g, keys32);
            hipLaunchKernelGGL(k_msm_hist_lds, dim3(slices, 1), dim3(kSortBlock), ((size_t)bins + 1) * 4, st, (const uint32_t*)keys32, T, bins, hist);
            hipLaunchKernelGGL(k_ki_offsets, dim3(1), dim3(1024), 0, st, (const uint32_t*)hist, bins, cursor, ix->start + (g << kGroupBits), base);
            hipLaunchKernelGGL(k_msm_scatter_lds, dim3(slices, 1), dim3(kSortBlock), ((size_t)bins + 1) * 4, st, (const uint32_t*)keys32, T, bins, cursor, ix->sorted);
            e = hipGetLastError();
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_ki_items, dim3(1), dim3(1024), 0, st, ix->start, (const uint32_t*)base, K, ix->item_start);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&ix->n_items, ix->item_start + K, 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);  // the item count sizes every later pushforward; the scratch below goes back to the pool
    }
    if (keys32) jolt_internal_dev_free(ctx, keys32);
    if (hist) jolt_internal_dev_free(ctx, hist);
    if (cursor) jolt_internal_dev_free(ctx, cursor);
    if (base) jolt_internal_dev_free(ctx, base);
    if (rc != JOLT_OK || e != hipSuccess) {
        if (e != hipSuccess) { (void)hipGetLastError(); ctx->last_error = std::string("key index: ") + hipGetErrorString(e); }
        (void)jolt_key_index_destroy(ctx, ix);
        return rc != JOLT_OK ? rc : (e == hipErrorOutOfMemory ? JOLT_ERR_OOM : JOLT_ERR_HIP);
    }
    *out = ix;
    return JOLT_OK;
}

extern "C" int32_t jolt_key_index_size(const jolt_key_index* ix, size_t* cycles, uint64_t* K, uint32_t* items) {
    if (!ix) return JOLT_ERR_INVALID_ARG;
    if (cycles) *cycles = ix->cycles;
    if (K) *K = ix->K;
    if (items) *items = ix->n_items;
    return JOLT_OK;
}

extern "C" int32_t jolt_key_index_pushforward(jolt_ctx* ctx, const jolt_key_index* ix, jolt_table* const* weights, size_t n_weights, jolt_table** out) {
    if (!ctx || !ix || !weights || !out || n_weights == 0) return JOLT_ERR_INVALID_ARG;
    if (n_weights > kMaxWeights) return JOLT_ERR_UNSUPPORTED;
    for (size_t s = 0; s < n_weights; ++s) {
        if (!weights[s]) return JOLT_ERR_INVALID_ARG;
        if (weights[s]->len != ix->cycles) return JOLT_ERR_SIZE_MISMATCH;
    }
    const uint32_t S = (uint32_t)n_weights;
    WeightPtrs wp{};
    OutPtrs op{};
    std::vector<jolt_table*> made;
    auto fail = [&](int32_t rc) {
        for (jolt_table* t : made) jolt_table_free(ctx, t);
        return rc;
    };
    for (uint32_t s = 0; s < S; ++s) {
        jolt_table* t = nullptr;
        const int32_t rc = jolt_internal_table_new(ctx, ix->K, &t);
        if (rc != JOLT_OK) return fail(rc);
        made.push_back(t);
        wp.w[s] = weights[s]->data();
        op.o[s] = t->data();
    }
    Fr* part = nullptr;
    const size_t items = std::max<size_t>(ix->n_items, 1);
    const int32_t rc = jolt_internal_dev_alloc(ctx, items * S * sizeof(Fr), (void**)&part);
    if (rc != JOLT_OK) return fail(rc);
    hipStream_t st = ctx->stream;
    const unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>((items + 3) / 4, (size_t)ctx->num_cus * 16));
    hipLaunchKernelGGL(k_ki_accumulate, dim3(grid), dim3(kBlock), 0, st, (const uint32_t*)ix->sorted, (const uint32_t*)ix->start, (const uint32_t*)ix->item_start, ix->K, wp, S, part);
    hipLaunchKernelGGL(k_ki_fold, dim3((unsigned)((ix->K * S + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, (const Fr*)part, (const uint32_t*)ix->item_start, ix->K, S, op);
    const hipError_t e = hipGetLastError();
    jolt_internal_dev_free(ctx, part);  // stream-ordered pool: the block is not handed out again before the kernels above ran
    if (e != hipSuccess) { ctx->last_error = std::string("key index pushforward: ") + hipGetErrorString(e); return fail(JOLT_ERR_HIP); }
    for (uint32_t s = 0; s < S; ++s) out[s] = made[s];
    return JOLT_OK;
}

extern "C" int32_t jolt_key_index_last_value(jolt_ctx* ctx, const jolt_key_index* ix, const jolt_ints* values, const jolt_table* init, jolt_table** out) {
    if (!ctx || !ix || !values || !init || !out) return JOLT_ERR_INVALID_ARG;
    if (values->kind != JOLT_INT_U64) return JOLT_ERR_INVALID_ARG;
    if (values->count != ix->cycles || init->len != ix->K) return JOLT_ERR_SIZE_MISMATCH;
    jolt_table* t = nullptr;
    JOLT_TRY(jolt_internal_table_new(ctx, ix->K, &t));
    hipLaunchKernelGGL(k_ki_last_value, dim3((unsigned)((ix->K + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, (const uint32_t*)ix->sorted, (const uint32_t*)ix->start, ix->K,
                       (const uint64_t*)values->data, (const Fr*)init->data(), t->data());
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ctx->last_error = std::string("key index last value: ") + hipGetErrorString(e); jolt_table_free(ctx, t); return JOLT_ERR_HIP; }
    *out = t;
    return JOLT_OK;
}
