// jolt_amd/csrc/onehot.hip -- the hot-index source of one-hot selector columns and its stand-alone operators:
//   upload / free                     ChunkIndexSource            crates/jolt-kernels/src/optimized/lazy_ra.rs:39-51
//   materialize (address-folded view) eq(r_chunk)[hot_i(j)]       lazy_ra.rs:9-11, optimized/booleanity.rs:32-38
//   pushforward G tables              G_i[k] = sum_j w_j [hot_i(j) = k]   optimized/booleanity.rs:24-31
// The lazily bound member that consumes the source lives with the other members in capi.hip.
#include <algorithm>

#include "onehot_kernels.cuh"

using namespace jolt;

int32_t jolt_internal_table_new(jolt_ctx* ctx, size_t len, jolt_table** out);
int32_t jolt_internal_ensure_scratch(jolt_ctx* ctx, size_t partials, size_t results);

extern "C" int32_t jolt_onehot_upload(jolt_ctx* ctx, const uint8_t* indices, size_t n_polys, size_t cycles, uint32_t k, jolt_onehot** out) {
    if (!ctx || !indices || !out || n_polys == 0 || cycles == 0) return JOLT_ERR_INVALID_ARG;
    if (k == 0 || k > 255) return JOLT_ERR_UNSUPPORTED;  // 0xFF marks a cold cycle
    for (size_t i = 0; i < n_polys * cycles; ++i)
        if (indices[i] != kOneHotCold && indices[i] >= k) { ctx->last_error = "hot index outside the scale table"; return JOLT_ERR_INVALID_ARG; }
    jolt_onehot* s = new (std::nothrow) jolt_onehot();
    if (!s) return JOLT_ERR_OOM;
    s->ctx = ctx;
    s->n_polys = n_polys;
    s->cycles = cycles;
    s->k = k;
    hipError_t e = hipMalloc((void**)&s->idx, n_polys * cycles);
    if (e == hipSuccess) e = hipMemcpyAsync(s->idx, indices, n_polys * cycles, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        ctx->last_error = std::string("onehot upload: ") + hipGetErrorString(e);
        if (s->idx) (void)hipFree(s->idx);
        delete s;
        return e == hipErrorOutOfMemory ? JOLT_ERR_OOM : JOLT_ERR_HIP;
    }
    *out = s;
    return JOLT_OK;
}

extern "C" int32_t jolt_onehot_free(jolt_ctx* ctx, jolt_onehot* s) {
    if (!s) return JOLT_OK;
    jolt_ctx* c = ctx ? ctx : s->ctx;
    if (c) { (void)jolt_internal_engine_quiesce(c); (void)hipStreamSynchronize(c->stream); }
    if (s->idx) (void)hipFree(s->idx);
    delete s;
    return JOLT_OK;
}

// out[j] = scale_table[index(poly, j)] (zero on cold cycles): the dense address-folded selector column
extern "C" int32_t jolt_onehot_materialize(jolt_ctx* ctx, const jolt_onehot* s, size_t poly, const jolt_table* scale_table, jolt_table** out) {
    if (!ctx || !s || !scale_table || !out) return JOLT_ERR_INVALID_ARG;
    if (poly >= s->n_polys) return JOLT_ERR_INVALID_ARG;
    if (scale_table->len != s->k) return JOLT_ERR_SIZE_MISMATCH;
    JOLT_TRY(jolt_internal_engine_quiesce(ctx));
    jolt_table* t = nullptr;
    JOLT_TRY(jolt_internal_table_new(ctx, s->cycles, &t));
    OneHotDense o;
    for (int i = 0; i < kMaxBatchTables; ++i) o.out[i] = i == 0 ? t->data() : nullptr;
    hipLaunchKernelGGL(k_onehot_materialize, dim3((unsigned)((s->cycles + kBlock - 1) / kBlock), 1), dim3(kBlock), 0, ctx->stream, (const Fr*)scale_table->data(),
                       (size_t)0, (const uint8_t*)s->idx, s->cycles, 1u, s->k, poly, o);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { jolt_table_free(ctx, t); ctx->last_error = hipGetErrorString(e); return JOLT_ERR_HIP; }
    *out = t;
    return JOLT_OK;
}

// out[p * K + k] = sum_j weights[j] * [index(p, j) == k]
extern "C" int32_t jolt_onehot_pushforward(jolt_ctx* ctx, const jolt_onehot* s, const jolt_table* weights, jolt_table** out) {
    if (!ctx || !s || !weights || !out) return JOLT_ERR_INVALID_ARG;
    if (weights->len != s->cycles) return JOLT_ERR_SIZE_MISMATCH;
    JOLT_TRY(jolt_internal_engine_quiesce(ctx));
    const int nblocks = (int)std::max<size_t>(1, std::min<size_t>((s->cycles + 4 * kBlock - 1) / (4 * kBlock), 256));
    const size_t part = s->n_polys * (size_t)nblocks * s->k;
    JOLT_TRY(jolt_internal_ensure_scratch(ctx, part + 8, 8));
    jolt_table* t = nullptr;
    JOLT_TRY(jolt_internal_table_new(ctx, s->n_polys * s->k, &t));
    hipLaunchKernelGGL(k_onehot_pushforward, dim3(nblocks, (unsigned)s->n_polys), dim3(kBlock), s->k * sizeof(Fr), ctx->stream, (const uint8_t*)s->idx,
                       (const Fr*)weights->data(), s->cycles, s->k, ctx->d_partials);
    hipLaunchKernelGGL(k_onehot_pushforward_reduce, dim3((s->k + kBlock - 1) / kBlock, (unsigned)s->n_polys), dim3(kBlock), 0, ctx->stream,
                       (const Fr*)ctx->d_partials, nblocks, s->k, t->data());
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { jolt_table_free(ctx, t); ctx->last_error = hipGetErrorString(e); return JOLT_ERR_HIP; }
    *out = t;
    return JOLT_OK;
}
