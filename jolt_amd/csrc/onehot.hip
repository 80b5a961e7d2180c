// jolt_amd/csrc/onehot.hip -- the hot-index source of one-hot selector columns and its stand-alone operators:
//   upload / free                     ChunkIndexSource            crates/jolt-kernels/src/optimized/lazy_ra.rs:39-51
//   materialize (address-folded view) eq(r_chunk)[hot_i(j)]       lazy_ra.rs:9-11, optimized/booleanity.rs:32-38
//   pushforward G tables              G_i[k] = sum_j w_j [hot_i(j) = k]   optimized/booleanity.rs:24-31
// The lazily bound member that consumes the source lives with the other members in capi.hip.
#include <algorithm>

#include "ints.hpp"
#include "onehot_kernels.hip.h"

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
        if (s->idx) jolt_internal_dev_free(ctx, s->idx);
        delete s;
        return e == hipErrorOutOfMemory ? JOLT_ERR_OOM : JOLT_ERR_HIP;
    }
    *out = s;
    return JOLT_OK;
}

// 16-bit hot indices: k up to 65535 with 0xFFFF = cold -- the K = 256 chunks of log T >= 25 (crates/jolt-prover/src/config.rs:175-186), where
// index 255 is a valid address and the one-byte cold marker is not available
extern "C" int32_t jolt_onehot_upload16(jolt_ctx* ctx, const uint16_t* indices, size_t n_polys, size_t cycles, uint32_t k, jolt_onehot** out) {
    if (!ctx || !indices || !out || n_polys == 0 || cycles == 0) return JOLT_ERR_INVALID_ARG;
    if (k == 0 || k > 1024) return JOLT_ERR_UNSUPPORTED;  // consumers keep K field elements in LDS; the reference never exceeds K = 256 chunks
    for (size_t i = 0; i < n_polys * cycles; ++i)
        if (indices[i] != kOneHotCold16 && indices[i] >= k) { ctx->last_error = "hot index outside the scale table"; return JOLT_ERR_INVALID_ARG; }
    jolt_onehot* s = new (std::nothrow) jolt_onehot();
    if (!s) return JOLT_ERR_OOM;
    s->ctx = ctx;
    s->n_polys = n_polys;
    s->cycles = cycles;
    s->k = k;
    s->wide = 1;
    hipError_t e = hipMalloc((void**)&s->idx, n_polys * cycles * 2);
    if (e == hipSuccess) e = hipMemcpyAsync(s->idx, indices, n_polys * cycles * 2, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        ctx->last_error = std::string("onehot upload: ") + hipGetErrorString(e);
        if (s->idx) jolt_internal_dev_free(ctx, s->idx);
        delete s;
        return e == hipErrorOutOfMemory ? JOLT_ERR_OOM : JOLT_ERR_HIP;
    }
    *out = s;
    return JOLT_OK;
}

extern "C" int32_t jolt_onehot_free(jolt_ctx* ctx, jolt_onehot* s) {
    if (!s) return JOLT_OK;
    jolt_ctx* c = s->ctx ? s->ctx : ctx;
    if (c) (void)jolt_internal_engine_quiesce(c);
    const bool pooled = c && s->idx && c->pool_live.count(s->idx);  // jolt_onehot_from_rows: back to the pool, reused in stream order; uploads: the runtime's block
    if (c && !pooled) (void)hipStreamSynchronize(c->stream);
    if (s->idx) { if (c) jolt_internal_dev_free(c, s->idx); else (void)hipFree(s->idx); }
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
                       (size_t)0, (const uint8_t*)s->idx, s->cycles, 1u, s->k, poly, o, s->wide);
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
    const bool lanes = s->k <= 32 && !s->wide;  // per-lane buckets in LDS (K * 2 KiB per wavefront): every column the reference folds this way has K = 16
    size_t per_block = 4096;        // cycles per wavefront: 64 per lane
    while ((s->cycles + per_block - 1) / per_block > 1024) per_block *= 2;
    const int nblocks = lanes ? (int)std::max<size_t>(1, (s->cycles + per_block - 1) / per_block)
                              : (int)std::max<size_t>(1, std::min<size_t>((s->cycles + 4 * kBlock - 1) / (4 * kBlock), 256));
    const size_t part = s->n_polys * (size_t)nblocks * s->k;
    JOLT_TRY(jolt_internal_ensure_scratch(ctx, part + 8, 8));
    jolt_table* t = nullptr;
    JOLT_TRY(jolt_internal_table_new(ctx, s->n_polys * s->k, &t));
    if (lanes)
        hipLaunchKernelGGL(k_onehot_pushforward_lanes, dim3((unsigned)s->n_polys, nblocks), dim3(64), (size_t)s->k * 2 * 64 * sizeof(uint4), ctx->stream,
                           (const uint8_t*)s->idx, (const Fr*)weights->data(), s->cycles, s->k, per_block, ctx->d_partials);
    else
        hipLaunchKernelGGL(k_onehot_pushforward, dim3(nblocks, (unsigned)s->n_polys), dim3(kBlock), s->k * sizeof(Fr), ctx->stream, (const uint8_t*)s->idx, s->wide,
                           (const Fr*)weights->data(), s->cycles, s->k, ctx->d_partials);
    if (s->k <= (uint32_t)kBlock && nblocks >= 8)
        hipLaunchKernelGGL(k_onehot_pushforward_reduce_split, dim3((unsigned)s->n_polys), dim3(kBlock), 0, ctx->stream, (const Fr*)ctx->d_partials, nblocks, s->k, t->data());
    else
        hipLaunchKernelGGL(k_onehot_pushforward_reduce, dim3((s->k + kBlock - 1) / kBlock, (unsigned)s->n_polys), dim3(kBlock), 0, ctx->stream,
                           (const Fr*)ctx->d_partials, nblocks, s->k, t->data());
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { jolt_table_free(ctx, t); ctx->last_error = hipGetErrorString(e); return JOLT_ERR_HIP; }
    *out = t;
    return JOLT_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Packed typed witness rows -> device tables: one upload of the trace-derived per-cycle records
// (e.g. CommittedColumnsWitness, crates/jolt-kernels/src/commitment.rs:25-32: rd_inc, ram_inc, lookup_index, bytecode_pc,
// ram_address) instead of one materialised field column per polynomial; the columns the members read are expanded on the
// device: integer fields promote to Fr (Polynomial::bind_to_field's compact scalars), address fields become hot-index columns
// through RaChunkSelector::chunk_u128 (crates/jolt-witness/src/witnesses/one_hot.rs:14-52).
// ------------------------------------------------------------------------------------------------------------------
struct jolt_rows {
    jolt_ctx* ctx = nullptr;
    uint8_t* data = nullptr;  // device, n_rows * row_bytes
    size_t n_rows = 0, row_bytes = 0;
    bool pending = false;  // jolt_rows_upload_begin's copy has not been waited for yet (jolt_rows_upload_wait synchronises the host with the copy stream)
};
// Every consumer of a row block: the extraction kernels run on the context's main stream, which has no dependency on the copy stream (by design: see
// jolt_rows_upload_begin), so a handle whose copy has not been waited for must be refused -- extracting from it would prove over a partly copied witness.
static int32_t rows_ready(jolt_ctx* ctx, const jolt_rows* rows) {
    if (rows->ctx != ctx) { ctx->last_error = "rows belong to another context"; return JOLT_ERR_INVALID_ARG; }
    if (rows->pending) { ctx->last_error = "rows used before jolt_rows_upload_wait: the copy begun by jolt_rows_upload_begin may not have landed"; return JOLT_ERR_INVALID_ARG; }
    return JOLT_OK;
}

namespace {
__device__ __forceinline__ uint64_t load_le(const uint8_t* p, uint32_t width) {
    uint64_t v = 0;
    for (uint32_t k = 0; k < width && k < 8; ++k) v |= (uint64_t)p[k] << (8 * k);
    return v;
}
static __global__ __launch_bounds__(kBlock) void k_rows_to_fr(const uint8_t* __restrict__ rows, size_t n_rows, size_t row_bytes, size_t offset, uint32_t width,
                                                              int is_signed, Fr* __restrict__ out) {
    size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n_rows) return;
    uint64_t v = load_le(rows + j * row_bytes + offset, width);
    bool negative = false;
    if (is_signed) {
        const int bits = (int)width * 8;
        if (bits < 64 && (v >> (bits - 1)) & 1) v |= ~0ull << bits;  // sign-extend
        if ((int64_t)v < 0) { negative = true; v = 0ull - v; }
    }
    Fr x = fr_from_u64(v);
    st_fr(out + j, negative ? neg(x) : x);
}
// The one-row lookahead window of the witness extractors (RandomAccessRows::window, crates/jolt-kernels/src/optimized/rows.rs:58-66):
// cycle j reads row j (lookahead = 0) or row j + 1 (lookahead = 1); rows at and beyond the physical trace are padding rows, and the
// last cycle of the domain has no next row at all (`None`).
static __global__ __launch_bounds__(kBlock) void k_rows_window_to_fr(const uint8_t* __restrict__ rows, size_t n_rows, size_t row_bytes, size_t offset, uint32_t width,
                                                                     int is_signed, int lookahead, size_t cycles, int64_t padding_value, int64_t none_value,
                                                                     Fr* __restrict__ out) {
    size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= cycles) return;
    const size_t src = j + (size_t)lookahead;
    int64_t sv;
    if (lookahead && src >= cycles) {
        sv = none_value;
    } else if (src >= n_rows) {
        sv = padding_value;
    } else {
        uint64_t v = load_le(rows + src * row_bytes + offset, width);
        if (is_signed) {
            const int bits = (int)width * 8;
            if (bits < 64 && (v >> (bits - 1)) & 1) v |= ~0ull << bits;
            sv = (int64_t)v;
        } else {
            if (width == 8 && (v >> 63)) {  // a full-width unsigned value does not fit the signed path
                st_fr(out + j, fr_from_u64(v));
                return;
            }
            sv = (int64_t)v;
        }
    }
    const bool negative = sv < 0;
    const Fr x = fr_from_u64(negative ? 0ull - (uint64_t)sv : (uint64_t)sv);
    st_fr(out + j, negative ? neg(x) : x);
}
struct ChunkShifts {
    uint32_t shift[kMaxBatchTables];
};
// Sentinel-packed address fields (InstructionCycleRow::{pc_plus_one, ram_address_plus_one}, optimized/instruction_read_raf.rs:82-123):
// 0 = no access (cold), v > 0 = address v - 1; chunk p = ((v - 1) >> shift[p]) & mask.  Fields of up to 16 bytes.
static __global__ __launch_bounds__(kBlock) void k_rows_sentinel_to_hot_indices(const uint8_t* __restrict__ rows, size_t n_rows, size_t row_bytes, size_t offset,
                                                                                uint32_t width, ChunkShifts sh, uint32_t log_k, size_t cycles, uint8_t* __restrict__ idx, uint32_t wide) {
    size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t p = blockIdx.y;
    if (j >= cycles) return;
    auto put = [&](uint32_t v, bool cold) {
        if (wide) reinterpret_cast<uint16_t*>(idx)[p * cycles + j] = cold ? kOneHotCold16 : (uint16_t)v;
        else idx[p * cycles + j] = cold ? kOneHotCold : (uint8_t)v;
    };
    uint64_t lo = 0, hi = 0;
    if (j < n_rows) {
        const uint8_t* f = rows + j * row_bytes + offset;
        lo = load_le(f, width < 8 ? width : 8);
        if (width > 8) hi = load_le(f + 8, width - 8);
    }
    if ((lo | hi) == 0) { put(0, true); return; }
    if (lo == 0) hi -= 1;  // borrow
    lo -= 1;
    const uint32_t s = sh.shift[p];
    uint64_t v = s < 64 ? (lo >> s) | (s ? hi << (64 - s) : 0ull) : hi >> (s - 64);
    put((uint32_t)(v & ((1u << log_k) - 1)), false);
}
// One thread per row: the address field (<= 16 bytes) and its validity byte are read ONCE and every chunk is cut from registers.  (Round 4 ran this with one grid row per
// chunk: 36 chunk columns re-read the 1 GB of rows 36 times, ~9 ms of the ~18 ms a step's extraction cost: profiles/r05_witness_upload_overlap.txt.)
static __global__ __launch_bounds__(kBlock) void k_rows_to_hot_indices(const uint8_t* __restrict__ rows, size_t n_rows, size_t row_bytes, size_t offset, uint32_t width,
                                                                       ChunkShifts sh, uint32_t n_polys, uint32_t log_k, size_t valid_offset, uint8_t* __restrict__ idx, uint32_t wide) {
    size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n_rows) return;
    const uint8_t* row = rows + j * row_bytes;
    const bool cold = valid_offset != ~(size_t)0 && row[valid_offset] == 0;
    uint64_t lo = 0, hi = 0;
    if (!cold) {
        lo = load_le(row + offset, width < 8 ? width : 8);
        if (width > 8) hi = load_le(row + offset + 8, width - 8);
    }
    const uint32_t mask = (1u << log_k) - 1;
    for (uint32_t p = 0; p < n_polys; ++p) {
        const uint32_t s = sh.shift[p];
        const uint64_t v = s == 0 ? lo : (s < 64 ? (lo >> s) | (hi << (64 - s)) : hi >> (s - 64));
        if (wide) reinterpret_cast<uint16_t*>(idx)[(size_t)p * n_rows + j] = cold ? kOneHotCold16 : (uint16_t)((uint32_t)v & mask);
        else idx[(size_t)p * n_rows + j] = cold ? kOneHotCold : (uint8_t)((uint32_t)v & mask);
    }
}
// Every integer field of the rows in ONE pass (jolt_ints_from_rows_many): a workgroup stages kRowsTile whole rows in LDS with coalesced 16-byte loads and every thread
// then writes its row's value of field after field -- the rows are read once (1 GB at T = 2^22) instead of once per field (27 GB for the catalogue's columns).
constexpr int kRowsTile = 256;
constexpr int kRowsMaxFields = 48;
struct RowFields {
    uint32_t offset[kRowsMaxFields];
    uint8_t width[kRowsMaxFields], is_signed[kRowsMaxFields];
    uint64_t* out[kRowsMaxFields];
    uint32_t n;
};
static __global__ __launch_bounds__(kRowsTile) void k_rows_to_ints_many(const uint8_t* __restrict__ rows, size_t n_rows, uint32_t row_bytes, RowFields f) {
    extern __shared__ __align__(16) uint8_t rows_tile[];
    const size_t row0 = (size_t)blockIdx.x * kRowsTile;
    const size_t n_here = n_rows - row0 < (size_t)kRowsTile ? n_rows - row0 : (size_t)kRowsTile;
    const size_t bytes = n_here * row_bytes;  // row_bytes is a multiple of 8 (checked by the caller), the tile starts 16-byte aligned for full tiles of 256 rows
    const uint8_t* src = rows + row0 * row_bytes;
    for (size_t b = (size_t)threadIdx.x * 8; b < bytes; b += (size_t)kRowsTile * 8) *reinterpret_cast<uint64_t*>(rows_tile + b) = *reinterpret_cast<const uint64_t*>(src + b);
    __syncthreads();
    if (threadIdx.x >= n_here) return;
    const uint8_t* row = rows_tile + (size_t)threadIdx.x * row_bytes;
    for (uint32_t k = 0; k < f.n; ++k) {
        const uint32_t width = f.width[k];
        uint64_t v = load_le(row + f.offset[k], width);
        if (f.is_signed[k]) {
            const int bits = (int)width * 8;
            if (bits < 64 && (v >> (bits - 1)) & 1) v |= ~0ull << bits;  // sign-extend
        }
        f.out[k][row0 + threadIdx.x] = v;
    }
}
}  // namespace

extern "C" int32_t jolt_rows_upload(jolt_ctx* ctx, const void* rows, size_t n_rows, size_t row_bytes, jolt_rows** out) {
    if (!ctx || !rows || !out || n_rows == 0 || row_bytes == 0) return JOLT_ERR_INVALID_ARG;
    jolt_rows* r = new (std::nothrow) jolt_rows();
    if (!r) return JOLT_ERR_OOM;
    r->ctx = ctx;
    r->n_rows = n_rows;
    r->row_bytes = row_bytes;
    // the device block comes from the context's pool (a proof per step re-uses last step's block instead of a 1 GB hipMalloc / hipFree pair)
    const int32_t as = jolt_internal_dev_alloc(ctx, n_rows * row_bytes, (void**)&r->data);
    if (as != JOLT_OK) { delete r; return as; }
    // pageable `rows`: the runtime stages the copy (~23 GB/s measured); memory from jolt_host_pinned_alloc goes over the link directly (~55 GB/s)
    hipError_t e = hipMemcpyAsync(r->data, rows, n_rows * row_bytes, hipMemcpyHostToDevice, ctx->stream);
    const hipError_t sync = hipStreamSynchronize(ctx->stream);  // the caller's buffer may be short-lived
    if (e == hipSuccess) e = sync;
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ctx->last_error = std::string("rows upload: ") + hipGetErrorString(e);
        jolt_internal_dev_free(ctx, r->data);
        delete r;
        return JOLT_ERR_HIP;
    }
    *out = r;
    return JOLT_OK;
}
// The same copy IN FLIGHT while the context works on something else -- the next proof's witness moving over the link under the current proof's kernels (round 5).
// `rows` must be page-locked (jolt_host_pinned_alloc) and stay untouched until jolt_rows_upload_wait returned.  begin: the device block comes from the pool, whose
// reuse is ordered on the MAIN stream, so the copy stream first waits for the main stream's position at this call, then copies.  wait: the host synchronises with the
// copy stream (the copy was begun a proof ago: nothing to wait for in practice); from then on the handle is what jolt_rows_upload returns.
extern "C" int32_t jolt_rows_upload_begin(jolt_ctx* ctx, const void* rows, size_t n_rows, size_t row_bytes, jolt_rows** out) {
    if (!ctx || !rows || !out || n_rows == 0 || row_bytes == 0) return JOLT_ERR_INVALID_ARG;
    if (!ctx->copy_stream) {
        JOLT_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        JOLT_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_copy_fork, hipEventDisableTiming));
    }
    jolt_rows* r = new (std::nothrow) jolt_rows();
    if (!r) return JOLT_ERR_OOM;
    r->ctx = ctx;
    r->n_rows = n_rows;
    r->row_bytes = row_bytes;
    int32_t st = jolt_internal_dev_alloc(ctx, n_rows * row_bytes, (void**)&r->data);
    if (st != JOLT_OK) { delete r; return st; }
    // NOTHING is enqueued behind the copy on the copy stream (no event record): the runtime multiplexes its streams onto four hardware queues, and a marker that waits
    // for the copy's completion blocks every later packet of whichever stream shares that queue -- with the main stream on it, the next proof's first kernels waited
    // out the whole copy (prepare 2.7 -> 20.3 ms, profiles/r05_witness_upload_overlap.txt).  _wait synchronises the HOST with the copy stream instead; by then the
    // copy has long landed.
    hipError_t e = hipEventRecord(ctx->ev_copy_fork, ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->copy_stream, ctx->ev_copy_fork, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(r->data, rows, n_rows * row_bytes, hipMemcpyHostToDevice, ctx->copy_stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(ctx->copy_stream);
        ctx->last_error = std::string("rows upload (begin): ") + hipGetErrorString(e);
        jolt_internal_dev_free(ctx, r->data);
        delete r;
        return JOLT_ERR_HIP;
    }
    r->pending = true;
    *out = r;
    return JOLT_OK;
}
extern "C" int32_t jolt_rows_upload_wait(jolt_ctx* ctx, jolt_rows* r) {
    if (!ctx || !r) return JOLT_ERR_INVALID_ARG;
    if (!r->pending) return JOLT_OK;
    JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));  // host-side: the copy was begun a proof ago
    r->pending = false;
    return JOLT_OK;
}
// Page-locked host memory for the buffers a caller fills once per proof and hands to jolt_rows_upload (the tracer's packed cycle rows): the H2D copy then runs
// at the link rate without the runtime's staging copy.  Plain memory otherwise: any thread may write it, jolt_host_pinned_free releases it.
extern "C" int32_t jolt_host_pinned_alloc(jolt_ctx* ctx, size_t bytes, void** out) {
    if (!ctx || !out || bytes == 0) return JOLT_ERR_INVALID_ARG;
    *out = nullptr;
    const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        ctx->last_error = std::string("pinned alloc: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? JOLT_ERR_OOM : JOLT_ERR_HIP;
    }
    return JOLT_OK;
}
extern "C" int32_t jolt_host_pinned_free(jolt_ctx* ctx, void* p) {
    if (!p) return JOLT_OK;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);  // a copy out of the block may still be queued
    const hipError_t e = hipHostFree(p);
    if (e != hipSuccess) { (void)hipGetLastError(); if (ctx) ctx->last_error = std::string("pinned free: ") + hipGetErrorString(e); return JOLT_ERR_HIP; }
    return JOLT_OK;
}
extern "C" int32_t jolt_rows_free(jolt_ctx* ctx, jolt_rows* r) {
    if (!r) return JOLT_OK;
    if (ctx && r->ctx && ctx != r->ctx) return JOLT_ERR_INVALID_ARG;  // the block belongs to the pool of the context that uploaded it
    jolt_ctx* c = r->ctx ? r->ctx : ctx;
    if (c) { (void)jolt_internal_engine_quiesce(c); (void)hipStreamSynchronize(c->stream); }
    if (r->pending && c && c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);  // a copy that was begun and never waited for still writes the block
    if (r->data) { if (c) jolt_internal_dev_free(c, r->data); else (void)hipFree(r->data); }
    delete r;
    return JOLT_OK;
}

extern "C" int32_t jolt_table_from_rows(jolt_ctx* ctx, const jolt_rows* rows, size_t offset, uint32_t width, int32_t is_signed, jolt_table** out) {
    if (!ctx || !rows || !out) return JOLT_ERR_INVALID_ARG;
    JOLT_TRY(rows_ready(ctx, rows));
    if (!(width == 1 || width == 2 || width == 4 || width == 8) || width > rows->row_bytes || offset > rows->row_bytes - width) return JOLT_ERR_INVALID_ARG;
    JOLT_TRY(jolt_internal_engine_quiesce(ctx));
    jolt_table* t = nullptr;
    JOLT_TRY(jolt_internal_table_new(ctx, rows->n_rows, &t));
    hipLaunchKernelGGL(k_rows_to_fr, dim3((unsigned)((rows->n_rows + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, (const uint8_t*)rows->data, rows->n_rows,
                       rows->row_bytes, offset, width, is_signed ? 1 : 0, t->data());
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { jolt_table_free(ctx, t); ctx->last_error = hipGetErrorString(e); return JOLT_ERR_HIP; }
    *out = t;
    return JOLT_OK;
}

// out[j] = the integer field of row j, widened to 64 bits (sign-extended when is_signed): a typed witness column as the COMPACT scalars the small-scalar members and
// operators read (jolt_member_create_lc_small, jolt_r1cs_*_small, ...), extracted on the device from the uploaded rows -- no promotion to field elements
static __global__ __launch_bounds__(kBlock) void k_rows_to_ints(const uint8_t* __restrict__ rows, size_t n_rows, size_t row_bytes, size_t offset, uint32_t width, int is_signed,
                                                                uint64_t* __restrict__ out) {
    size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n_rows) return;
    uint64_t v = load_le(rows + j * row_bytes + offset, width);
    const int bits = (int)width * 8;
    if (is_signed && bits < 64 && ((v >> (bits - 1)) & 1)) v |= ~0ull << bits;
    out[j] = v;
}
extern "C" int32_t jolt_ints_from_rows(jolt_ctx* ctx, const jolt_rows* rows, size_t offset, uint32_t width, int32_t is_signed, jolt_ints** out) {
    if (!ctx || !rows || !out) return JOLT_ERR_INVALID_ARG;
    JOLT_TRY(rows_ready(ctx, rows));
    if (!(width == 1 || width == 2 || width == 4 || width == 8) || width > rows->row_bytes || offset > rows->row_bytes - width) return JOLT_ERR_INVALID_ARG;
    JOLT_TRY(jolt_internal_engine_quiesce(ctx));
    jolt_ints* v = new (std::nothrow) jolt_ints();
    if (!v) return JOLT_ERR_OOM;
    v->ctx = ctx;
    v->count = rows->n_rows;
    v->kind = is_signed ? JOLT_INT_I64 : JOLT_INT_U64;
    // from the context's pool (round 5): a proof per step extracts ~30 columns, and hipMalloc / hipFree pairs cost ~0.4 ms each and synchronise the device --
    // 24 of the 26 ms the overlapped witness upload still paid per step (profiles/r05_witness_upload_overlap.txt); jolt_ints_free hands pool blocks back in stream order
    {
        const int32_t as = jolt_internal_dev_alloc(ctx, std::max<size_t>(v->count, 1) * 8, &v->data);
        if (as != JOLT_OK) { delete v; return as; }
    }
    hipError_t e = hipSuccess;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_rows_to_ints, dim3((unsigned)((rows->n_rows + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, (const uint8_t*)rows->data, rows->n_rows,
                           rows->row_bytes, offset, width, is_signed ? 1 : 0, (uint64_t*)v->data);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        ctx->last_error = std::string("ints from rows: ") + hipGetErrorString(e);
        if (v->data) jolt_internal_dev_free(ctx, v->data);
        delete v;
        return e == hipErrorOutOfMemory ? JOLT_ERR_OOM : JOLT_ERR_HIP;
    }
    *out = v;
    return JOLT_OK;
}

// jolt_ints_from_rows for n_fields fields at once: out[k] = field k as a resident integer column (the same values, one pass over the rows)
extern "C" int32_t jolt_ints_from_rows_many(jolt_ctx* ctx, const jolt_rows* rows, const size_t* offsets, const uint32_t* widths, const int32_t* is_signed, size_t n_fields,
                                            jolt_ints** out) {
    if (!ctx || !rows || !offsets || !widths || !is_signed || !out) return JOLT_ERR_INVALID_ARG;
    JOLT_TRY(rows_ready(ctx, rows));
    for (size_t k = 0; k < n_fields; ++k) {
        const uint32_t w = widths[k];
        if (!(w == 1 || w == 2 || w == 4 || w == 8) || w > rows->row_bytes || offsets[k] > rows->row_bytes - w) return JOLT_ERR_INVALID_ARG;
        out[k] = nullptr;
    }
    const size_t tile_bytes = (size_t)kRowsTile * rows->row_bytes;
    if (rows->row_bytes % 8 != 0 || tile_bytes > ctx->max_lds_per_block || rows->row_bytes > 0xFFFFFFFFull) {  // rows that do not stage: field by field
        for (size_t k = 0; k < n_fields; ++k) {
            const int32_t st = jolt_ints_from_rows(ctx, rows, offsets[k], widths[k], is_signed[k], &out[k]);
            if (st != JOLT_OK) { for (size_t q = 0; q < k; ++q) { jolt_ints_free(ctx, out[q]); out[q] = nullptr; } return st; }
        }
        return JOLT_OK;
    }
    JOLT_TRY(jolt_internal_engine_quiesce(ctx));
    auto release = [&](size_t upto) { for (size_t q = 0; q < upto; ++q) { jolt_ints_free(ctx, out[q]); out[q] = nullptr; } };
    for (size_t k = 0; k < n_fields; ++k) {
        jolt_ints* v = new (std::nothrow) jolt_ints();
        if (!v) { release(k); return JOLT_ERR_OOM; }
        v->ctx = ctx;
        v->count = rows->n_rows;
        v->kind = is_signed[k] ? JOLT_INT_I64 : JOLT_INT_U64;
        const int32_t as = jolt_internal_dev_alloc(ctx, std::max<size_t>(v->count, 1) * 8, &v->data);
        if (as != JOLT_OK) { delete v; release(k); return as; }
        out[k] = v;
    }
    if (!ctx->rows_many_attr_set) {  // once per context (= per device): the attribute is per device
        if (hipFuncSetAttribute((const void*)k_rows_to_ints_many, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->max_lds_per_block) != hipSuccess) {
            (void)hipGetLastError();  // refused: the tile does not stage on this device -> field by field
            release(n_fields);
            for (size_t k = 0; k < n_fields; ++k) {
                const int32_t st = jolt_ints_from_rows(ctx, rows, offsets[k], widths[k], is_signed[k], &out[k]);
                if (st != JOLT_OK) { release(k); return st; }
            }
            return JOLT_OK;
        }
        ctx->rows_many_attr_set = true;
    }
    for (size_t k0 = 0; k0 < n_fields; k0 += kRowsMaxFields) {
        RowFields f;
        f.n = (uint32_t)std::min<size_t>(kRowsMaxFields, n_fields - k0);
        for (uint32_t k = 0; k < (uint32_t)kRowsMaxFields; ++k) {
            const bool live = k < f.n;
            f.offset[k] = live ? (uint32_t)offsets[k0 + k] : 0u;
            f.width[k] = live ? (uint8_t)widths[k0 + k] : (uint8_t)1;
            f.is_signed[k] = live && is_signed[k0 + k] ? 1 : 0;
            f.out[k] = live ? (uint64_t*)out[k0 + k]->data : nullptr;
        }
        hipLaunchKernelGGL(k_rows_to_ints_many, dim3((unsigned)((rows->n_rows + kRowsTile - 1) / kRowsTile)), dim3(kRowsTile), tile_bytes, ctx->stream, (const uint8_t*)rows->data,
                           rows->n_rows, (uint32_t)rows->row_bytes, f);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ctx->last_error = std::string("ints from rows (many): ") + hipGetErrorString(e);
        release(n_fields);
        return JOLT_ERR_HIP;
    }
    return JOLT_OK;
}

extern "C" int32_t jolt_onehot_from_rows(jolt_ctx* ctx, const jolt_rows* rows, size_t offset, uint32_t width, const uint32_t* shifts, size_t n_polys,
                                         uint32_t log_k, size_t valid_offset, jolt_onehot** out) {
    if (!ctx || !rows || !shifts || !out || n_polys == 0 || n_polys > (size_t)kMaxBatchTables) return JOLT_ERR_INVALID_ARG;
    JOLT_TRY(rows_ready(ctx, rows));
    if (log_k == 0 || log_k > 8 || width == 0 || width > 16 || width > rows->row_bytes || offset > rows->row_bytes - width) return JOLT_ERR_INVALID_ARG;  // log_k = 8: 16-bit indices
    if (valid_offset != ~(size_t)0 && valid_offset >= rows->row_bytes) return JOLT_ERR_INVALID_ARG;
    ChunkShifts sh;
    for (size_t p = 0; p < (size_t)kMaxBatchTables; ++p) {
        sh.shift[p] = p < n_polys ? shifts[p] : 0;
        if (p < n_polys && shifts[p] + log_k > width * 8) return JOLT_ERR_INVALID_ARG;
    }
    JOLT_TRY(jolt_internal_engine_quiesce(ctx));
    jolt_onehot* s = new (std::nothrow) jolt_onehot();
    if (!s) return JOLT_ERR_OOM;
    s->ctx = ctx;
    s->n_polys = n_polys;
    s->cycles = rows->n_rows;
    s->k = 1u << log_k;
    s->wide = log_k > 7 ? 1u : 0u;
    hipError_t e = hipSuccess;
    {
        const int32_t as = jolt_internal_dev_alloc(ctx, std::max<size_t>((n_polys * rows->n_rows) << s->wide, 1), (void**)&s->idx);  // pooled, as jolt_ints_from_rows
        if (as != JOLT_OK) { delete s; return as; }
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_rows_to_hot_indices, dim3((unsigned)((rows->n_rows + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream,
                           (const uint8_t*)rows->data, rows->n_rows, rows->row_bytes, offset, width, sh, (uint32_t)n_polys, log_k, valid_offset, s->idx, s->wide);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        ctx->last_error = std::string("onehot from rows: ") + hipGetErrorString(e);
        if (s->idx) jolt_internal_dev_free(ctx, s->idx);
        delete s;
        return e == hipErrorOutOfMemory ? JOLT_ERR_OOM : JOLT_ERR_HIP;
    }
    *out = s;
    return JOLT_OK;
}

extern "C" int32_t jolt_table_from_rows_window(jolt_ctx* ctx, const jolt_rows* rows, size_t offset, uint32_t width, int32_t is_signed, int32_t lookahead, size_t cycles,
                                               int64_t padding_value, int64_t none_value, jolt_table** out) {
    if (!ctx || !rows || !out) return JOLT_ERR_INVALID_ARG;
    JOLT_TRY(rows_ready(ctx, rows));
    if (!(width == 1 || width == 2 || width == 4 || width == 8) || width > rows->row_bytes || offset > rows->row_bytes - width || (lookahead != 0 && lookahead != 1)) return JOLT_ERR_INVALID_ARG;
    if (rows->n_rows > cycles) return JOLT_ERR_SIZE_MISMATCH;  // rows.rs:44-53: the physical trace must fit the cycle domain
    JOLT_TRY(jolt_internal_engine_quiesce(ctx));
    jolt_table* t = nullptr;
    JOLT_TRY(jolt_internal_table_new(ctx, cycles, &t));
    if (cycles) {
        hipLaunchKernelGGL(k_rows_window_to_fr, dim3((unsigned)((cycles + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, (const uint8_t*)rows->data, rows->n_rows,
                           rows->row_bytes, offset, width, is_signed ? 1 : 0, lookahead, cycles, padding_value, none_value, t->data());
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { jolt_table_free(ctx, t); ctx->last_error = hipGetErrorString(e); return JOLT_ERR_HIP; }
    }
    *out = t;
    return JOLT_OK;
}

extern "C" int32_t jolt_onehot_from_rows_sentinel(jolt_ctx* ctx, const jolt_rows* rows, size_t offset, uint32_t width, const uint32_t* shifts, size_t n_polys,
                                                  uint32_t log_k, size_t cycles, jolt_onehot** out) {
    if (!ctx || !rows || !shifts || !out || n_polys == 0 || n_polys > (size_t)kMaxBatchTables) return JOLT_ERR_INVALID_ARG;
    JOLT_TRY(rows_ready(ctx, rows));
    if (log_k == 0 || log_k > 8 || width == 0 || width > 16 || width > rows->row_bytes || offset > rows->row_bytes - width) return JOLT_ERR_INVALID_ARG;
    if (rows->n_rows > cycles || cycles == 0) return JOLT_ERR_SIZE_MISMATCH;
    ChunkShifts sh;
    for (size_t p = 0; p < (size_t)kMaxBatchTables; ++p) {
        sh.shift[p] = p < n_polys ? shifts[p] : 0;
        if (p < n_polys && shifts[p] + log_k > width * 8) return JOLT_ERR_INVALID_ARG;
    }
    JOLT_TRY(jolt_internal_engine_quiesce(ctx));
    jolt_onehot* s = new (std::nothrow) jolt_onehot();
    if (!s) return JOLT_ERR_OOM;
    s->ctx = ctx;
    s->n_polys = n_polys;
    s->cycles = cycles;
    s->k = 1u << log_k;
    s->wide = log_k > 7 ? 1u : 0u;
    hipError_t e = hipSuccess;
    {
        const int32_t as = jolt_internal_dev_alloc(ctx, std::max<size_t>((n_polys * cycles) << s->wide, 1), (void**)&s->idx);  // pooled, as jolt_ints_from_rows
        if (as != JOLT_OK) { delete s; return as; }
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_rows_sentinel_to_hot_indices, dim3((unsigned)((cycles + kBlock - 1) / kBlock), (unsigned)n_polys), dim3(kBlock), 0, ctx->stream,
                           (const uint8_t*)rows->data, rows->n_rows, rows->row_bytes, offset, width, sh, log_k, cycles, s->idx, s->wide);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        ctx->last_error = std::string("onehot from sentinel rows: ") + hipGetErrorString(e);
        if (s->idx) jolt_internal_dev_free(ctx, s->idx);
        delete s;
        return e == hipErrorOutOfMemory ? JOLT_ERR_OOM : JOLT_ERR_HIP;
    }
    *out = s;
    return JOLT_OK;
}

// the hot indices back on the host (tests, debugging)
extern "C" int32_t jolt_onehot_download(jolt_ctx* ctx, const jolt_onehot* s, uint8_t* out) {
    if (!ctx || !s || !out) return JOLT_ERR_INVALID_ARG;
    JOLT_TRY(jolt_internal_engine_quiesce(ctx));
    if (s->wide) { ctx->last_error = "16-bit source: use jolt_onehot_download16"; return JOLT_ERR_INVALID_ARG; }
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(out, s->idx, s->n_polys * s->cycles, hipMemcpyDeviceToHost, ctx->stream));
    JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return JOLT_OK;
}

extern "C" int32_t jolt_onehot_download16(jolt_ctx* ctx, const jolt_onehot* s, uint16_t* out) {
    if (!ctx || !s || !out) return JOLT_ERR_INVALID_ARG;
    if (!s->wide) { ctx->last_error = "8-bit source: use jolt_onehot_download"; return JOLT_ERR_INVALID_ARG; }
    JOLT_TRY(jolt_internal_engine_quiesce(ctx));
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(out, s->idx, s->n_polys * s->cycles * 2, hipMemcpyDeviceToHost, ctx->stream));
    JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return JOLT_OK;
}
