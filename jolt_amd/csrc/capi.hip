// jolt_amd/csrc/capi.hip -- implementation of include/jolt_hip.h: context, tables, bind/eq kernels launches and the
// sumcheck members.  (MSM / HyperKZG device pieces live in msm.hip, the host-side mirror in host_mirror.hip.)
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <type_traits>

#include "host_mirror.hpp"
#include "ints.hpp"
#include "member.hpp"
#include "poly_kernels.hip.h"
#include "small_round.hip.h"
#include "engine_kernel.hip.h"
#include "onehot_kernels.hip.h"

using namespace jolt;

// ------------------------------------------------------------------------------------------------------------------
// status / context
// ------------------------------------------------------------------------------------------------------------------
extern "C" const char* jolt_status_string(int32_t s) {
    switch (s) {
        case JOLT_OK: return "ok";
        case JOLT_ERR_INVALID_ARG: return "invalid argument";
        case JOLT_ERR_NO_DEVICE: return "no usable gfx950 device";
        case JOLT_ERR_OOM: return "out of device memory";
        case JOLT_ERR_HIP: return "HIP runtime error";
        case JOLT_ERR_SIZE_MISMATCH: return "size mismatch";
        case JOLT_ERR_UNSUPPORTED: return "unsupported descriptor";
        case JOLT_ERR_NOT_FULLY_BOUND: return "member not fully bound";
        case JOLT_ERR_ROUND_CHECK: return "round check failed";
        case JOLT_ERR_SRS_TOO_SMALL: return "SRS too small";
        case JOLT_ERR_EMPTY_POINT: return "empty opening point";
        case JOLT_ERR_NOT_INVERTIBLE: return "value not invertible";
    }
    return "unknown status";
}
extern "C" int32_t jolt_abi_version(void) { return JOLT_HIP_ABI_VERSION; }

// The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and packets of streams that share a queue execute in order: with
// the main stream, three MSM lanes, the batch stream, the copy stream and the background hint stream, a long background kernel could sit in front of the main stream's
// latency-bound launches.  Eight queues measured -7 ms per proof at T = 2^22 (profiles/r06_hint_ab.txt).  The runtime reads the variable once, when it initialises --
// before any HIP call of a process that loads this library first; a caller's own setting wins.
namespace {
struct HwQueueDefault {
    HwQueueDefault() { (void)setenv("GPU_MAX_HW_QUEUES", "8", 0); }
} g_hw_queue_default;
}  // namespace

extern "C" int32_t jolt_ctx_create(int32_t device_id, void* stream, jolt_ctx** out) {
    if (!out) return JOLT_ERR_INVALID_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device_id < 0 || device_id >= count) return JOLT_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return JOLT_ERR_NO_DEVICE;
    // the code objects are gfx950-only: refuse anything else loudly instead of failing at the first launch
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return JOLT_ERR_NO_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return JOLT_ERR_NO_DEVICE;
    jolt_ctx* ctx = new (std::nothrow) jolt_ctx();
    if (!ctx) return JOLT_ERR_OOM;
    ctx->device = device_id;
    if (const char* fr_ = std::getenv("JOLT_FUSE_RATIO")) ctx->fuse_ratio = (size_t)std::max(0, std::atoi(fr_));
    if (const char* tp = std::getenv("JOLT_TAIL_PAIRS")) { if (std::atoll(tp) > 0) ctx->tail_pairs = (size_t)std::atoll(tp); }
    if (const char* ft = std::getenv("JOLT_FUSE_TAIL")) ctx->fuse_tail = std::atoi(ft) != 0;
    if (const char* rt = std::getenv("JOLT_ROUND_TRACE")) ctx->round_trace = std::atoi(rt) != 0;
    if (const char* gf = std::getenv("JOLT_GRID_FLOOR")) { if (std::atoi(gf) > 0) ctx->grid_floor = (size_t)std::atoi(gf); }
    if (const char* gm = std::getenv("JOLT_GRID_MULT")) { if (std::atoi(gm) > 0) ctx->grid_mult = (size_t)std::atoi(gm); }
    if (const char* ss = std::getenv("JOLT_SERIAL_STREAMS")) ctx->serial_streams = std::atoi(ss) != 0;
    if (const char* ll = std::getenv("JOLT_LAZY_LDS")) ctx->lazy_lds = std::atoi(ll) != 0;
    if (const char* rp = std::getenv("JOLT_UNIFORM_ROWS_PAIRS")) { if (std::atoll(rp) > 0) ctx->uniform_rows_pairs = (size_t)std::atoll(rp); }
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (prop.sharedMemPerBlock > 0) ctx->max_lds_per_block = prop.sharedMemPerBlock;
    if (const char* ml = std::getenv("JOLT_MSM_LDS_SORT")) ctx->msm_lds_sort = std::atoi(ml) != 0;
    if (const char* pe = std::getenv("JOLT_POOL")) ctx->pool_enabled = std::atoi(pe) != 0;
    if (const char* la = std::getenv("JOLT_MSM_LANES")) ctx->msm_lanes = std::max(1, std::min(4, std::atoi(la)));
    if (const char* mb = std::getenv("JOLT_MSM_BATCH")) ctx->msm_batch = std::atoi(mb) != 0;
    if (const char* po = std::getenv("JOLT_MSM_PAIR_OVERLAP")) ctx->msm_pair_overlap = std::atoi(po) != 0;
    if (const char* fx = std::getenv("JOLT_MSM_FIXED")) ctx->msm_fixed = std::atoi(fx) != 0;
    if (const char* sg = std::getenv("JOLT_MSM_STAGGER")) ctx->msm_stagger = std::atoi(sg) != 0;
    if (const char* gr = std::getenv("JOLT_FX_REDUCE")) ctx->msm_fx_grid_reduce = std::atoi(gr) != 0;
    if (const char* so = std::getenv("JOLT_FX_SOA")) ctx->msm_fx_soa = std::atoi(so) != 0;
    if (const char* cs = std::getenv("JOLT_MSM_CU_SPLIT")) ctx->msm_cu_split = std::max(0, std::min(7, std::atoi(cs)));
    if (const char* rd = std::getenv("JOLT_FX_REDUCE_DIV")) ctx->msm_fx_reduce_div = std::max(1, std::atoi(rd));
    if (const char* fl = std::getenv("JOLT_FX_LFORM")) ctx->msm_fx_lform = std::atoi(fl) != 0;
    if (const char* fs = std::getenv("JOLT_FX_STAGE")) ctx->msm_fx_stage = std::atoi(fs) != 0;
    if (const char* fp = std::getenv("JOLT_FX_PARTITION")) ctx->msm_fx_partition = std::atoi(fp) == 1 ? 1 : 2;
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return JOLT_ERR_HIP; }
        ctx->own_stream = true;
    }
    if (hipEventCreate(&ctx->ev_begin) != hipSuccess || hipEventCreate(&ctx->ev_end) != hipSuccess) { delete ctx; return JOLT_ERR_HIP; }
    for (hipEvent_t& e : ctx->ev_sort)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { delete ctx; return JOLT_ERR_HIP; }
    int32_t s = jolt_internal_ensure_scratch(ctx, 4096 * 8, 1024);
    if (s == JOLT_OK) {
        ctx->round_cap = 1024;
        if (hipHostMalloc((void**)&ctx->h_round, ctx->round_cap * sizeof(Fr), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostMalloc((void**)&ctx->h_flag, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipMalloc((void**)&ctx->d_counters, kTicketWords * sizeof(uint32_t)) != hipSuccess ||
            hipMalloc((void**)&ctx->d_round, ctx->round_cap * sizeof(Fr)) != hipSuccess ||
            hipMemset(ctx->d_counters, 0, kTicketWords * sizeof(uint32_t)) != hipSuccess)
            s = JOLT_ERR_HIP;
        else
            *ctx->h_flag = 0;
        // JOLT_SIDE_PRIORITY=1: the side streams above the main stream in priority -- short, latency-bound chains queued there (the dense commitments of
        // jolt_msm_g1_tables_begin) then get their workgroups in ahead of a long kernel on the main stream instead of behind it
        int prio_least = 0, prio_greatest = 0;
        const char* sp = std::getenv("JOLT_SIDE_PRIORITY");
        const bool side_high = sp && std::atoi(sp) != 0 && hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) == hipSuccess && prio_greatest != prio_least;
        (void)hipGetLastError();
        for (int k = 0; k < 3 && s == JOLT_OK; ++k) {
            const hipError_t ce = side_high ? hipStreamCreateWithPriority(&ctx->side[k], hipStreamNonBlocking, prio_greatest) : hipStreamCreateWithFlags(&ctx->side[k], hipStreamNonBlocking);
            if (ce != hipSuccess) s = JOLT_ERR_HIP;
        }
        if (s == JOLT_OK && ctx->msm_cu_split > 0) {
            // bit n of the mask <-> compute unit n; the split is taken inside every group of 8 consecutive bits (bits n with n mod 8 < k), which gives k of 8
            // CUs on every XCD whether the runtime numbers the CUs XCD-major or round-robin over the XCDs.  (Round 3 tested ((n / 8) mod 8) < k, which hands out WHOLE
            // groups of 8: under XCD-major numbering with k <= 4 the odd XCDs got no sort CUs -- profiles/r03_cu_split_ab.txt measured that partition.)
            const int words = (ctx->num_cus + 31) / 32;
            std::vector<uint32_t> mask_sort(words, 0u), mask_bucket(words, 0u);
            for (int n = 0; n < ctx->num_cus; ++n) {
                const bool sort_cu = (n % 8) < ctx->msm_cu_split;
                (sort_cu ? mask_sort : mask_bucket)[n / 32] |= 1u << (n % 32);
            }
            for (int k = 0; k < 4 && s == JOLT_OK; ++k) {
                if (hipExtStreamCreateWithCUMask(&ctx->sort_stream[k], (uint32_t)words, mask_sort.data()) != hipSuccess ||
                    hipExtStreamCreateWithCUMask(&ctx->bucket_stream[k], (uint32_t)words, mask_bucket.data()) != hipSuccess) {
                    (void)hipGetLastError();
                    ctx->msm_cu_split = 0;  // the runtime refuses CU masks: keep the single-stream lanes
                    break;
                }
                for (int j = 0; j < 4; ++j)
                    if (hipEventCreateWithFlags(&ctx->ev_phase[k][j], hipEventDisableTiming) != hipSuccess) s = JOLT_ERR_HIP;
            }
        }
        if (s == JOLT_OK && hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess) s = JOLT_ERR_HIP;
        for (int k = 0; k < 3 && s == JOLT_OK; ++k)
            if (hipEventCreateWithFlags(&ctx->ev_join[k], hipEventDisableTiming) != hipSuccess) s = JOLT_ERR_HIP;
    }
    if (s != JOLT_OK) { jolt_ctx_destroy(ctx); return s; }
    *out = ctx;
    return JOLT_OK;
}

void jolt_internal_engine_free(jolt_ctx* ctx);

extern "C" int32_t jolt_ctx_destroy(jolt_ctx* ctx) {
    if (!ctx) return JOLT_OK;
    (void)hipSetDevice(ctx->device);
    (void)jolt_internal_engine_quiesce(ctx);
    jolt_internal_engine_free(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    for (int k = 0; k < 3; ++k) if (ctx->side[k]) { (void)hipStreamSynchronize(ctx->side[k]); (void)hipStreamDestroy(ctx->side[k]); }
    for (int k = 0; k < 4; ++k) {
        if (ctx->sort_stream[k]) { (void)hipStreamSynchronize(ctx->sort_stream[k]); (void)hipStreamDestroy(ctx->sort_stream[k]); }
        if (ctx->bucket_stream[k]) { (void)hipStreamSynchronize(ctx->bucket_stream[k]); (void)hipStreamDestroy(ctx->bucket_stream[k]); }
        for (int j = 0; j < 4; ++j) if (ctx->ev_phase[k][j]) (void)hipEventDestroy(ctx->ev_phase[k][j]);
    }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    for (int k = 0; k < 3; ++k) if (ctx->ev_join[k]) (void)hipEventDestroy(ctx->ev_join[k]);
    (void)jolt_internal_pool_trim(ctx);
    for (auto& kv : ctx->pool_live) (void)hipFree(kv.first);  // blocks of handles the caller never freed
    ctx->pool_live.clear();
    if (ctx->d_partials) (void)hipFree(ctx->d_partials);
    if (ctx->d_results) (void)hipFree(ctx->d_results);
    if (ctx->h_results) (void)hipHostFree(ctx->h_results);
    if (ctx->h_round) (void)hipHostFree(ctx->h_round);
    if (ctx->h_flag) (void)hipHostFree(ctx->h_flag);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->d_round) (void)hipFree(ctx->d_round);
    for (int k = 0; k < 4; ++k) {
        if (ctx->msm_ws[k]) (void)hipFree(ctx->msm_ws[k]);
        if (ctx->msm_host[k]) (void)hipHostFree(ctx->msm_host[k]);
    }
    if (ctx->msm_batch_stream) { (void)hipStreamSynchronize(ctx->msm_batch_stream); (void)hipStreamDestroy(ctx->msm_batch_stream); }
    if (ctx->hint_stream) { (void)hipStreamSynchronize(ctx->hint_stream); (void)hipStreamDestroy(ctx->hint_stream); }
    if (ctx->msm_batch_ws) (void)hipFree(ctx->msm_batch_ws);
    if (ctx->msm_aux_stream) { (void)hipStreamSynchronize(ctx->msm_aux_stream); (void)hipStreamDestroy(ctx->msm_aux_stream); }
    for (auto& pair : ctx->ev_aux) for (hipEvent_t e : pair) if (e) (void)hipEventDestroy(e);
    if (ctx->msm_batch_host) (void)hipHostFree(ctx->msm_batch_host);
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    for (hipEvent_t e : ctx->ev_fx)
        if (e) (void)hipEventDestroy(e);
    if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
    if (ctx->ev_copy_fork) (void)hipEventDestroy(ctx->ev_copy_fork);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    for (hipEvent_t e : ctx->ev_sort) if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return JOLT_OK;
}

extern "C" int32_t jolt_ctx_synchronize(jolt_ctx* ctx) {
    if (!ctx) return JOLT_ERR_INVALID_ARG;
    JOLT_TRY(jolt_internal_engine_quiesce(ctx));
    for (int k = 0; k < 3; ++k) if (ctx->side[k]) JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->side[k]));
    if (ctx->hint_stream) JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->hint_stream));
    JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return JOLT_OK;
}
// The same without the background (hint) stream: what a caller that times a leg waits for -- the opening hint's class sums are meant to outlive the leg that began them.
// A host thread other than the one that created the context must select the context's device before it calls into the library (the HIP runtime's current device is
// per thread and starts at device 0): the stage operators of one protocol stage run on their own context and thread beside the stage's batched sumcheck (workload.py).
extern "C" int32_t jolt_ctx_bind_thread(jolt_ctx* ctx) {
    if (!ctx) return JOLT_ERR_INVALID_ARG;
    JOLT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return JOLT_OK;
}

extern "C" int32_t jolt_ctx_synchronize_foreground(jolt_ctx* ctx) {
    if (!ctx) return JOLT_ERR_INVALID_ARG;
    JOLT_TRY(jolt_internal_engine_quiesce(ctx));
    for (int k = 0; k < 3; ++k) if (ctx->side[k]) JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->side[k]));
    JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return JOLT_OK;
}
extern "C" const char* jolt_last_error(const jolt_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

extern "C" int32_t jolt_timer_begin(jolt_ctx* ctx) {
    if (!ctx) return JOLT_ERR_INVALID_ARG;
    JOLT_HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, ctx->stream));
    return JOLT_OK;
}
extern "C" int32_t jolt_timer_end(jolt_ctx* ctx, float* ms) {
    if (!ctx || !ms) return JOLT_ERR_INVALID_ARG;
    JOLT_HIP_TRY(ctx, hipEventRecord(ctx->ev_end, ctx->stream));
    JOLT_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_end));
    JOLT_HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev_begin, ctx->ev_end));
    return JOLT_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// device-memory pool (see ctx.hpp)
// ------------------------------------------------------------------------------------------------------------------
static inline size_t pool_class(size_t bytes) {
    if (bytes <= 4096) return 4096;
    size_t top = (size_t)1 << (63 - __builtin_clzll((unsigned long long)bytes));  // largest power of two <= bytes
    size_t step = top >> 2;                                                         // classes at 1, 1.25, 1.5, 1.75 x 2^k: <= 25 % slack
    return (bytes + step - 1) / step * step;
}
int32_t jolt_internal_pool_trim(jolt_ctx* ctx) {
    if (ctx->pool_free.empty()) return JOLT_OK;
    (void)hipDeviceSynchronize();  // cached blocks may still be read by work in flight
    for (auto& kv : ctx->pool_free)
        for (void* p : kv.second) (void)hipFree(p);
    ctx->pool_free.clear();
    ctx->pool_cached_bytes = 0;
    return JOLT_OK;
}
int32_t jolt_internal_dev_alloc(jolt_ctx* ctx, size_t bytes, void** out) {
    const size_t cls = pool_class(std::max<size_t>(bytes, 1));
    auto it = ctx->pool_free.find(cls);
    if (it != ctx->pool_free.end() && !it->second.empty()) {
        // a recycled block may have been written by a side stream in the previous batch round: the main stream joins those first
        JOLT_TRY(jolt_internal_join_side_writers(ctx));
        *out = it->second.back();
        it->second.pop_back();
        ctx->pool_cached_bytes -= cls;
    } else {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, cls);
        if (e == hipErrorOutOfMemory && ctx->pool_cached_bytes) {  // give the cached blocks back and retry once
            (void)hipGetLastError();
            JOLT_TRY(jolt_internal_pool_trim(ctx));
            e = hipMalloc(&p, cls);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            ctx->last_error = std::string("hipMalloc(") + std::to_string(cls) + "): " + hipGetErrorString(e);
            return e == hipErrorOutOfMemory ? JOLT_ERR_OOM : JOLT_ERR_HIP;
        }
        *out = p;
    }
    ctx->pool_live[*out] = cls;
    ctx->pool_live_bytes += cls;
    ctx->pool_peak_bytes = std::max(ctx->pool_peak_bytes, ctx->pool_live_bytes + ctx->pool_cached_bytes);
    return JOLT_OK;
}
void jolt_internal_dev_free(jolt_ctx* ctx, void* p) {
    if (!p) return;
    auto it = ctx ? ctx->pool_live.find(p) : std::unordered_map<void*, size_t>::iterator();
    if (!ctx || it == ctx->pool_live.end()) { (void)hipFree(p); return; }  // not ours (should not happen): the runtime's free synchronises
    const size_t cls = it->second;
    ctx->pool_live.erase(it);
    ctx->pool_live_bytes -= cls;
    if (!ctx->pool_enabled) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(p); return; }
    ctx->pool_free[cls].push_back(p);
    ctx->pool_cached_bytes += cls;
}
/* Release the cached device blocks of the context's pool back to the runtime (synchronises the device). */
extern "C" int32_t jolt_ctx_trim(jolt_ctx* ctx) {
    if (!ctx) return JOLT_ERR_INVALID_ARG;
    return jolt_internal_pool_trim(ctx);
}
/* Bytes held by live pool blocks / cached by the pool / high-water mark of both (device memory accounting of a proof). */
extern "C" int32_t jolt_ctx_memory_stats(const jolt_ctx* ctx, size_t* live_bytes, size_t* cached_bytes, size_t* peak_bytes) {
    if (!ctx) return JOLT_ERR_INVALID_ARG;
    if (live_bytes) *live_bytes = ctx->pool_live_bytes;
    if (cached_bytes) *cached_bytes = ctx->pool_cached_bytes;
    if (peak_bytes) *peak_bytes = ctx->pool_peak_bytes;
    return JOLT_OK;
}

/* Device memory the context holds OUTSIDE the pool: the grow-only workspaces of the four MSM lanes and of the batch of short MSMs (what the pool's statistics do not see). */
extern "C" int32_t jolt_ctx_workspace_stats(const jolt_ctx* ctx, size_t* msm_lane_bytes, size_t* msm_batch_bytes) {
    if (!ctx) return JOLT_ERR_INVALID_ARG;
    size_t lanes = 0;
    for (size_t cap : ctx->msm_ws_cap) lanes += cap;
    if (msm_lane_bytes) *msm_lane_bytes = lanes;
    if (msm_batch_bytes) *msm_batch_bytes = ctx->msm_batch_ws_cap;
    return JOLT_OK;
}

int32_t jolt_internal_ensure_scratch(jolt_ctx* ctx, size_t partials, size_t results) {
    if (partials > ctx->partials_cap || results > ctx->results_cap)
        for (int k = 0; k < 3; ++k) if (ctx->side[k]) JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->side[k]));
    if (partials > ctx->partials_cap) {
        partials = partials + partials / 2;  // grow geometrically: a reallocation stalls every stream
        if (ctx->d_partials) { JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); JOLT_HIP_TRY(ctx, hipFree(ctx->d_partials)); }
        ctx->d_partials = nullptr;
        JOLT_HIP_TRY(ctx, hipMalloc((void**)&ctx->d_partials, partials * sizeof(Fr)));
        ctx->partials_cap = partials;
    }
    if (results > ctx->results_cap) {
        if (ctx->d_results) { JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); JOLT_HIP_TRY(ctx, hipFree(ctx->d_results)); }
        if (ctx->h_results) JOLT_HIP_TRY(ctx, hipHostFree(ctx->h_results));
        ctx->d_results = nullptr;
        ctx->h_results = nullptr;
        JOLT_HIP_TRY(ctx, hipMalloc((void**)&ctx->d_results, results * sizeof(Fr)));
        JOLT_HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_results, results * sizeof(Fr), hipHostMallocDefault));
        ctx->results_cap = results;
    }
    return JOLT_OK;
}

static bool bool_lds_on() {
    static const bool on = !(std::getenv("JOLT_BOOL_LDS") && std::atoi(std::getenv("JOLT_BOOL_LDS")) == 0);
    return on;
}
// LDS bytes of one product group's branch tables (k_split_eq_uniform_lazy_lds): F polynomials x width x (K + 1) entries
constexpr size_t kLazyLdsMax = 48 * 1024;
static inline size_t lazy_lds_bytes(const jolt_member* m) {
    return (size_t)m->uni_F * m->lazy_width * ((size_t)m->onehot->k + 1) * sizeof(Fr);
}

// grid for a grid-stride sweep over n work items: enough blocks to fill 256 CUs x 8, never more than needed
static inline int sweep_grid(const jolt_ctx* ctx, size_t n) {
    size_t need = (n + kBlock - 1) / kBlock;
    size_t cap = (size_t)ctx->num_cus * 8;
    return (int)std::max<size_t>(1, std::min(need, cap));
}
// grid of a round-sum kernel: three workgroups per CU (fewer when the round is smaller), up to eight once every thread would
// still have ~16 items.  Every wavefront of these kernels ends with a shuffle reduction of its NE 256-bit accumulators and a
// ticket: with the fenced completion path of the first version small rounds wanted ONE workgroup per CU (T = 2^20, per pass: 7.6 /
// 8.1 / 7.9 / 8.1 ms for 1 / 2 / 4 / 8); without the fences 1..4 per CU measure the same at 2^20 and the big rounds of long traces
// want the memory-level parallelism of more (T = 2^22: 16.0 / 15.5 / 15.0 ms for 1 / 4 / 8, T = 2^24: 52.6 / 49.9 / 47.8 for 1 / 4 /
// this rule).  JOLT_GRID_MULT fixes the count per CU, JOLT_GRID_FLOOR the lower bound.
static inline int round_grid(const jolt_ctx* ctx, size_t n) {
    size_t need = (n + kBlock - 1) / kBlock;
    size_t cus = (size_t)ctx->num_cus;
    size_t cap = ctx->grid_mult ? cus * ctx->grid_mult : std::min(cus * 8, std::max(cus * ctx->grid_floor, need / 16));
    return (int)std::max<size_t>(1, std::min(need, cap));
}

// results[slot .. slot+ne) = sum over blocks of the partials just written
static int32_t reduce_into_results(jolt_ctx* ctx, int nblocks, int ne, size_t slot) {
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->d_partials, nblocks, ne, ctx->d_results + slot);
    JOLT_HIP_TRY(ctx, hipGetLastError());
    return JOLT_OK;
}
// copy results[0..count) to the host and wait (the protocol's per-round sync point)
static int32_t fetch_results(jolt_ctx* ctx, size_t count, jolt_fr_t* out) {
    (void)jolt_internal_engine_quiesce(ctx);
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(ctx->h_results, ctx->d_results, count * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(out, ctx->h_results, count * sizeof(Fr));
    return JOLT_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// tables
// ------------------------------------------------------------------------------------------------------------------
int32_t jolt_internal_table_new(jolt_ctx* ctx, size_t len, jolt_table** out) {
    jolt_table* t = new (std::nothrow) jolt_table();
    if (!t) return JOLT_ERR_OOM;
    t->ctx = ctx;
    t->len = len;
    size_t bytes = std::max<size_t>(len, 1) * sizeof(Fr);
    int32_t st = jolt_internal_dev_alloc(ctx, bytes, (void**)&t->buf[0]);
    if (st != JOLT_OK) {
        delete t;
        return st;
    }
    t->cap[0] = std::max<size_t>(len, 1);
    *out = t;
    return JOLT_OK;
}
int32_t jolt_internal_table_ensure_alt(jolt_table* t, size_t need) {
    int alt = t->cur < 0 ? 0 : 1 - t->cur;
    if (t->cap[alt] >= need) return JOLT_OK;
    jolt_ctx* ctx = t->ctx;
    if (t->buf[alt]) { jolt_internal_dev_free(ctx, t->buf[alt]); t->buf[alt] = nullptr; t->cap[alt] = 0; }
    JOLT_TRY(jolt_internal_dev_alloc(ctx, std::max<size_t>(need, 1) * sizeof(Fr), (void**)&t->buf[alt]));
    t->cap[alt] = std::max<size_t>(need, 1);
    return JOLT_OK;
}

extern "C" int32_t jolt_table_upload(jolt_ctx* ctx, const jolt_fr_t* host, size_t len, jolt_table** out) {
    if (!ctx || !out || (!host && len)) return JOLT_ERR_INVALID_ARG;
    JOLT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    jolt_table* t = nullptr;
    JOLT_TRY(jolt_internal_table_new(ctx, len, &t));
    if (len) {
        hipError_t e = hipMemcpyAsync(t->buf[0], host, len * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // host buffer may be pageable and short-lived
        if (e != hipSuccess) { jolt_table_free(ctx, t); ctx->last_error = hipGetErrorString(e); return JOLT_ERR_HIP; }
    }
    *out = t;
    return JOLT_OK;
}
extern "C" int32_t jolt_table_from_device(jolt_ctx* ctx, const void* dptr, size_t len, jolt_table** out) {
    if (!ctx || !out || (!dptr && len)) return JOLT_ERR_INVALID_ARG;
    jolt_table* t = nullptr;
    JOLT_TRY(jolt_internal_table_new(ctx, len, &t));
    if (len) JOLT_HIP_TRY(ctx, hipMemcpyAsync(t->buf[0], dptr, len * sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream));
    *out = t;
    return JOLT_OK;
}
extern "C" int32_t jolt_table_alloc(jolt_ctx* ctx, size_t len, jolt_table** out) {
    if (!ctx || !out) return JOLT_ERR_INVALID_ARG;
    jolt_table* t = nullptr;
    JOLT_TRY(jolt_internal_table_new(ctx, len, &t));
    if (len) JOLT_HIP_TRY(ctx, hipMemsetAsync(t->buf[0], 0, len * sizeof(Fr), ctx->stream));
    *out = t;
    return JOLT_OK;
}
extern "C" int32_t jolt_table_clone(jolt_ctx* ctx, const jolt_table* src, jolt_table** out) {
    if (!ctx || !src || !out) return JOLT_ERR_INVALID_ARG;
    return jolt_table_from_device(ctx, src->data(), src->len, out);
}
template <typename T, typename K>
static int32_t table_from_small(jolt_ctx* ctx, const T* host, size_t len, jolt_table** out, K kernel) {
    if (!ctx || !out || (!host && len)) return JOLT_ERR_INVALID_ARG;
    jolt_table* t = nullptr;
    JOLT_TRY(jolt_internal_table_new(ctx, len, &t));
    if (len) {
        T* staging = nullptr;
        hipError_t e = jolt_internal_dev_alloc(ctx, len * sizeof(T), (void**)&staging) == JOLT_OK ? hipSuccess : hipErrorOutOfMemory;
        if (e == hipSuccess) e = hipMemcpyAsync(staging, host, len * sizeof(T), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(kernel, dim3(sweep_grid(ctx, len)), dim3(kBlock), 0, ctx->stream, (const T*)staging, t->buf[0], len);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (staging) jolt_internal_dev_free(ctx, staging);
        if (e != hipSuccess) { jolt_table_free(ctx, t); ctx->last_error = hipGetErrorString(e); return JOLT_ERR_HIP; }
    }
    *out = t;
    return JOLT_OK;
}
extern "C" int32_t jolt_table_from_u64(jolt_ctx* ctx, const uint64_t* host, size_t len, jolt_table** out) {
    return table_from_small(ctx, host, len, out, k_from_u64);
}
extern "C" int32_t jolt_table_from_i64(jolt_ctx* ctx, const int64_t* host, size_t len, jolt_table** out) {
    return table_from_small(ctx, host, len, out, k_from_i64);
}
extern "C" int32_t jolt_table_download(jolt_ctx* ctx, const jolt_table* t, size_t offset, size_t len, jolt_fr_t* host) {
    (void)jolt_internal_engine_quiesce(ctx);
    if (!ctx || !t || (!host && len)) return JOLT_ERR_INVALID_ARG;
    if (len > t->len || offset > t->len - len) return JOLT_ERR_SIZE_MISMATCH;
    if (len) {
        JOLT_HIP_TRY(ctx, hipMemcpyAsync(host, t->data() + offset, len * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
        JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return JOLT_OK;
}
extern "C" int32_t jolt_table_len(const jolt_table* t, size_t* len) {
    if (!t || !len) return JOLT_ERR_INVALID_ARG;
    *len = t->len;
    return JOLT_OK;
}
extern "C" int32_t jolt_table_device_ptr(const jolt_table* t, void** p) {
    if (!t || !p) return JOLT_ERR_INVALID_ARG;
    *p = t->data();
    return JOLT_OK;
}
extern "C" int32_t jolt_table_free(jolt_ctx* ctx, jolt_table* t) {
    (void)jolt_internal_engine_quiesce(ctx ? ctx : (t ? t->ctx : nullptr));
    if (!t) return JOLT_OK;
    jolt_ctx* c = t->ctx ? t->ctx : ctx;
    // no synchronisation: the blocks go back to the context's pool and are reused in stream order (ctx.hpp)
    if (t->buf[0]) jolt_internal_dev_free(c, t->buf[0]);
    if (t->buf[1]) jolt_internal_dev_free(c, t->buf[1]);
    delete t;
    return JOLT_OK;
}

extern "C" int32_t jolt_table_slice(jolt_ctx* ctx, const jolt_table* parent, size_t offset, size_t len, jolt_table** out) {
    if (!ctx || !parent || !out) return JOLT_ERR_INVALID_ARG;
    if (len > parent->len || offset > parent->len - len) return JOLT_ERR_SIZE_MISMATCH;
    jolt_table* v = new (std::nothrow) jolt_table();
    if (!v) return JOLT_ERR_OOM;
    v->ctx = ctx;
    v->cur = -1;
    v->view = parent->data() + offset;
    v->view_len = len;
    v->len = len;
    *out = v;
    return JOLT_OK;
}
extern "C" int32_t jolt_table_write(jolt_ctx* ctx, jolt_table* t, size_t offset, const jolt_fr_t* host, size_t len) {
    (void)jolt_internal_engine_quiesce(ctx);
    if (!ctx || !t || (!host && len)) return JOLT_ERR_INVALID_ARG;
    if (len > t->len || offset > t->len - len) return JOLT_ERR_SIZE_MISMATCH;
    if (len) {
        JOLT_HIP_TRY(ctx, hipMemcpyAsync(t->data() + offset, host, len * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return JOLT_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// bind
// ------------------------------------------------------------------------------------------------------------------
// Bind k tables (possibly of different lengths: members of different round counts share a batch round) with one
// challenge: ceil(k/40) launches, blockIdx.y = table.
// Polynomial::bind_to_field (crates/jolt-poly/src/dense.rs:129-142): the first bind of tables that are still u64 witness columns (jolt_member_create_lc_small)
// writes field elements; from then on they are ordinary tables.
static int32_t bind_ints_to_field(jolt_ctx* ctx, jolt_table* const* tables, size_t k, const Fr& r) {
    const Fr a_rr = mul(sub(Fr::one(), r), Fr::r2()), b_rr = mul(r, Fr::r2());
    for (size_t base = 0; base < k; base += kMaxBatchTables) {
        const size_t cnt = std::min<size_t>(kMaxBatchTables, k - base);
        BindIntsBatch b;
        size_t max_half = 0;
        for (size_t i = 0; i < cnt; ++i) {
            jolt_table* t = tables[base + i];
            const size_t half = t->len / 2;
            JOLT_TRY(jolt_internal_table_ensure_alt(t, half));  // a view (cur < 0): buffer 0
            b.in[i] = reinterpret_cast<const uint64_t*>(t->ints);
            b.out[i] = t->buf[0];
            b.half[i] = half;
            max_half = std::max(max_half, half);
        }
        dim3 grid((unsigned)std::max<size_t>(1, std::min<size_t>((max_half + kBlock - 1) / kBlock, 1u << 20)), (unsigned)cnt);
        hipLaunchKernelGGL(k_bind_ints_to_field, grid, dim3(kBlock), 0, ctx->stream, b, a_rr, b_rr);
        JOLT_HIP_TRY(ctx, hipGetLastError());
        for (size_t i = 0; i < cnt; ++i) {
            jolt_table* t = tables[base + i];
            t->ints = nullptr;
            t->cur = 0;
            t->len /= 2;
        }
    }
    return JOLT_OK;
}

int32_t jolt_internal_bind(jolt_ctx* ctx, jolt_table* const* tables, size_t k, const Fr& r, int32_t order) {
    if (k == 0) return JOLT_OK;
    JOLT_TRY(jolt_internal_engine_quiesce(ctx));
    bool any_ints = false;
    for (size_t i = 0; i < k; ++i) {
        if (!tables[i]) return JOLT_ERR_INVALID_ARG;
        if (tables[i]->len < 2) { ctx->last_error = "cannot bind a zero-variable polynomial"; return JOLT_ERR_INVALID_ARG; }  // dense.rs:190,225 assert
        any_ints = any_ints || tables[i]->ints != nullptr;
    }
    if (any_ints) {  // integer-backed tables take the bind_to_field kernel, the others the ordinary one
        if (order != JOLT_ORDER_LOW_TO_HIGH) { ctx->last_error = "integer-backed tables bind LowToHigh only"; return JOLT_ERR_UNSUPPORTED; }
        std::vector<jolt_table*> small, dense;
        for (size_t i = 0; i < k; ++i) (tables[i]->ints ? small : dense).push_back(tables[i]);
        JOLT_TRY(bind_ints_to_field(ctx, small.data(), small.size(), r));
        return jolt_internal_bind(ctx, dense.data(), dense.size(), r, order);
    }
    const bool shifted = fr_low_limbs_zero(r);
    for (size_t base = 0; base < k; base += kMaxBatchTables) {
        size_t cnt = std::min<size_t>(kMaxBatchTables, k - base);
        BindBatch b;
        size_t max_half = 0;
        for (size_t i = 0; i < cnt; ++i) {
            jolt_table* t = tables[base + i];
            size_t half = t->len / 2;
            b.in[i] = t->data();
            b.half[i] = half;
            max_half = std::max(max_half, half);
            if (order == JOLT_ORDER_LOW_TO_HIGH || t->cur < 0) {  // out of place (a borrowed view is never written)
                JOLT_TRY(jolt_internal_table_ensure_alt(t, half));
                b.out[i] = t->buf[t->cur < 0 ? 0 : 1 - t->cur];
            } else {
                b.out[i] = t->data();
            }
        }
        // one output per thread: measured fastest (5.8 TB/s at 2^24, microbench) -- no grid-stride cap for bind
        dim3 grid((unsigned)std::max<size_t>(1, std::min<size_t>((max_half + kBlock - 1) / kBlock, 1u << 20)), (unsigned)cnt);
        if (order == JOLT_ORDER_LOW_TO_HIGH) {
            if (shifted) hipLaunchKernelGGL(k_bind_low_to_high<true>, grid, dim3(kBlock), 0, ctx->stream, b, r);
            else hipLaunchKernelGGL(k_bind_low_to_high<false>, grid, dim3(kBlock), 0, ctx->stream, b, r);
        } else {
            if (shifted) hipLaunchKernelGGL(k_bind_high_to_low<true>, grid, dim3(kBlock), 0, ctx->stream, b, r);
            else hipLaunchKernelGGL(k_bind_high_to_low<false>, grid, dim3(kBlock), 0, ctx->stream, b, r);
        }
        JOLT_HIP_TRY(ctx, hipGetLastError());
        for (size_t i = 0; i < cnt; ++i) {
            jolt_table* t = tables[base + i];
            if (t->cur < 0) t->cur = 0;
            else if (order == JOLT_ORDER_LOW_TO_HIGH) t->cur = 1 - t->cur;
            t->len = t->len / 2;
        }
    }
    return JOLT_OK;
}

extern "C" int32_t jolt_bind(jolt_ctx* ctx, jolt_table* const* tables, size_t k, const jolt_fr_t* r, int32_t order) {
    if (!ctx || (!tables && k) || !r) return JOLT_ERR_INVALID_ARG;
    if (order != JOLT_ORDER_LOW_TO_HIGH && order != JOLT_ORDER_HIGH_TO_LOW) return JOLT_ERR_INVALID_ARG;
    Fr rr = fr_from_abi(r);
    JOLT_REQUIRE(ctx, fr_is_canonical(rr), "bind challenge is not a canonical Fr");
    for (size_t i = 0; i < k; ++i) {
        if (!tables[i]) return JOLT_ERR_INVALID_ARG;
        if (tables[i]->len != tables[0]->len) return JOLT_ERR_SIZE_MISMATCH;
    }
    return jolt_internal_bind(ctx, tables, k, rr, order);
}

// ------------------------------------------------------------------------------------------------------------------
// eq / LT / eq+1 tables
// ------------------------------------------------------------------------------------------------------------------
// Build eq(r[..n], .) * scale into a fresh table by tensor steps of <= 8 variables; optionally keep every
// prefix level (used by eq+1 and by the split-eq member's cached tables when step = 1).
static int32_t eq_build(jolt_ctx* ctx, const Fr* r, size_t n, const Fr& scale, size_t step_vars, std::vector<jolt_table*>* levels,
                        jolt_table** out) {
    if (levels && step_vars == 1 && n >= 1 && n <= (size_t)kEqLevelsMax) {  // every level of a short point: one launch (k_eq_levels)
        EqLevels a;
        a.n = (int)n;
        const size_t first = levels->size();
        for (size_t j = 0; j <= n; ++j) {
            jolt_table* t = nullptr;
            const int32_t s = jolt_internal_table_new(ctx, (size_t)1 << j, &t);
            if (s != JOLT_OK) {
                for (size_t k = first; k < levels->size(); ++k) jolt_table_free(ctx, (*levels)[k]);
                levels->resize(first);
                return s;
            }
            levels->push_back(t);
            a.level[j] = t->data();
            if (j < n) a.r[j] = r[j];
        }
        hipLaunchKernelGGL(k_eq_levels, dim3(1), dim3(kBlock), 0, ctx->stream, a, scale);
        JOLT_HIP_TRY(ctx, hipGetLastError());
        *out = levels->back();
        return JOLT_OK;
    }
    jolt_table* cur = nullptr;
    JOLT_TRY(jolt_internal_table_new(ctx, 1, &cur));
    hipLaunchKernelGGL(k_fill_fr, dim3(1), dim3(64), 0, ctx->stream, cur->buf[0], scale, (size_t)1);  // by value: no host source to wait for
    JOLT_HIP_TRY(ctx, hipGetLastError());
    if (levels) levels->push_back(cur);
    size_t done = 0;
    while (done < n) {
        size_t c = std::min(step_vars, n - done);
        EqChunk ch;
        ch.c = (int)c;
        for (size_t k = 0; k < 8; ++k) ch.r[k] = k < c ? r[done + k] : Fr::zero();
        size_t out_len = (size_t)1 << (done + c);
        jolt_table* nxt = nullptr;
        int32_t s = jolt_internal_table_new(ctx, out_len, &nxt);
        if (s != JOLT_OK) { if (!levels) jolt_table_free(ctx, cur); return s; }
        hipLaunchKernelGGL(k_eq_expand, dim3(sweep_grid(ctx, out_len)), dim3(kBlock), 0, ctx->stream, (const Fr*)cur->data(), nxt->data(), out_len, ch);
        JOLT_HIP_TRY(ctx, hipGetLastError());
        if (levels) levels->push_back(nxt);
        else jolt_table_free(ctx, cur);
        cur = nxt;
        done += c;
    }
    *out = cur;
    return JOLT_OK;
}

// evals_cached (crates/jolt-poly/src/eq.rs:317-340) on the device: levels[j] = eq over the first j coordinates (used by rw_matrix.hip)
int32_t jolt_internal_eq_levels(jolt_ctx* ctx, const Fr* r, size_t n, const Fr& scale, std::vector<jolt_table*>* levels) {
    jolt_table* last = nullptr;
    return eq_build(ctx, r, n, scale, 1, levels, &last);
}

static int32_t read_point(jolt_ctx* ctx, const jolt_fr_t* r, size_t n, std::vector<Fr>& out) {
    out.resize(n);
    for (size_t i = 0; i < n; ++i) {
        out[i] = fr_from_abi(&r[i]);
        JOLT_REQUIRE(ctx, fr_is_canonical(out[i]), "point coordinate is not a canonical Fr");
    }
    return JOLT_OK;
}

extern "C" int32_t jolt_eq_evals(jolt_ctx* ctx, const jolt_fr_t* r, size_t n, const jolt_fr_t* scale, jolt_table** out) {
    if (!ctx || !out || (!r && n) || n > 40) return JOLT_ERR_INVALID_ARG;
    std::vector<Fr> pt;
    JOLT_TRY(read_point(ctx, r, n, pt));
    Fr s = scale ? fr_from_abi(scale) : Fr::one();
    return eq_build(ctx, pt.data(), n, s, 8, nullptr, out);
}

extern "C" int32_t jolt_eq_evals_aligned_block(jolt_ctx* ctx, const jolt_fr_t* r, size_t n, size_t start, size_t block, jolt_table** out) {
    if (!ctx || !out || (!r && n) || n > 40) return JOLT_ERR_INVALID_ARG;
    // eq.rs:243-245 asserts
    JOLT_REQUIRE(ctx, block != 0 && (block & (block - 1)) == 0, "block_size must be a power of two");
    JOLT_REQUIRE(ctx, start % block == 0, "start_index must be aligned to block_size");
    size_t block_vars = 0;
    while (((size_t)1 << block_vars) < block) block_vars++;
    JOLT_REQUIRE(ctx, block_vars <= n, "block larger than the domain");
    std::vector<Fr> pt;
    JOLT_TRY(read_point(ctx, r, n, pt));
    size_t prefix_len = n - block_vars;
    size_t prefix_value = start >> block_vars;
    Fr prefix_scale = Fr::one();
    for (size_t pos = 0; pos < prefix_len; ++pos) {  // eq.rs:253-260
        int bit = (int)((prefix_value >> (prefix_len - 1 - pos)) & 1);
        prefix_scale = mul(prefix_scale, bit ? pt[pos] : sub(Fr::one(), pt[pos]));
    }
    return eq_build(ctx, pt.data() + prefix_len, block_vars, prefix_scale, 8, nullptr, out);
}

extern "C" int32_t jolt_lt_evals(jolt_ctx* ctx, const jolt_fr_t* r, size_t n, jolt_table** out) {
    if (!ctx || !out || (!r && n) || n > 40) return JOLT_ERR_INVALID_ARG;
    std::vector<Fr> pt;
    JOLT_TRY(read_point(ctx, r, n, pt));
    // LT over the first `done` variables and eq over the same prefix, extended <= 8 variables at a time
    jolt_table *lt = nullptr, *eq = nullptr;
    JOLT_TRY(jolt_internal_table_new(ctx, 1, &lt));
    JOLT_TRY(jolt_internal_table_new(ctx, 1, &eq));
    hipLaunchKernelGGL(k_fill_fr, dim3(1), dim3(64), 0, ctx->stream, lt->buf[0], Fr::zero(), (size_t)1);
    hipLaunchKernelGGL(k_fill_fr, dim3(1), dim3(64), 0, ctx->stream, eq->buf[0], Fr::one(), (size_t)1);
    JOLT_HIP_TRY(ctx, hipGetLastError());
    size_t done = 0;
    while (done < n) {
        size_t c = std::min<size_t>(8, n - done);
        EqChunk ch;
        ch.c = (int)c;
        for (size_t k = 0; k < 8; ++k) ch.r[k] = k < c ? pt[done + k] : Fr::zero();
        size_t out_len = (size_t)1 << (done + c);
        jolt_table *lt2 = nullptr, *eq2 = nullptr;
        JOLT_TRY(jolt_internal_table_new(ctx, out_len, &lt2));
        hipLaunchKernelGGL(k_lt_expand, dim3(sweep_grid(ctx, out_len)), dim3(kBlock), 0, ctx->stream, (const Fr*)lt->data(), (const Fr*)eq->data(), lt2->data(), out_len, ch);
        JOLT_HIP_TRY(ctx, hipGetLastError());
        if (done + c < n) {
            JOLT_TRY(jolt_internal_table_new(ctx, out_len, &eq2));
            hipLaunchKernelGGL(k_eq_expand, dim3(sweep_grid(ctx, out_len)), dim3(kBlock), 0, ctx->stream, (const Fr*)eq->data(), eq2->data(), out_len, ch);
            JOLT_HIP_TRY(ctx, hipGetLastError());
        }
        jolt_table_free(ctx, lt);
        jolt_table_free(ctx, eq);
        lt = lt2;
        eq = eq2;
        done += c;
    }
    if (eq) jolt_table_free(ctx, eq);
    *out = lt;
    return JOLT_OK;
}

extern "C" int32_t jolt_eq_plus_one_evals(jolt_ctx* ctx, const jolt_fr_t* r, size_t n, const jolt_fr_t* scale, jolt_table** eq_out,
                                          jolt_table** eqp1_out) {
    if (!ctx || !eq_out || !eqp1_out || (!r && n) || n > 32) return JOLT_ERR_INVALID_ARG;
    std::vector<Fr> pt;
    JOLT_TRY(read_point(ctx, r, n, pt));
    Fr s = scale ? fr_from_abi(scale) : Fr::one();
    std::vector<jolt_table*> levels;
    jolt_table* full = nullptr;
    JOLT_TRY(eq_build(ctx, pt.data(), n, s, 1, &levels, &full));
    jolt_table* p1 = nullptr;
    size_t len = (size_t)1 << n;
    JOLT_TRY(jolt_internal_table_new(ctx, len, &p1));
    EqP1Args a;
    a.n = (int)n;
    for (size_t i = 0; i < 33; ++i) a.prefix[i] = i < levels.size() ? levels[i]->data() : nullptr;
    for (size_t i = 0; i < 32; ++i) a.lower[i] = Fr::zero();
    for (size_t i = 0; i < n; ++i) {  // eq_plus_one.rs:97-102
        Fr lower = Fr::one();
        for (size_t m = i + 1; m < n; ++m) lower = mul(lower, pt[m]);
        a.lower[i] = mul(lower, sub(Fr::one(), pt[i]));
    }
    hipLaunchKernelGGL(k_eq_plus_one, dim3(sweep_grid(ctx, len)), dim3(kBlock), 0, ctx->stream, a, p1->data(), len);
    JOLT_HIP_TRY(ctx, hipGetLastError());
    for (size_t i = 0; i + 1 < levels.size(); ++i) jolt_table_free(ctx, levels[i]);
    *eq_out = full;
    *eqp1_out = p1;
    return JOLT_OK;
}

extern "C" int32_t jolt_table_sum(jolt_ctx* ctx, const jolt_table* t, jolt_fr_t* out) {
    if (!ctx || !t || !out) return JOLT_ERR_INVALID_ARG;
    int grid = sweep_grid(ctx, t->len);
    JOLT_TRY(jolt_internal_ensure_scratch(ctx, (size_t)grid, 1));
    hipLaunchKernelGGL(k_sum_or_dot<false>, dim3(grid), dim3(kBlock), 0, ctx->stream, (const Fr*)t->data(), (const Fr*)nullptr, t->len, ctx->d_partials);
    JOLT_HIP_TRY(ctx, hipGetLastError());
    JOLT_TRY(reduce_into_results(ctx, grid, 1, 0));
    return fetch_results(ctx, 1, out);
}

extern "C" int32_t jolt_table_evaluate(jolt_ctx* ctx, const jolt_table* t, const jolt_fr_t* point, size_t n, jolt_fr_t* out) {
    if (!ctx || !t || !out || (!point && n)) return JOLT_ERR_INVALID_ARG;
    if (t->len != ((size_t)1 << n)) return JOLT_ERR_SIZE_MISMATCH;  // dense.rs:341-345 assert
    jolt_table* eq = nullptr;
    JOLT_TRY(jolt_eq_evals(ctx, point, n, nullptr, &eq));
    int grid = sweep_grid(ctx, t->len);
    JOLT_TRY(jolt_internal_ensure_scratch(ctx, (size_t)grid, 1));
    hipLaunchKernelGGL(k_sum_or_dot<true>, dim3(grid), dim3(kBlock), 0, ctx->stream, (const Fr*)t->data(), (const Fr*)eq->data(), t->len, ctx->d_partials);
    JOLT_HIP_TRY(ctx, hipGetLastError());
    JOLT_TRY(reduce_into_results(ctx, grid, 1, 0));
    int32_t s = fetch_results(ctx, 1, out);
    jolt_table_free(ctx, eq);
    return s;
}

// ------------------------------------------------------------------------------------------------------------------
// sumcheck members
// ------------------------------------------------------------------------------------------------------------------
static int32_t member_upload_desc(jolt_member* m) {
    jolt_ctx* ctx = m->ctx;
    JOLT_TRY(jolt_internal_dev_alloc(ctx, sizeof(MemberDesc), (void**)&m->d_desc));
    // m->desc lives as long as the member and is not modified after this point: no synchronisation needed for the host source
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(m->d_desc, &m->desc, sizeof(MemberDesc), hipMemcpyHostToDevice, ctx->stream));
    return JOLT_OK;
}

static int32_t member_common_init(jolt_ctx* ctx, jolt_table* const* tables, uint32_t n_tables, jolt_member* m, bool borrow = false) {
    if (n_tables == 0 || n_tables > kMaxBatchTables) return JOLT_ERR_UNSUPPORTED;
    size_t len = tables[0]->len;
    for (uint32_t i = 0; i < n_tables; ++i) {
        if (!tables[i]) return JOLT_ERR_INVALID_ARG;
        if (tables[i]->len != len) return JOLT_ERR_SIZE_MISMATCH;  // KernelError::TableSizeMismatch, naive.rs:145-155
    }
    if (len == 0 || (len & (len - 1)) != 0) return JOLT_ERR_SIZE_MISMATCH;
    m->ctx = ctx;
    m->len = len;
    m->rounds = 0;
    while (((size_t)1 << m->rounds) < len) m->rounds++;
    m->borrowed = borrow;
    if (!borrow) {
        m->tables.assign(tables, tables + n_tables);
        return JOLT_OK;
    }
    // borrowed: the member reads the caller's tables and binds into its own scratch, so the same resident table can
    // serve several members/stages (and the PCS opening later) without re-materialisation
    for (uint32_t i = 0; i < n_tables; ++i) {
        jolt_table* v = new (std::nothrow) jolt_table();
        if (!v) return JOLT_ERR_OOM;
        v->ctx = ctx;
        v->cur = -1;
        v->view = tables[i]->data();
        v->view_len = len;
        v->len = len;
        m->tables.push_back(v);
    }
    return JOLT_OK;
}

extern "C" int32_t jolt_member_create_lc(jolt_ctx* ctx, jolt_table* const* tables, const jolt_member_lc_desc* d, jolt_member** out) {
    if (!ctx || !tables || !d || !out) return JOLT_ERR_INVALID_ARG;
    if (d->n_groups > kMaxGroups || d->n_factors > kMaxFactors || d->n_lc > kMaxLc || d->degree < 1 || d->degree > JOLT_MAX_DEGREE)
        return JOLT_ERR_UNSUPPORTED;
    if (d->order != JOLT_ORDER_LOW_TO_HIGH && d->order != JOLT_ORDER_HIGH_TO_LOW) return JOLT_ERR_INVALID_ARG;
    jolt_member* m = new (std::nothrow) jolt_member();
    if (!m) return JOLT_ERR_OOM;
    const bool borrow = (d->flags & JOLT_MEMBER_FLAG_BORROW_TABLES) != 0;
    int32_t s = member_common_init(ctx, tables, d->n_tables, m, borrow);
    if (s != JOLT_OK) { if (!borrow) m->tables.clear(); jolt_member_destroy(m); return s; }
    m->kind = jolt_member::kExpr;
    m->degree = d->degree;
    m->order = d->order;
    m->skip_one = (d->flags & JOLT_MEMBER_FLAG_SKIP_ONE) != 0;
    MemberDesc& md = m->desc;
    std::memset(&md, 0, sizeof(md));
    md.n_groups = d->n_groups;
    md.n_factors = d->n_factors;
    md.n_lc = d->n_lc;
    bool ok = d->group_factor_offsets[0] == 0 && d->group_factor_offsets[d->n_groups] == d->n_factors && d->factor_lc_offsets[0] == 0 &&
              d->factor_lc_offsets[d->n_factors] == d->n_lc;
    for (uint32_t g = 0; ok && g <= d->n_groups; ++g) {
        md.grp_fac_off[g] = d->group_factor_offsets[g];
        if (g && md.grp_fac_off[g] < md.grp_fac_off[g - 1]) ok = false;
        if (g && md.grp_fac_off[g] - md.grp_fac_off[g - 1] > d->degree) ok = false;  // a product of more factors than the degree bound
    }
    for (uint32_t f = 0; ok && f <= d->n_factors; ++f) {
        md.fac_lc_off[f] = d->factor_lc_offsets[f];
        if (f && md.fac_lc_off[f] < md.fac_lc_off[f - 1]) ok = false;
    }
    for (uint32_t f = 0; ok && f < d->n_factors; ++f) {
        Fr c = d->factor_consts ? fr_from_abi(&d->factor_consts[f]) : Fr::zero();
        if (!fr_is_canonical(c)) ok = false;
        md.fac_const[f] = c;
        md.fac_has_const[f] = c.is_zero() ? 0u : 1u;
    }
    for (uint32_t k = 0; ok && k < d->n_lc; ++k) {
        if (d->lc_tables[k] >= d->n_tables) { ok = false; break; }
        md.lc_tab[k] = d->lc_tables[k];
        Fr c = fr_from_abi(&d->lc_coeffs[k]);
        if (!fr_is_canonical(c)) ok = false;
        md.lc_coeff[k] = c;
        md.lc_one[k] = (c == Fr::one()) ? 1u : 0u;
        md.lc_owner[k] = 1u;
        for (uint32_t j = 0; j < k; ++j) if (d->lc_tables[j] == d->lc_tables[k]) { md.lc_owner[k] = 0u; break; }
    }
    {   // multiplies per pair of the round kernel: 2 per non-unit LC coefficient + (factors - 1) per evaluation point
        const size_t ne = m->skip_one ? d->degree : d->degree + 1;
        size_t muls = 0;
        for (uint32_t g = 0; ok && g < d->n_groups; ++g) {
            uint32_t nf = md.grp_fac_off[g + 1] - md.grp_fac_off[g];
            if (nf > 1) muls += (size_t)(nf - 1) * ne;
            for (uint32_t f = md.grp_fac_off[g]; f < md.grp_fac_off[g + 1]; ++f)
                for (uint32_t k = md.fac_lc_off[f]; k < md.fac_lc_off[f + 1]; ++k) muls += md.lc_one[k] ? 0 : 2;
        }
        m->muls_per_pair = muls;
    }
    // every table must be mentioned by the summand: a fused round binds a table through its owner entry
    for (uint32_t t = 0; ok && t < d->n_tables; ++t) {
        bool used = false;
        for (uint32_t k = 0; k < d->n_lc; ++k) used = used || d->lc_tables[k] == t;
        m->all_tables_used = m->all_tables_used && used;
    }
    if (!ok) { if (!borrow) m->tables.clear(); jolt_member_destroy(m); ctx->last_error = "malformed member descriptor"; return JOLT_ERR_INVALID_ARG; }
    s = member_upload_desc(m);
    if (s != JOLT_OK) { if (!borrow) m->tables.clear(); jolt_member_destroy(m); return s; }
    *out = m;
    return JOLT_OK;
}

extern "C" int32_t jolt_member_create_expr(jolt_ctx* ctx, jolt_table* const* tables, const jolt_member_desc* d, jolt_member** out) {
    if (!ctx || !tables || !d || !out) return JOLT_ERR_INVALID_ARG;
    if (d->n_terms > kMaxGroups || d->n_terms > JOLT_MAX_MEMBER_TERMS) return JOLT_ERR_UNSUPPORTED;
    // flat Expr -> LC form: term k is group k, every factor a one-entry LC; the term coefficient rides on the first
    // factor (exact: coefficient * prod factors), a factor-less term is a constant factor.
    std::vector<uint32_t> goff{0}, foff{0}, ltab;
    std::vector<jolt_fr_t> fconst, lcoef;
    jolt_fr_t one_abi, zero_abi;
    Fr one = Fr::one(), zero = Fr::zero();
    fr_to_abi(&one_abi, one);
    fr_to_abi(&zero_abi, zero);
    for (uint32_t k = 0; k < d->n_terms; ++k) {
        uint32_t a = d->term_offsets[k], b = d->term_offsets[k + 1];
        if (b < a || b > JOLT_MAX_MEMBER_FACTORS) return JOLT_ERR_UNSUPPORTED;
        if (a == b) {
            fconst.push_back(d->coeffs[k]);
            foff.push_back((uint32_t)ltab.size());
        } else {
            for (uint32_t f = a; f < b; ++f) {
                ltab.push_back(d->factors[f]);
                lcoef.push_back(f == a ? d->coeffs[k] : one_abi);
                fconst.push_back(zero_abi);
                foff.push_back((uint32_t)ltab.size());
            }
        }
        goff.push_back((uint32_t)fconst.size());
    }
    jolt_member_lc_desc lc;
    lc.n_tables = d->n_tables;
    lc.n_groups = d->n_terms;
    lc.n_factors = (uint32_t)fconst.size();
    lc.n_lc = (uint32_t)ltab.size();
    lc.degree = d->degree;
    lc.order = d->order;
    lc.flags = 0;
    lc.group_factor_offsets = goff.data();
    lc.factor_lc_offsets = foff.data();
    lc.factor_consts = fconst.data();
    uint32_t dummy_tab = 0;
    lc.lc_tables = ltab.empty() ? &dummy_tab : ltab.data();
    lc.lc_coeffs = lcoef.empty() ? &one_abi : lcoef.data();
    return jolt_member_create_lc(ctx, tables, &lc, out);
}

static int32_t init_split_eq(jolt_ctx* ctx, jolt_member* m, const jolt_fr_t* w, size_t n, const jolt_fr_t* scale, const jolt_fr_t* shard_scale);

// eq(w, j) * q(j) with q in "sum of products of linear combinations" form: the eq weight is factored out as in the optimized tier
// (GruenRoundMessage, crates/jolt-kernels/src/optimized/support.rs:340-412; consumers instruction_input.rs, instruction_claim_reduction.rs):
// no T-sized eq table is read or bound, each round returns q(0), q(2), .., q(dq) (dq = d->degree, the INNER degree; s(1) is
// recovered from the claim), the host assembles s = l * q with gruen_poly_from_q.  LowToHigh only.
extern "C" int32_t jolt_member_create_split_eq_lc(jolt_ctx* ctx, jolt_table* const* tables, const jolt_member_lc_desc* d, const jolt_fr_t* w, size_t n,
                                                  const jolt_fr_t* scale, const jolt_fr_t* shard_scale, jolt_member** out) {
    if (!ctx || !tables || !d || (!w && n) || !out) return JOLT_ERR_INVALID_ARG;
    if (d->order != JOLT_ORDER_LOW_TO_HIGH || n == 0 || d->degree + 1 > JOLT_MAX_DEGREE) return JOLT_ERR_UNSUPPORTED;
    jolt_member_lc_desc inner = *d;
    inner.flags |= JOLT_MEMBER_FLAG_SKIP_ONE;
    jolt_member* m = nullptr;
    JOLT_TRY(jolt_member_create_lc(ctx, tables, &inner, &m));
    int32_t s = m->rounds == n ? JOLT_OK : JOLT_ERR_SIZE_MISMATCH;
    if (s == JOLT_OK) {
        m->eq_weighted = true;
        m->muls_per_pair += (size_t)d->n_groups * d->degree + 1;  // the row weight
        s = init_split_eq(ctx, m, w, n, scale, shard_scale);
    }
    if (s != JOLT_OK) { if (m->borrowed) { /* views only */ } else m->tables.clear(); jolt_member_destroy(m); return s; }
    *out = m;
    return JOLT_OK;
}

// What the integer round kernel needs to know about a descriptor (small_round.hip.h): which tables are integer columns, the R-scaled coefficients, and which
// product groups are integer groups (every factor ONE integer column without a constant: evaluated exactly in integers, coefficient = product of its entries')
static void build_small_desc(const MemberDesc& md, const std::vector<bool>& is_int, SmallDesc& sd) {
    std::memset(&sd, 0, sizeof(sd));
    const Fr r2 = Fr::r2();
    for (size_t t = 0; t < is_int.size() && t < (size_t)kMaxBatchTables; ++t) sd.tab_int[t] = is_int[t] ? 1u : 0u;
    for (uint32_t k = 0; k < md.n_lc; ++k) sd.lc_coeff_rr[k] = mul(md.lc_coeff[k], r2);
    for (uint32_t g = 0; g < md.n_groups; ++g) {
        const uint32_t f0 = md.grp_fac_off[g], f1 = md.grp_fac_off[g + 1];
        bool ok = f1 - f0 == 1 || f1 - f0 == 2;
        Fr coeff = Fr::one();
        for (uint32_t f = f0; ok && f < f1; ++f) {
            const uint32_t k0 = md.fac_lc_off[f], k1 = md.fac_lc_off[f + 1];
            if (md.fac_has_const[f] || k1 - k0 != 1 || !is_int[md.lc_tab[k0]]) ok = false;
            else if (!md.lc_one[k0]) coeff = mul(coeff, md.lc_coeff[k0]);
        }
        if (!ok) continue;
        sd.grp_int[g] = 1;
        sd.grp_coeff_rr[g] = mul(coeff, r2);
        sd.n_int_groups += 1;
    }
}

// jolt_member_create_lc / jolt_member_create_split_eq_lc (w != NULL) over tables some of which are still resident u64 witness columns: the compact-scalar
// polynomials of the optimized tier (Polynomial<T>, crates/jolt-poly/src/dense.rs:129-142 bind_to_field; products through FrSmallScalarAccumulator,
// crates/jolt-field/src/bn254/mont.rs:343-427).  Slot i is tables[i] (a field table, borrowed) or ints[i] (JOLT_INT_U64, borrowed: it must outlive the member).
// Round 0 reads the integers (8 bytes per entry), the first bind writes field tables; from then on the member is an ordinary one.  Same round sums, bit for bit.
extern "C" int32_t jolt_member_create_lc_small(jolt_ctx* ctx, jolt_table* const* tables, const jolt_ints* const* ints, const jolt_member_lc_desc* d, const jolt_fr_t* w,
                                               size_t n, const jolt_fr_t* scale, const jolt_fr_t* shard_scale, jolt_member** out) {
    if (!ctx || !tables || !ints || !d || !out) return JOLT_ERR_INVALID_ARG;
    if (d->order != JOLT_ORDER_LOW_TO_HIGH || d->n_tables == 0 || d->n_tables > (uint32_t)kMaxBatchTables) return JOLT_ERR_UNSUPPORTED;
    std::vector<jolt_table> stubs(d->n_tables);
    std::vector<jolt_table*> slots(d->n_tables);
    std::vector<bool> is_int(d->n_tables, false);
    bool any = false;
    for (uint32_t i = 0; i < d->n_tables; ++i) {
        if ((tables[i] != nullptr) == (ints[i] != nullptr)) { ctx->last_error = "every slot is either a field table or an integer column"; return JOLT_ERR_INVALID_ARG; }
        if (tables[i]) { slots[i] = tables[i]; continue; }
        if (ints[i]->kind != JOLT_INT_U64) { ctx->last_error = "integer-backed member tables are u64 columns (promote i64 / i128 columns with jolt_table_from_ints)"; return JOLT_ERR_UNSUPPORTED; }
        stubs[i].ctx = ctx;
        stubs[i].cur = -1;
        stubs[i].len = ints[i]->count;
        slots[i] = &stubs[i];
        is_int[i] = any = true;
    }
    jolt_member_lc_desc dd = *d;
    dd.flags |= JOLT_MEMBER_FLAG_BORROW_TABLES;
    jolt_member* m = nullptr;
    if (w) JOLT_TRY(jolt_member_create_split_eq_lc(ctx, slots.data(), &dd, w, n, scale, shard_scale, &m));
    else JOLT_TRY(jolt_member_create_lc(ctx, slots.data(), &dd, &m));
    if (!any) { *out = m; return JOLT_OK; }
    // Rounds with few pairs run the tail kernel, which reads field tables only: such members promote their columns once, here.
    const bool eager = m->len / 2 <= std::max<size_t>(ctx->tail_pairs, (size_t)1 << 12);
    int32_t st = JOLT_OK;
    for (uint32_t i = 0; st == JOLT_OK && i < d->n_tables; ++i) {
        if (!is_int[i]) continue;
        jolt_table* v = m->tables[i];
        if (eager) {
            Fr* buf = nullptr;
            st = jolt_internal_dev_alloc(ctx, m->len * sizeof(Fr), (void**)&buf);
            if (st != JOLT_OK) break;
            m->promoted.push_back(buf);
            hipLaunchKernelGGL(k_from_u64, dim3(sweep_grid(ctx, m->len)), dim3(kBlock), 0, ctx->stream, reinterpret_cast<const uint64_t*>(ints[i]->data), buf, m->len);
            if (hipGetLastError() != hipSuccess) st = JOLT_ERR_HIP;
            v->view = buf;
        } else {
            v->ints = v->ints_src = ints[i]->data;
        }
    }
    if (st == JOLT_OK && !eager) {
        m->h_small = new (std::nothrow) SmallDesc();
        if (!m->h_small) st = JOLT_ERR_OOM;
        if (st == JOLT_OK) {
            build_small_desc(m->desc, is_int, *m->h_small);
            st = jolt_internal_dev_alloc(ctx, sizeof(SmallDesc), (void**)&m->d_small);
        }
        if (st == JOLT_OK && hipMemcpyAsync(m->d_small, m->h_small, sizeof(SmallDesc), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) st = JOLT_ERR_HIP;
    }
    if (st != JOLT_OK) { jolt_member_destroy(m); return st; }
    *out = m;
    return JOLT_OK;
}

// The per-pair evaluation of the integer round kernel on the HOST (small_round.hip.h: the same small_pair_eval, the same descriptor analysis), for the CPU suite:
// one LowToHigh pair of a member in jolt_member_lc_desc form; slot i is an integer column iff is_int[i] (then int_pairs[2 i], [2 i + 1] = its lo / hi entries, else
// fr_pairs[2 i], [2 i + 1]).  out[s], s < n_evals: the summand at the points 0, 1, 2, .. (skip_one: 0, 2, 3, ..).
extern "C" int32_t jolt_host_small_round_pair(const jolt_member_lc_desc* d, const uint8_t* is_int, const uint64_t* int_pairs, const jolt_fr_t* fr_pairs, uint32_t n_evals,
                                              int32_t skip_one, jolt_fr_t* out) {
    if (!d || !is_int || !int_pairs || !fr_pairs || !out || n_evals < 1 || n_evals > 4) return JOLT_ERR_INVALID_ARG;
    if (d->n_groups > (uint32_t)kMaxGroups || d->n_factors > (uint32_t)kMaxFactors || d->n_lc > (uint32_t)kMaxLc || d->n_tables > (uint32_t)kMaxBatchTables) return JOLT_ERR_UNSUPPORTED;
    MemberDesc md;
    std::memset(&md, 0, sizeof(md));
    md.n_groups = d->n_groups; md.n_factors = d->n_factors; md.n_lc = d->n_lc;
    for (uint32_t g = 0; g <= d->n_groups; ++g) md.grp_fac_off[g] = d->group_factor_offsets[g];
    for (uint32_t f = 0; f <= d->n_factors; ++f) md.fac_lc_off[f] = d->factor_lc_offsets[f];
    for (uint32_t f = 0; f < d->n_factors; ++f) {
        md.fac_const[f] = d->factor_consts ? fr_from_abi(&d->factor_consts[f]) : Fr::zero();
        md.fac_has_const[f] = md.fac_const[f].is_zero() ? 0u : 1u;
    }
    for (uint32_t k = 0; k < d->n_lc; ++k) {
        if (d->lc_tables[k] >= d->n_tables) return JOLT_ERR_INVALID_ARG;
        md.lc_tab[k] = d->lc_tables[k];
        md.lc_coeff[k] = fr_from_abi(&d->lc_coeffs[k]);
        md.lc_one[k] = md.lc_coeff[k] == Fr::one() ? 1u : 0u;
    }
    std::vector<bool> mask(d->n_tables);
    for (uint32_t i = 0; i < d->n_tables; ++i) mask[i] = is_int[i] != 0;
    SmallDesc sd;
    build_small_desc(md, mask, sd);
    auto ldf = [&](uint32_t ti, Fr& lo, Fr& hi) { lo = fr_from_abi(&fr_pairs[2 * ti]); hi = fr_from_abi(&fr_pairs[2 * ti + 1]); };
    auto ldi = [&](uint32_t ti, uint64_t& lo, uint64_t& hi) { lo = int_pairs[2 * ti]; hi = int_pairs[2 * ti + 1]; };
    auto run = [&](auto ne_tag, auto skip_tag) {
        constexpr int NE = decltype(ne_tag)::value;
        constexpr bool SK = decltype(skip_tag)::value;
        Fr o[NE];
        small_pair_eval<NE, SK>(&md, &sd, ldf, ldi, o);
        for (int s2 = 0; s2 < NE; ++s2) fr_to_abi(&out[s2], o[s2]);
    };
    auto with_ne = [&](auto skip_tag) {
        switch (n_evals) {
            case 1: run(std::integral_constant<int, 1>{}, skip_tag); break;
            case 2: run(std::integral_constant<int, 2>{}, skip_tag); break;
            case 3: run(std::integral_constant<int, 3>{}, skip_tag); break;
            default: run(std::integral_constant<int, 4>{}, skip_tag); break;
        }
    };
    if (skip_one) with_ne(std::true_type{});
    else with_ne(std::false_type{});
    return JOLT_OK;
}

// GruenSplitEqPolynomial::new_with_scaling(w, LowToHigh, scale) (split_eq.rs:187-236): head = w[..n-1], out_point = head[..split],
// in_point = rest; evals_cached -> one device table per prefix length.  shard_scale multiplies the E_out tables.
static int32_t init_split_eq(jolt_ctx* ctx, jolt_member* m, const jolt_fr_t* w, size_t n, const jolt_fr_t* scale, const jolt_fr_t* shard_scale) {
    JOLT_TRY(read_point(ctx, w, n, m->w));
    m->current_scalar = scale ? fr_from_abi(scale) : Fr::one();
    m->initial_scalar = m->current_scalar;
    if (n > 0) {
        size_t split = n / 2, head_len = n - 1;
        m->out_len = std::min(split, head_len);
        m->in_len = head_len - m->out_len;
        jolt_table* last = nullptr;
        Fr e_out_scale = shard_scale ? fr_from_abi(shard_scale) : Fr::one();
        JOLT_TRY(eq_build(ctx, m->w.data(), m->out_len, e_out_scale, 1, &m->e_out_cache, &last));
        JOLT_TRY(eq_build(ctx, m->w.data() + m->out_len, m->in_len, Fr::one(), 1, &m->e_in_cache, &last));
        m->e_out_bits = m->out_len;
        m->e_in_bits = m->in_len;
    }
    return JOLT_OK;
}

extern "C" int32_t jolt_member_create_split_eq_uniform(jolt_ctx* ctx, jolt_table* const* tables, uint32_t V, uint32_t F, const jolt_fr_t* coeffs,
                                                       const jolt_fr_t* w, size_t n, const jolt_fr_t* scale, const jolt_fr_t* shard_scale, uint32_t flags,
                                                       jolt_member** out) {
    if (!ctx || !tables || !coeffs || (!w && n) || !out) return JOLT_ERR_INVALID_ARG;
    if (F < 2 || F > 4 || V < 1 || V > (uint32_t)kMaxGroups || (size_t)V * F > (size_t)kMaxBatchTables || n == 0) return JOLT_ERR_UNSUPPORTED;
    const bool borrow = (flags & JOLT_MEMBER_FLAG_BORROW_TABLES) != 0;
    jolt_member* m = new (std::nothrow) jolt_member();
    if (!m) return JOLT_ERR_OOM;
    int32_t s = member_common_init(ctx, tables, V * F, m, borrow);
    if (s == JOLT_OK && m->rounds != n) s = JOLT_ERR_SIZE_MISMATCH;
    if (s == JOLT_OK) {
        m->kind = jolt_member::kSplitEqUniform;
        m->degree = F + 1;
        m->order = JOLT_ORDER_LOW_TO_HIGH;
        m->uni_V = V;
        m->uni_F = F;
        for (uint32_t v = 0; v < V && s == JOLT_OK; ++v) {
            Fr c = fr_from_abi(&coeffs[v]);
            if (!fr_is_canonical(c)) s = JOLT_ERR_INVALID_ARG;
            m->uni_coeff.push_back(c);
        }
    }
    if (s == JOLT_OK) s = init_split_eq(ctx, m, w, n, scale, shard_scale);
    if (s != JOLT_OK) { if (!borrow) m->tables.clear(); jolt_member_destroy(m); return s; }
    *out = m;
    return JOLT_OK;
}

// eq(w, j) * sum_v c_v * prod_{i<F} ra_{vF+i}(j) with ra_p(j) = scale_tables[p][index(p, j)]: the RA-virtualization summand over
// lazily bound one-hot selector columns (crates/jolt-kernels/src/optimized/lazy_ra.rs:55-182; consumers
// optimized/{ram,instruction}_ra_virtualization.rs).  The selector columns are never materialised at T entries: the first four
// rounds gather through the hot indices, the fourth bind writes them dense at T/16.
// shared by the two lazily bound members: `rho` != NULL selects the booleanity summand (then V, F, coeffs are ignored)
static int32_t create_lazy_member(jolt_ctx* ctx, const jolt_onehot* source, const jolt_fr_t* scale_tables, uint32_t V, uint32_t F, const jolt_fr_t* coeffs,
                                  const jolt_fr_t* rho, const jolt_fr_t* w, size_t n, const jolt_fr_t* scale, const jolt_fr_t* shard_scale, jolt_member** out);

extern "C" int32_t jolt_member_create_lazy_ra_uniform(jolt_ctx* ctx, const jolt_onehot* source, const jolt_fr_t* scale_tables, uint32_t V, uint32_t F,
                                                      const jolt_fr_t* coeffs, const jolt_fr_t* w, size_t n, const jolt_fr_t* scale, jolt_member** out) {
    if (!coeffs) return JOLT_ERR_INVALID_ARG;
    if (F < 2 || F > 4 || V < 1 || V > (uint32_t)kMaxGroups || (size_t)V * F > (size_t)kMaxBatchTables) return JOLT_ERR_UNSUPPORTED;
    if (source && source->n_polys != (size_t)V * F) return JOLT_ERR_SIZE_MISMATCH;
    return create_lazy_member(ctx, source, scale_tables, V, F, coeffs, nullptr, w, n, scale, nullptr, out);
}
// One shard of a hypercube-sharded batch: `source` holds this rank's block of cycles, w its n LOCAL
// coordinates, shard_scale = eq(w_hi, rank) multiplies E_out -- as jolt_member_create_split_eq_product_sharded.
extern "C" int32_t jolt_member_create_lazy_ra_uniform_sharded(jolt_ctx* ctx, const jolt_onehot* source, const jolt_fr_t* scale_tables, uint32_t V, uint32_t F,
                                                              const jolt_fr_t* coeffs, const jolt_fr_t* w, size_t n, const jolt_fr_t* scale,
                                                              const jolt_fr_t* shard_scale, jolt_member** out) {
    if (!coeffs) return JOLT_ERR_INVALID_ARG;
    if (F < 2 || F > 4 || V < 1 || V > (uint32_t)kMaxGroups || (size_t)V * F > (size_t)kMaxBatchTables) return JOLT_ERR_UNSUPPORTED;
    if (source && source->n_polys != (size_t)V * F) return JOLT_ERR_SIZE_MISMATCH;
    return create_lazy_member(ctx, source, scale_tables, V, F, coeffs, nullptr, w, n, scale, shard_scale, out);
}

// eq(w, j) * sum_i (H_i(j)^2 - rho[i] * H_i(j)), H_i(j) = scale_tables[i][index(i, j)]: the booleanity cycle-phase summand over the
// gamma-pre-scaled address-folded selector columns (crates/jolt-kernels/src/optimized/booleanity.rs:436-633), bound lazily like
// the RA-virtualization member.  Two round sums (constant and leading coefficient of the inner quadratic) -> gruen_poly_deg_3.
extern "C" int32_t jolt_member_create_lazy_booleanity(jolt_ctx* ctx, const jolt_onehot* source, const jolt_fr_t* scale_tables, const jolt_fr_t* rho,
                                                      const jolt_fr_t* w, size_t n, const jolt_fr_t* scale, jolt_member** out) {
    if (!rho) return JOLT_ERR_INVALID_ARG;
    if (source && source->n_polys > (size_t)kMaxBatchTables) return JOLT_ERR_UNSUPPORTED;
    return create_lazy_member(ctx, source, scale_tables, 0, 0, nullptr, rho, w, n, scale, nullptr, out);
}

static int32_t create_lazy_member(jolt_ctx* ctx, const jolt_onehot* source, const jolt_fr_t* scale_tables, uint32_t V, uint32_t F, const jolt_fr_t* coeffs,
                                  const jolt_fr_t* rho, const jolt_fr_t* w, size_t n, const jolt_fr_t* scale, const jolt_fr_t* shard_scale, jolt_member** out) {
    if (!ctx || !source || !scale_tables || (!w && n) || !out) return JOLT_ERR_INVALID_ARG;
    const bool booleanity = rho != nullptr;
    if (n < 4 || ((size_t)1 << n) != source->cycles) return JOLT_ERR_SIZE_MISMATCH;  // dense from the fourth bind on
    const size_t N = source->n_polys, K = source->k;
    jolt_member* m = new (std::nothrow) jolt_member();
    if (!m) return JOLT_ERR_OOM;
    m->ctx = ctx;
    m->kind = booleanity ? jolt_member::kSplitEqBooleanity : jolt_member::kSplitEqUniform;
    m->rounds = n;
    m->len = source->cycles;
    m->degree = booleanity ? 3 : F + 1;
    m->order = JOLT_ORDER_LOW_TO_HIGH;
    m->uni_V = V;
    m->uni_F = F;
    m->onehot = source;
    m->lazy_width = 1;
    int32_t s = JOLT_OK;
    for (uint32_t v = 0; v < V && s == JOLT_OK; ++v) {
        Fr c = fr_from_abi(&coeffs[v]);
        if (!fr_is_canonical(c)) s = JOLT_ERR_INVALID_ARG;
        m->uni_coeff.push_back(c);
    }
    for (size_t i = 0; booleanity && i < N && s == JOLT_OK; ++i) {
        Fr c = fr_from_abi(&rho[i]);
        if (!fr_is_canonical(c)) s = JOLT_ERR_INVALID_ARG;
        m->bool_rho.push_back(c);
    }
    std::vector<Fr> host_tables(N * K);
    for (size_t i = 0; i < N * K && s == JOLT_OK; ++i) {
        host_tables[i] = fr_from_abi(&scale_tables[i]);
        if (!fr_is_canonical(host_tables[i])) s = JOLT_ERR_INVALID_ARG;
    }
    // fold c_v into the scale table of product v's first factor: the round kernels then need no coefficient multiply; the
    // reported final values of those columns are multiplied back by 1 / c_v (exact: same canonical value)
    if (s == JOLT_OK && !booleanity) {
        bool all_invertible = true;
        for (uint32_t v = 0; v < V; ++v) all_invertible = all_invertible && !m->uni_coeff[v].is_zero();
        if (all_invertible) {
            m->uni_prescaled = true;
            m->final_unscale.assign(N, Fr::one());
            for (uint32_t v = 0; v < V; ++v) {
                if (m->uni_coeff[v] == Fr::one()) continue;
                for (size_t k = 0; k < K; ++k) host_tables[(size_t)v * F * K + k] = mul(host_tables[(size_t)v * F * K + k], m->uni_coeff[v]);
                m->final_unscale[(size_t)v * F] = inv(m->uni_coeff[v]);
            }
        }
    }
    // first-round pair tables (F = 4, K <= 16, coefficients folded in): P[v][h][a*17+b] = T_{4v+2h}[a] * T_{4v+2h+1}[b], row/column 16 = 0
    std::vector<Fr> host_pair;
    if (s == JOLT_OK && !booleanity && F == 4 && K <= 16 && m->uni_prescaled) {
        host_pair.assign((size_t)V * 2 * 289, Fr::zero());
        for (uint32_t v = 0; v < V; ++v)
            for (int h = 0; h < 2; ++h) {
                const Fr* T0 = &host_tables[((size_t)v * 4 + 2 * h) * K];
                const Fr* T1 = &host_tables[((size_t)v * 4 + 2 * h + 1) * K];
                Fr* P = &host_pair[((size_t)v * 2 + h) * 289];
                for (size_t x = 0; x < K; ++x)
                    for (size_t y = 0; y < K; ++y) P[x * 17 + y] = mul(T0[x], T1[y]);
            }
    }
    // dense targets of the fourth bind (cycles/16 entries each); until then `len` is bookkeeping only
    for (size_t p = 0; p < N && s == JOLT_OK; ++p) {
        jolt_table* t = nullptr;
        s = jolt_internal_table_new(ctx, source->cycles / 16, &t);
        if (s == JOLT_OK) { t->len = source->cycles; m->tables.push_back(t); }
    }
    if (s == JOLT_OK) {
        auto palloc = [&](void** p, size_t bytes) { return jolt_internal_dev_alloc(ctx, bytes, p) == JOLT_OK ? hipSuccess : hipErrorOutOfMemory; };
        hipError_t e = palloc((void**)&m->d_base, N * K * sizeof(Fr));
        if (e == hipSuccess) e = palloc((void**)&m->d_branch[0], N * 16 * K * sizeof(Fr));
        if (e == hipSuccess) e = palloc((void**)&m->d_branch[1], N * 16 * K * sizeof(Fr));
        if (e == hipSuccess) e = hipMemcpyAsync(m->d_base, host_tables.data(), N * K * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && !host_pair.empty()) e = palloc((void**)&m->d_pair, host_pair.size() * sizeof(Fr));
        if (e == hipSuccess && !host_pair.empty()) e = hipMemcpyAsync(m->d_pair, host_pair.data(), host_pair.size() * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(m->d_branch[0], m->d_base, N * K * sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // the host buffer may be short-lived
        if (e != hipSuccess) { ctx->last_error = std::string("lazy member: ") + hipGetErrorString(e); s = e == hipErrorOutOfMemory ? JOLT_ERR_OOM : JOLT_ERR_HIP; }
    }
    if (s == JOLT_OK) s = init_split_eq(ctx, m, w, n, scale, shard_scale);
    if (s != JOLT_OK) { jolt_member_destroy(m); return s; }
    *out = m;
    return JOLT_OK;
}

// One LowToHigh bind of a lazily bound member (LazyFoldedRa::bind, lazy_ra.rs:153-182): double the branch tables; at the fourth
// bind gather every column dense at cycles/16 and leave the lazy state.  Device work only -- member_note_bind keeps the books.
static int32_t lazy_bind_enqueue(jolt_member* m, const Fr& c) {
    jolt_ctx* ctx = m->ctx;
    const jolt_onehot* src = m->onehot;
    const size_t N = src->n_polys, K = src->k;
    const uint32_t width = m->lazy_width;
    const size_t per_poly = (size_t)width * K;
    const Fr* in = m->d_branch[m->branch_cur];
    Fr* outb = m->d_branch[1 - m->branch_cur];
    const size_t work = per_poly * N;
    hipLaunchKernelGGL(k_onehot_double_branches, dim3((unsigned)((work + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, in, outb, per_poly, N, c,
                       fr_low_limbs_zero(c) ? 1 : 0);
    JOLT_HIP_TRY(ctx, hipGetLastError());
    m->branch_cur = 1 - m->branch_cur;
    if (width < 8) {
        m->lazy_width = width * 2;
        for (jolt_table* t : m->tables) t->len /= 2;
        return JOLT_OK;
    }
    const uint32_t branches = 16;
    const size_t new_len = src->cycles / branches;
    for (size_t base = 0; base < N; base += kMaxBatchTables) {
        size_t cnt = std::min<size_t>(kMaxBatchTables, N - base);
        OneHotDense o;
        for (size_t i = 0; i < (size_t)kMaxBatchTables; ++i) o.out[i] = i < cnt ? m->tables[base + i]->buf[0] : nullptr;
        hipLaunchKernelGGL(k_onehot_materialize, dim3((unsigned)((new_len + kBlock - 1) / kBlock), (unsigned)cnt), dim3(kBlock), 0, ctx->stream,
                           (const Fr*)outb, (size_t)branches * K, (const uint8_t*)src->idx, src->cycles, branches, (uint32_t)K, base, o, src->wide);
    }
    JOLT_HIP_TRY(ctx, hipGetLastError());
    for (jolt_table* t : m->tables) { t->cur = 0; t->len = new_len; }
    m->lazy_width = 0;
    return JOLT_OK;
}

static int32_t create_split_eq_product(jolt_ctx* ctx, jolt_table* a, jolt_table* b, const jolt_fr_t* w, size_t n, const jolt_fr_t* scale,
                                       bool borrow, jolt_member** out, const jolt_fr_t* shard_scale = nullptr) {
    if (!ctx || !a || !b || (!w && n) || !out) return JOLT_ERR_INVALID_ARG;
    jolt_member* m = new (std::nothrow) jolt_member();
    if (!m) return JOLT_ERR_OOM;
    jolt_table* tabs[2] = {a, b};
    int32_t s = member_common_init(ctx, tabs, 2, m, borrow);
    if (s == JOLT_OK && m->rounds != n) s = JOLT_ERR_SIZE_MISMATCH;
    if (s != JOLT_OK) { if (!borrow) m->tables.clear(); jolt_member_destroy(m); return s; }
    m->kind = jolt_member::kSplitEqProduct;
    m->degree = 3;
    m->order = JOLT_ORDER_LOW_TO_HIGH;
    s = init_split_eq(ctx, m, w, n, scale, shard_scale);
    if (s != JOLT_OK) { if (!borrow) m->tables.clear(); jolt_member_destroy(m); return s; }
    *out = m;
    return JOLT_OK;
}
extern "C" int32_t jolt_member_create_split_eq_product(jolt_ctx* ctx, jolt_table* a, jolt_table* b, const jolt_fr_t* w, size_t n,
                                                       const jolt_fr_t* scale, jolt_member** out) {
    return create_split_eq_product(ctx, a, b, w, n, scale, false, out);
}
extern "C" int32_t jolt_member_create_split_eq_product_borrowed(jolt_ctx* ctx, jolt_table* a, jolt_table* b, const jolt_fr_t* w, size_t n,
                                                                const jolt_fr_t* scale, jolt_member** out) {
    return create_split_eq_product(ctx, a, b, w, n, scale, true, out);
}

extern "C" int32_t jolt_member_create_split_eq_product_sharded(jolt_ctx* ctx, jolt_table* a, jolt_table* b, const jolt_fr_t* w, size_t n,
                                                               const jolt_fr_t* scale, const jolt_fr_t* shard_scale, jolt_member** out) {
    if (n == 0) return JOLT_ERR_INVALID_ARG;  // a shard has at least one local variable
    return create_split_eq_product(ctx, a, b, w, n, scale, true, out, shard_scale);
}

// rewind a member that borrows its tables to round 0 (no device work): re-prove with fresh challenges
extern "C" int32_t jolt_member_reset(jolt_member* m) {
    (void)jolt_internal_engine_quiesce(m ? m->ctx : nullptr);
    if (!m) return JOLT_ERR_INVALID_ARG;
    if (m->onehot) {  // lazily bound member: back to the index-encoded state with the unbound scale tables
        const size_t N = m->onehot->n_polys, K = m->onehot->k;
        JOLT_HIP_TRY(m->ctx, hipMemcpyAsync(m->d_branch[0], m->d_base, N * K * sizeof(Fr), hipMemcpyDeviceToDevice, m->ctx->stream));
        m->branch_cur = 0;
        m->lazy_width = 1;
        for (jolt_table* t : m->tables) { t->cur = 0; t->len = m->onehot->cycles; }
        m->len = m->onehot->cycles;
        m->bound = 0;
        m->current_scalar = m->initial_scalar;
        m->e_out_bits = m->out_len;
        m->e_in_bits = m->in_len;
        return JOLT_OK;
    }
    if (!m->borrowed) { m->ctx->last_error = "only members that borrow their tables can be reset"; return JOLT_ERR_UNSUPPORTED; }
    for (jolt_table* t : m->tables) { t->cur = -1; t->len = t->view_len; t->ints = t->ints_src; }
    m->len = m->tables[0]->view_len;
    m->bound = 0;
    m->current_scalar = m->initial_scalar;
    m->e_out_bits = m->out_len;
    m->e_in_bits = m->in_len;
    return JOLT_OK;
}

// Replace the scaling factor of a split-eq member that has not bound anything yet (GruenSplitEqPolynomial::new_with_scaling,
// crates/jolt-poly/src/split_eq.rs:180-215): the E tables do not depend on it, so a reset member can be re-used with the scalar of
// the next proof (the tail members of a sharded batch, whose scalar is the product of the shard-local challenges).
extern "C" int32_t jolt_member_set_scale(jolt_member* m, const jolt_fr_t* scale) {
    if (!m || !scale) return JOLT_ERR_INVALID_ARG;
    if (!m->has_split_eq()) return JOLT_ERR_UNSUPPORTED;
    if (m->bound != 0) { m->ctx->last_error = "set_scale on a member that has already bound a variable"; return JOLT_ERR_INVALID_ARG; }
    Fr c = fr_from_abi(scale);
    JOLT_REQUIRE(m->ctx, fr_is_canonical(c), "scale is not a canonical Fr");
    m->initial_scalar = c;
    m->current_scalar = c;
    return JOLT_OK;
}

extern "C" int32_t jolt_member_num_rounds(const jolt_member* m, size_t* rounds) {
    if (!m || !rounds) return JOLT_ERR_INVALID_ARG;
    *rounds = m->rounds;
    return JOLT_OK;
}
extern "C" int32_t jolt_member_degree(const jolt_member* m, uint32_t* degree) {
    if (!m || !degree) return JOLT_ERR_INVALID_ARG;
    *degree = m->degree + (m->eq_weighted ? 1u : 0u);  // message degree
    return JOLT_OK;
}

// naive.rs:211-219 bind_tables (+ split_eq.rs:334-350 for the split-eq member): host-side state only; the table
// binds themselves are enqueued by the caller (grouped over members)
static int32_t member_note_bind(jolt_member* m, const Fr& c) {
    if (m->bound >= m->rounds) { m->ctx->last_error = "member already fully bound"; return JOLT_ERR_INVALID_ARG; }
    if (m->has_split_eq()) {
        size_t n = m->rounds;
        size_t current_index = n - m->bound;
        Fr p = m->w[current_index - 1];
        Fr prod = mul(p, c);
        Fr f = add(add(sub(sub(Fr::one(), p), c), prod), prod);
        m->current_scalar = mul(m->current_scalar, f);
        current_index -= 1;
        if (n / 2 < current_index && m->e_in_bits > 0) m->e_in_bits -= 1;
        else if (0 < current_index && m->e_out_bits > 0) m->e_out_bits -= 1;
    }
    m->len /= 2;
    m->bound += 1;
    return JOLT_OK;
}
static int32_t lazy_bind_enqueue(jolt_member* m, const Fr& c);
static int32_t member_bind(jolt_member* m, const Fr& c) {
    JOLT_TRY(member_note_bind(m, c));
    if (m->lazy_width) {
        JOLT_TRY(jolt_internal_engine_quiesce(m->ctx));
        return lazy_bind_enqueue(m, c);
    }
    return jolt_internal_bind(m->ctx, m->tables.data(), m->tables.size(), c, m->order);
}

size_t jolt_internal_member_n_evals(const jolt_member* m) {
    if (m->kind == jolt_member::kSplitEqProduct || m->kind == jolt_member::kSplitEqBooleanity) return 2;
    if (m->kind == jolt_member::kSplitEqUniform) return m->uni_F;
    return m->skip_one ? m->degree : m->degree + 1;
}

template <int ORDER, bool SKIP1, bool FUSED>
static void launch_round_group(int ne, dim3 grid, hipStream_t s, const RoundGroupArgs& a, const Fr& r, int shifted, Fr* partials, const RoundDone& rd) {
    switch (ne) {
        case 1: hipLaunchKernelGGL((k_round_evals_group<1, ORDER, SKIP1, FUSED>), grid, dim3(kBlock), 0, s, a, r, shifted, partials, rd); break;
        case 2: hipLaunchKernelGGL((k_round_evals_group<2, ORDER, SKIP1, FUSED>), grid, dim3(kBlock), 0, s, a, r, shifted, partials, rd); break;
        case 3: hipLaunchKernelGGL((k_round_evals_group<3, ORDER, SKIP1, FUSED>), grid, dim3(kBlock), 0, s, a, r, shifted, partials, rd); break;
        case 4: hipLaunchKernelGGL((k_round_evals_group<4, ORDER, SKIP1, FUSED>), grid, dim3(kBlock), 0, s, a, r, shifted, partials, rd); break;
        case 5: hipLaunchKernelGGL((k_round_evals_group<5, ORDER, SKIP1, FUSED>), grid, dim3(kBlock), 0, s, a, r, shifted, partials, rd); break;
        case 6: hipLaunchKernelGGL((k_round_evals_group<6, ORDER, SKIP1, FUSED>), grid, dim3(kBlock), 0, s, a, r, shifted, partials, rd); break;
        case 7: hipLaunchKernelGGL((k_round_evals_group<7, ORDER, SKIP1, FUSED>), grid, dim3(kBlock), 0, s, a, r, shifted, partials, rd); break;
        case 8: hipLaunchKernelGGL((k_round_evals_group<8, ORDER, SKIP1, FUSED>), grid, dim3(kBlock), 0, s, a, r, shifted, partials, rd); break;
    }
}

template <bool SKIP1>
static void launch_round_small(int ne, dim3 grid, hipStream_t s, const RoundGroupArgs& a, const SmallGroupArgs& sa, Fr* partials, const RoundDone& rd) {
    switch (ne) {
        case 1: hipLaunchKernelGGL((k_round_evals_small<1, SKIP1>), grid, dim3(kBlock), 0, s, a, sa, partials, rd); break;
        case 2: hipLaunchKernelGGL((k_round_evals_small<2, SKIP1>), grid, dim3(kBlock), 0, s, a, sa, partials, rd); break;
        case 3: hipLaunchKernelGGL((k_round_evals_small<3, SKIP1>), grid, dim3(kBlock), 0, s, a, sa, partials, rd); break;
        case 4: hipLaunchKernelGGL((k_round_evals_small<4, SKIP1>), grid, dim3(kBlock), 0, s, a, sa, partials, rd); break;
        case 5: hipLaunchKernelGGL((k_round_evals_small<5, SKIP1>), grid, dim3(kBlock), 0, s, a, sa, partials, rd); break;
        case 6: hipLaunchKernelGGL((k_round_evals_small<6, SKIP1>), grid, dim3(kBlock), 0, s, a, sa, partials, rd); break;
        case 7: hipLaunchKernelGGL((k_round_evals_small<7, SKIP1>), grid, dim3(kBlock), 0, s, a, sa, partials, rd); break;
        case 8: hipLaunchKernelGGL((k_round_evals_small<8, SKIP1>), grid, dim3(kBlock), 0, s, a, sa, partials, rd); break;
    }
}

// Enqueue one batch round for n members.  Pending binds of LowToHigh members are FUSED into the round kernels (one pass
// over memory per round: read the unbound table, write the bound one, accumulate the round sums); the remaining binds
// (HighToLow members, tables a summand does not mention) are grouped by challenge into ceil(tables/40) launches.
// Round sums: one launch per (NE, order, skip, fused, challenge) class with blockIdx.y = member; rounds with few pairs
// left go into the tail kernel (all members, pair x group x point work items).  The last workgroup of every member
// publishes its sums into host-mapped memory (finish_member).
// The main stream waits for the side streams that ran table-writing kernels in the previous batch round.
int32_t jolt_internal_join_side_writers(jolt_ctx* ctx) {
    for (int k = 0; k < 3; ++k) {
        if (!ctx->join_pending[k]) continue;
        JOLT_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join[k], 0));
        ctx->join_pending[k] = false;
    }
    return JOLT_OK;
}

static int32_t group_enqueue(jolt_ctx* ctx, jolt_member* const* members, size_t n, const Fr* const* binds) {
    using eclk = std::chrono::steady_clock;
    static double eacc[4] = {0, 0, 0, 0};
    static size_t ecalls = 0;
    const bool etrace = ctx->round_trace;
    eclk::time_point e0, e1, e2, e3;
    if (etrace) e0 = eclk::now();
    const size_t kTailPairs = ctx->tail_pairs;
    const size_t kUniformRowsMajorPairs = ctx->uniform_rows_pairs;
    JOLT_TRY(jolt_internal_engine_quiesce(ctx));  // also joins the side streams that wrote tables last round
    if (n > (size_t)kGroupTicket) { ctx->last_error = "batch round has too many members"; return JOLT_ERR_UNSUPPORTED; }
    {   // validate BEFORE anything is bound: an error below must leave every member exactly as it was (a caller may fall back)
        size_t sums = 0;
        for (size_t i = 0; i < n; ++i) {
            const jolt_member* m = members[i];
            if (!m) return JOLT_ERR_INVALID_ARG;
            const bool binds_now = binds && binds[i];
            if (binds_now && m->bound >= m->rounds) { ctx->last_error = "member already fully bound"; return JOLT_ERR_INVALID_ARG; }
            if ((binds_now ? m->len / 2 : m->len) < 2) { ctx->last_error = "prove_round on a fully bound member"; return JOLT_ERR_INVALID_ARG; }
            sums += jolt_internal_member_n_evals(m);
        }
        if (sums > ctx->round_cap) { ctx->last_error = "batch round returns too many sums"; return JOLT_ERR_UNSUPPORTED; }
    }
    struct Item {
        size_t ne, slot;
        bool fused = false, tail = false, done = false, rows_major = false, small = false;
        uint32_t lds_blocks = 0;   // > 0: k_split_eq_uniform_lazy_lds with this many workgroups per product group (booleanity: per column group)
        uint32_t bool_cols = 0;    // > 0: k_split_eq_booleanity_lds with this many columns per group
        Fr r;                      // challenge of the fused bind
        std::vector<const Fr*> in; // table pointers the round kernel reads
        std::vector<Fr*> out;      // fused: where the bound tables go
        int grid = 1;
        uint32_t part_off = 0;
    };
    std::vector<Item> items(n);
    // ---- (1) binds: fused where possible, grouped launches otherwise
    struct BindGroup { Fr r; int32_t order; std::vector<jolt_table*> tabs; };
    std::vector<BindGroup> bgs;
    for (size_t i = 0; i < n; ++i) {
        jolt_member* m = members[i];
        Item& it = items[i];
        if (binds && binds[i]) {
            // Fuse only where it pays: the separate bind kernel runs at the HBM roofline with its multiplies hidden, so moving
            // them into an ALU-bound round kernel (many multiplies per table) costs more than the saved pass; fuse the
            // bandwidth-bound members (<= 2 multiplies per table per pair) and never the latency-bound tail rounds.
            const bool can_fuse = !m->ints_live() && m->order == JOLT_ORDER_LOW_TO_HIGH && (m->len / 4 > kTailPairs || (ctx->fuse_tail && m->kind == jolt_member::kExpr)) &&
                                  (m->kind == jolt_member::kSplitEqProduct || (m->kind == jolt_member::kExpr && m->all_tables_used && m->muls_per_pair <= ctx->fuse_ratio * m->tables.size()));
            JOLT_TRY(member_note_bind(m, *binds[i]));  // m->len is now the bound length
            if (can_fuse) {
                it.fused = true;
                it.r = *binds[i];
                for (jolt_table* t : m->tables) {
                    size_t half = t->len / 2;
                    JOLT_TRY(jolt_internal_table_ensure_alt(t, half));
                    it.in.push_back(t->data());
                    Fr* o = t->buf[t->cur < 0 ? 0 : 1 - t->cur];
                    it.out.push_back(o);
                    t->cur = t->cur < 0 ? 0 : 1 - t->cur;  // the round kernel fills it
                    t->len = half;
                }
            } else if (m->lazy_width) {
                JOLT_TRY(lazy_bind_enqueue(m, *binds[i]));  // index-encoded selector columns: re-scale the branch tables / materialise
            } else {
                BindGroup* g = nullptr;
                for (BindGroup& c : bgs) if (c.order == m->order && c.r == *binds[i]) { g = &c; break; }
                if (!g) { bgs.push_back(BindGroup{*binds[i], m->order, {}}); g = &bgs.back(); }
                g->tabs.insert(g->tabs.end(), m->tables.begin(), m->tables.end());
            }
        }
    }
    if (etrace) e1 = eclk::now();
    for (BindGroup& g : bgs) JOLT_TRY(jolt_internal_bind(ctx, g.tabs.data(), g.tabs.size(), g.r, g.order));
    if (etrace) e2 = eclk::now();
    size_t slot = 0, part_total = 0;
    for (size_t i = 0; i < n; ++i) {
        jolt_member* m = members[i];
        Item& it = items[i];
        if (m->len < 2) { ctx->last_error = "prove_round on a fully bound member"; return JOLT_ERR_INVALID_ARG; }
        if (!it.fused) for (jolt_table* t : m->tables) { it.in.push_back(t->ints ? reinterpret_cast<const Fr*>(t->ints) : t->data()); it.out.push_back(nullptr); }
        it.small = m->ints_live();  // round 0 off the integer columns (small_round.hip.h): never a tail round (such members promote when they are created)
        if (it.small && (m->kind != jolt_member::kExpr || m->order != JOLT_ORDER_LOW_TO_HIGH || !m->d_small)) { ctx->last_error = "integer-backed tables in a member that cannot read them"; return JOLT_ERR_INVALID_ARG; }
        it.ne = jolt_internal_member_n_evals(m);
        it.slot = slot;
        slot += it.ne;
        it.tail = m->kind == jolt_member::kExpr && m->len / 2 <= kTailPairs && !it.small;
    }
    if (slot > ctx->round_cap) { ctx->last_error = "batch round returns too many sums"; return JOLT_ERR_UNSUPPORTED; }
    auto same_challenge = [&](const Item& a, const Item& b) { return a.fused == b.fused && (!a.fused || a.r == b.r); };
    // ---- (2a) tail launches
    struct TailLaunch { TailArgs args; int count; unsigned gx, gz; Fr r; int shifted; std::vector<size_t> who; };
    std::vector<TailLaunch> tails;
    for (size_t i = 0; i < n; ++i) {
        if (!items[i].tail || items[i].done) continue;
        TailLaunch T;
        T.count = 0; T.gx = 1; T.gz = 1;
        T.r = items[i].fused ? items[i].r : Fr::zero();
        T.shifted = fr_low_limbs_zero(T.r) ? 1 : 0;
        uint32_t cursor = 0;
        for (size_t j = i; j < n; ++j) {
            jolt_member* m = members[j];
            Item& it = items[j];
            if (!it.tail || it.done || !same_challenge(items[i], it)) continue;
            if (T.count == kMaxGroupMembers || cursor + m->tables.size() > (size_t)kMaxGroupTables) break;
            int c = T.count++;
            T.args.g.desc[c] = m->d_desc;
            T.args.g.half[c] = m->len / 2;
            T.args.g.tab_off[c] = cursor;
            for (size_t k = 0; k < m->tables.size(); ++k) { T.args.g.tabs[cursor] = it.in[k]; T.args.g.outs[cursor] = it.out[k]; cursor++; }
            T.args.ne[c] = (uint32_t)it.ne;
            T.args.order[c] = (uint32_t)m->order;
            T.args.skip[c] = m->skip_one ? 1u : 0u;
            T.args.fused[c] = it.fused ? 1u : 0u;
            T.args.g.ticket[c] = (uint32_t)j;
            T.args.g.slot[c] = (uint32_t)it.slot;
            T.args.g.e_out[c] = m->eq_weighted ? m->e_out_cache[m->e_out_bits]->data() : nullptr;
            T.args.g.e_in[c] = m->eq_weighted ? m->e_in_cache[m->e_in_bits]->data() : nullptr;
            T.args.g.in_bits[c] = (int32_t)m->e_in_bits;
            size_t work = (m->len / 2) * std::max<uint32_t>(1, m->desc.n_groups);
            T.gx = std::max<unsigned>(T.gx, (unsigned)std::min<size_t>((work + kBlock - 1) / kBlock, 256));
            T.gz = std::max<unsigned>(T.gz, (unsigned)it.ne);
            T.who.push_back(j);
            it.done = true;
        }
        for (int c = 0; c < T.count; ++c) {
            Item& it = items[T.who[c]];
            it.grid = (int)T.gx;
            it.part_off = (uint32_t)part_total;
            T.args.g.part_off[c] = (uint32_t)part_total;
            part_total += (size_t)T.gx * it.ne;
        }
        tails.push_back(std::move(T));
    }
    // ---- (2b) class launches of the remaining expr members
    struct Launch { RoundGroupArgs args; SmallGroupArgs sargs; int ne, order, skip, fused, count, shifted, small; unsigned grid; Fr r; std::vector<size_t> who; };
    std::vector<Launch> launches;
    for (size_t i = 0; i < n; ++i) {
        jolt_member* m = members[i];
        if (items[i].done || m->kind != jolt_member::kExpr) continue;
        Launch L;
        L.ne = (int)items[i].ne; L.order = m->order; L.skip = m->skip_one ? 1 : 0; L.fused = items[i].fused ? 1 : 0; L.count = 0; L.grid = 1;
        L.small = items[i].small ? 1 : 0;
        L.r = items[i].fused ? items[i].r : Fr::zero();
        L.shifted = fr_low_limbs_zero(L.r) ? 1 : 0;
        uint32_t cursor = 0;
        for (size_t j = i; j < n; ++j) {
            jolt_member* mj = members[j];
            Item& it = items[j];
            if (it.done || mj->kind != jolt_member::kExpr) continue;
            if ((int)it.ne != L.ne || mj->order != L.order || (mj->skip_one ? 1 : 0) != L.skip || !same_challenge(items[i], it) || (it.small ? 1 : 0) != L.small) continue;
            if (L.count == kMaxGroupMembers || cursor + mj->tables.size() > (size_t)kMaxGroupTables) break;
            int c = L.count++;
            L.args.desc[c] = mj->d_desc;
            L.args.half[c] = mj->len / 2;
            L.args.tab_off[c] = cursor;
            for (size_t k = 0; k < mj->tables.size(); ++k) { L.args.tabs[cursor] = it.in[k]; L.args.outs[cursor] = it.out[k]; cursor++; }
            L.args.ticket[c] = (uint32_t)j;
            L.args.slot[c] = (uint32_t)it.slot;
            L.args.e_out[c] = mj->eq_weighted ? mj->e_out_cache[mj->e_out_bits]->data() : nullptr;
            L.args.e_in[c] = mj->eq_weighted ? mj->e_in_cache[mj->e_in_bits]->data() : nullptr;
            L.args.in_bits[c] = (int32_t)mj->e_in_bits;
            L.sargs.sd[c] = mj->d_small;
            // the integer round kernel walks pairs (all groups of a pair inside one item), the field one (group, pair) items
            L.grid = std::max<unsigned>(L.grid, (unsigned)round_grid(ctx, (mj->len / 2) * (L.small ? 1u : std::max<uint32_t>(1, mj->desc.n_groups))));
            L.who.push_back(j);
            it.done = true;
        }
        for (int c = 0; c < L.count; ++c) {
            Item& it = items[L.who[c]];
            it.grid = (int)L.grid;
            it.part_off = (uint32_t)part_total;
            L.args.part_off[c] = (uint32_t)part_total;
            part_total += (size_t)L.grid * it.ne;
        }
        launches.push_back(std::move(L));
    }
    for (size_t i = 0; i < n; ++i) {
        if (members[i]->kind == jolt_member::kExpr) continue;
        size_t work = members[i]->len / 2;
        // uniform members: one item per pair (the V products inside) while the pairs alone fill the chip, (v, pair) items below
        if (members[i]->kind == jolt_member::kSplitEqUniform) {
            items[i].rows_major = work >= kUniformRowsMajorPairs && members[i]->uni_V > 1;
            if (!items[i].rows_major) work *= members[i]->uni_V;
        }
        items[i].grid = round_grid(ctx, work);
        // index-encoded selector columns past the first bind: one product group per workgroup, its branch tables in LDS
        if (members[i]->kind == jolt_member::kSplitEqUniform && members[i]->lazy_width >= 2 && !members[i]->onehot->wide && lazy_lds_bytes(members[i]) <= kLazyLdsMax && ctx->lazy_lds) {
            items[i].rows_major = false;
            items[i].lds_blocks = (uint32_t)std::max<size_t>(1, (size_t)round_grid(ctx, (members[i]->len / 2) * members[i]->uni_V) / members[i]->uni_V);
            items[i].grid = (int)(items[i].lds_blocks * members[i]->uni_V);
        }
        // the booleanity member's index-encoded rounds: column groups with their branch tables in LDS (k_split_eq_booleanity_lds; JOLT_BOOL_LDS=0: global gathers)
        if (members[i]->kind == jolt_member::kSplitEqBooleanity && members[i]->lazy_width >= 1 && members[i]->lazy_width <= 8 && members[i]->onehot && !members[i]->onehot->wide &&
            ctx->lazy_lds && bool_lds_on()) {
            const size_t per_col = (size_t)members[i]->lazy_width * ((size_t)members[i]->onehot->k + 1) * sizeof(Fr), n_cols = members[i]->tables.size();
            const size_t cpg = std::min(n_cols, kLazyLdsMax / per_col);
            if (cpg >= 1) {
                const size_t groups = (n_cols + cpg - 1) / cpg;
                items[i].bool_cols = (uint32_t)((n_cols + groups - 1) / groups);  // even groups
                items[i].lds_blocks = (uint32_t)std::max<size_t>(1, (size_t)round_grid(ctx, (members[i]->len / 2) * groups) / groups);
                items[i].grid = (int)(items[i].lds_blocks * groups);
            }
        }
        items[i].part_off = (uint32_t)part_total;
        part_total += (size_t)items[i].grid * items[i].ne;
    }
    JOLT_TRY(jolt_internal_ensure_scratch(ctx, part_total + 8, slot + 8));
    RoundDone rd;
    rd.counters = ctx->d_counters;
    rd.results = ctx->h_round;
    rd.results_dev = ctx->d_round;
    rd.flag = ctx->h_flag;
    rd.seq = ++ctx->seq;
    rd.group_total = (uint32_t)n;
    if (etrace) e3 = eclk::now();
    // fork: the round's independent kernels go round-robin over main + side streams, all ordered after the binds above
    int n_kernels = (int)tails.size() + (int)launches.size();
    for (size_t i = 0; i < n; ++i) if (members[i]->kind != jolt_member::kExpr) n_kernels++;
    int rr = 0;
    hipStream_t streams[4] = {ctx->stream, ctx->side[0], ctx->side[1], ctx->side[2]};
    const int n_streams = n_kernels > 1 && !ctx->serial_streams ? std::min(4, n_kernels) : 1;
    if (n_streams > 1) JOLT_HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    bool wrote[4] = {false, false, false, false};
    bool forked[4] = {true, false, false, false};
    auto next_stream = [&](bool writes_tables = false) {
        int k = rr % n_streams;
        rr++;
        if (writes_tables) wrote[k] = true;
        if (!forked[k]) {  // a side stream waits for the binds only when it gets its first kernel: the round's first (longest) kernel
            (void)hipStreamWaitEvent(streams[k], ctx->ev_fork, 0);  // goes to the main stream before any of these calls
            forked[k] = true;
        }
        return streams[k];
    };
    // longest kernels first so that they overlap with the short ones
    for (size_t i = 0; i < n; ++i) {
        jolt_member* m = members[i];
        if (m->kind != jolt_member::kSplitEqUniform) continue;
        const Item& it = items[i];
        UniformArgs ua;
        ua.V = (int)m->uni_V;
        for (size_t k = 0; k < (size_t)kMaxBatchTables; ++k) ua.tabs[k] = k < it.in.size() ? it.in[k] : nullptr;
        for (size_t v = 0; v < (size_t)kMaxGroups; ++v) {
            ua.coeff[v] = v < m->uni_coeff.size() ? (m->uni_prescaled ? Fr::one() : m->uni_coeff[v]) : Fr::zero();
            ua.coeff_one[v] = v < m->uni_coeff.size() && (m->uni_prescaled || m->uni_coeff[v] == Fr::one()) ? 1u : 0u;
        }
        const Fr* e_out = m->e_out_cache[m->e_out_bits]->data();
        const Fr* e_in = m->e_in_cache[m->e_in_bits]->data();
        dim3 g(it.grid), b(kBlock);
        Fr* part = ctx->d_partials + it.part_off;
        hipStream_t st = next_stream();
        if (m->lazy_width) {  // selector columns still index-encoded: gather instead of loading dense pairs
            LazyArgs la;
            la.idx = m->onehot->idx;
            la.wide = m->onehot->wide;
            la.branch = m->d_branch[m->branch_cur];
            la.cycles0 = m->onehot->cycles;
            la.width = m->lazy_width;
            la.K = m->onehot->k;
            la.V = ua.V;
            for (size_t v = 0; v < (size_t)kMaxGroups; ++v) { la.coeff[v] = ua.coeff[v]; la.coeff_one[v] = ua.coeff_one[v]; }
            if (m->lazy_width == 1 && m->d_pair && m->uni_F == 4 && !m->onehot->wide) {  // unbound columns: quadratic halves from the pair tables, no multiplies
                LazyPairArgs pa;
                pa.idx = m->onehot->idx;
                pa.pair = m->d_pair;
                pa.cycles0 = m->onehot->cycles;
                pa.V = ua.V;
                hipLaunchKernelGGL(k_split_eq_uniform_lazy_first, g, b, 0, st, pa, e_out, e_in, (int)m->e_in_bits, m->len / 2, part, (uint32_t)i, (uint32_t)it.slot, rd);
            } else if (it.lds_blocks) {
                const size_t lds = lazy_lds_bytes(m);
                if (m->uni_F == 2) hipLaunchKernelGGL(k_split_eq_uniform_lazy_lds<2>, g, b, lds, st, la, it.lds_blocks, e_out, e_in, (int)m->e_in_bits, m->len / 2, part, (uint32_t)i, (uint32_t)it.slot, rd);
                else if (m->uni_F == 3) hipLaunchKernelGGL(k_split_eq_uniform_lazy_lds<3>, g, b, lds, st, la, it.lds_blocks, e_out, e_in, (int)m->e_in_bits, m->len / 2, part, (uint32_t)i, (uint32_t)it.slot, rd);
                else hipLaunchKernelGGL(k_split_eq_uniform_lazy_lds<4>, g, b, lds, st, la, it.lds_blocks, e_out, e_in, (int)m->e_in_bits, m->len / 2, part, (uint32_t)i, (uint32_t)it.slot, rd);
            } else if (it.rows_major) {
                if (m->uni_F == 2) hipLaunchKernelGGL(k_split_eq_uniform_lazy_rows<2>, g, b, 0, st, la, e_out, e_in, (int)m->e_in_bits, m->len / 2, part, (uint32_t)i, (uint32_t)it.slot, rd);
                else if (m->uni_F == 3) hipLaunchKernelGGL(k_split_eq_uniform_lazy_rows<3>, g, b, 0, st, la, e_out, e_in, (int)m->e_in_bits, m->len / 2, part, (uint32_t)i, (uint32_t)it.slot, rd);
                else hipLaunchKernelGGL(k_split_eq_uniform_lazy_rows<4>, g, b, 0, st, la, e_out, e_in, (int)m->e_in_bits, m->len / 2, part, (uint32_t)i, (uint32_t)it.slot, rd);
            } else if (m->uni_F == 2) hipLaunchKernelGGL(k_split_eq_uniform_lazy<2>, g, b, 0, st, la, e_out, e_in, (int)m->e_in_bits, m->len / 2, part, (uint32_t)i, (uint32_t)it.slot, rd);
            else if (m->uni_F == 3) hipLaunchKernelGGL(k_split_eq_uniform_lazy<3>, g, b, 0, st, la, e_out, e_in, (int)m->e_in_bits, m->len / 2, part, (uint32_t)i, (uint32_t)it.slot, rd);
            else hipLaunchKernelGGL(k_split_eq_uniform_lazy<4>, g, b, 0, st, la, e_out, e_in, (int)m->e_in_bits, m->len / 2, part, (uint32_t)i, (uint32_t)it.slot, rd);
            JOLT_HIP_TRY(ctx, hipGetLastError());
            continue;
        }
        if (it.rows_major) {
            if (m->uni_F == 2) hipLaunchKernelGGL(k_split_eq_uniform_rows<2>, g, b, 0, st, ua, e_out, e_in, (int)m->e_in_bits, m->len / 2, part, (uint32_t)i, (uint32_t)it.slot, rd);
            else if (m->uni_F == 3) hipLaunchKernelGGL(k_split_eq_uniform_rows<3>, g, b, 0, st, ua, e_out, e_in, (int)m->e_in_bits, m->len / 2, part, (uint32_t)i, (uint32_t)it.slot, rd);
            else hipLaunchKernelGGL(k_split_eq_uniform_rows<4>, g, b, 0, st, ua, e_out, e_in, (int)m->e_in_bits, m->len / 2, part, (uint32_t)i, (uint32_t)it.slot, rd);
            JOLT_HIP_TRY(ctx, hipGetLastError());
            continue;
        }
        if (m->uni_F == 2) hipLaunchKernelGGL(k_split_eq_uniform<2>, g, b, 0, st, ua, e_out, e_in, (int)m->e_in_bits, m->len / 2, part, (uint32_t)i, (uint32_t)it.slot, rd);
        else if (m->uni_F == 3) hipLaunchKernelGGL(k_split_eq_uniform<3>, g, b, 0, st, ua, e_out, e_in, (int)m->e_in_bits, m->len / 2, part, (uint32_t)i, (uint32_t)it.slot, rd);
        else hipLaunchKernelGGL(k_split_eq_uniform<4>, g, b, 0, st, ua, e_out, e_in, (int)m->e_in_bits, m->len / 2, part, (uint32_t)i, (uint32_t)it.slot, rd);
        JOLT_HIP_TRY(ctx, hipGetLastError());
    }
    for (TailLaunch& T : tails) {
        hipLaunchKernelGGL(k_round_evals_tail, dim3(T.gx, (unsigned)T.count, T.gz), dim3(kBlock), 0, next_stream(ctx->fuse_tail), T.args, T.r, T.shifted, ctx->d_partials, rd);
        JOLT_HIP_TRY(ctx, hipGetLastError());
    }
    for (Launch& L : launches) {
        dim3 grid(L.grid, (unsigned)L.count);
        hipStream_t lst = next_stream(L.fused);
        if (L.small) {
            if (L.skip) launch_round_small<true>(L.ne, grid, lst, L.args, L.sargs, ctx->d_partials, rd);
            else launch_round_small<false>(L.ne, grid, lst, L.args, L.sargs, ctx->d_partials, rd);
        } else if (L.order == JOLT_ORDER_LOW_TO_HIGH) {
            if (L.fused) {
                if (L.skip) launch_round_group<0, true, true>(L.ne, grid, lst, L.args, L.r, L.shifted, ctx->d_partials, rd);
                else launch_round_group<0, false, true>(L.ne, grid, lst, L.args, L.r, L.shifted, ctx->d_partials, rd);
            } else {
                if (L.skip) launch_round_group<0, true, false>(L.ne, grid, lst, L.args, L.r, L.shifted, ctx->d_partials, rd);
                else launch_round_group<0, false, false>(L.ne, grid, lst, L.args, L.r, L.shifted, ctx->d_partials, rd);
            }
        } else {
            if (L.skip) launch_round_group<1, true, false>(L.ne, grid, lst, L.args, L.r, L.shifted, ctx->d_partials, rd);
            else launch_round_group<1, false, false>(L.ne, grid, lst, L.args, L.r, L.shifted, ctx->d_partials, rd);
        }
        JOLT_HIP_TRY(ctx, hipGetLastError());
    }
    for (size_t i = 0; i < n; ++i) {
        jolt_member* m = members[i];
        if (m->kind != jolt_member::kSplitEqBooleanity) continue;
        const Item& it = items[i];
        BooleanityArgs ba;
        ba.n = (int)m->tables.size();
        for (size_t k = 0; k < (size_t)kMaxBatchTables; ++k) {
            ba.tabs[k] = k < it.in.size() ? it.in[k] : nullptr;
            ba.rho[k] = k < m->bool_rho.size() ? m->bool_rho[k] : Fr::zero();
        }
        ba.idx = m->onehot ? m->onehot->idx : nullptr;
        ba.wide = m->onehot ? m->onehot->wide : 0u;
        ba.branch = m->d_branch[m->branch_cur];
        ba.cycles0 = m->onehot ? m->onehot->cycles : 0;
        ba.width = m->lazy_width;
        ba.K = m->onehot ? m->onehot->k : 0;
        const Fr* e_out = m->e_out_cache[m->e_out_bits]->data();
        const Fr* e_in = m->e_in_cache[m->e_in_bits]->data();
        hipStream_t bst = next_stream();
        if (m->lazy_width && it.bool_cols)
            hipLaunchKernelGGL(k_split_eq_booleanity_lds, dim3(it.grid), dim3(kBlock), (size_t)it.bool_cols * m->lazy_width * ((size_t)m->onehot->k + 1) * sizeof(Fr), bst, ba,
                               it.bool_cols, it.lds_blocks, e_out, e_in, (int)m->e_in_bits, m->len / 2, ctx->d_partials + it.part_off, (uint32_t)i, (uint32_t)it.slot, rd);
        else if (m->lazy_width)
            hipLaunchKernelGGL(k_split_eq_booleanity<true>, dim3(it.grid), dim3(kBlock), 0, bst, ba, e_out, e_in, (int)m->e_in_bits, m->len / 2,
                               ctx->d_partials + it.part_off, (uint32_t)i, (uint32_t)it.slot, rd);
        else
            hipLaunchKernelGGL(k_split_eq_booleanity<false>, dim3(it.grid), dim3(kBlock), 0, bst, ba, e_out, e_in, (int)m->e_in_bits, m->len / 2,
                               ctx->d_partials + it.part_off, (uint32_t)i, (uint32_t)it.slot, rd);
        JOLT_HIP_TRY(ctx, hipGetLastError());
    }
    for (size_t i = 0; i < n; ++i) {
        jolt_member* m = members[i];
        if (m->kind != jolt_member::kSplitEqProduct) continue;
        const Item& it = items[i];
        const Fr* e_out = m->e_out_cache[m->e_out_bits]->data();
        const Fr* e_in = m->e_in_cache[m->e_in_bits]->data();
        Fr r = it.fused ? it.r : Fr::zero();
        int shifted = fr_low_limbs_zero(r) ? 1 : 0;
        hipStream_t sst = next_stream(it.fused);
        if (it.fused)
            hipLaunchKernelGGL(k_split_eq_product<true>, dim3(it.grid), dim3(kBlock), 0, sst, it.in[0], it.in[1], it.out[0], it.out[1], r, shifted,
                               e_out, e_in, (int)m->e_in_bits, m->len / 2, ctx->d_partials + it.part_off, (uint32_t)i, (uint32_t)it.slot, rd);
        else
            hipLaunchKernelGGL(k_split_eq_product<false>, dim3(it.grid), dim3(kBlock), 0, sst, it.in[0], it.in[1], (Fr*)nullptr, (Fr*)nullptr, r,
                               shifted, e_out, e_in, (int)m->e_in_bits, m->len / 2, ctx->d_partials + it.part_off, (uint32_t)i, (uint32_t)it.slot, rd);
        JOLT_HIP_TRY(ctx, hipGetLastError());
    }
    if (etrace) {
        auto e4 = eclk::now();
        auto us = [](eclk::time_point a, eclk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        eacc[0] += us(e0, e1); eacc[1] += us(e1, e2); eacc[2] += us(e2, e3); eacc[3] += us(e3, e4);
        if (++ecalls % 500 == 0)
            std::fprintf(stderr, "[jolt enqueue trace] %zu rounds: bind bookkeeping %.1f us, bind launches %.1f us, launch preparation %.1f us, fork + round launches %.1f us\n",
                         ecalls, eacc[0] / ecalls, eacc[1] / ecalls, eacc[2] / ecalls, eacc[3] / ecalls);
    }
    for (int k = 1; k < n_streams; ++k) {
        if (!wrote[k]) continue;
        JOLT_HIP_TRY(ctx, hipEventRecord(ctx->ev_join[k - 1], streams[k]));
        ctx->join_pending[k - 1] = true;
    }
    return JOLT_OK;
}

// Wait for the batch round enqueued last (spin on the host-mapped flag; fall back to a stream sync on timeout) and copy
// its `count` round sums out of the pinned buffer.
static int32_t round_wait(jolt_ctx* ctx, size_t count, jolt_fr_t* out) {
    const uint64_t want = ctx->seq;
    volatile uint64_t* flag = ctx->h_flag;
    uint64_t spins = 0;
    while (*flag != want) {
        if (++spins > (1ull << 22)) {  // ~ tens of ms: something is off, ask the runtime
            spins = 0;
            // the round's kernels run on the main stream AND the side streams: all of them must have drained
            hipError_t q = hipStreamQuery(ctx->stream);
            for (int k = 0; k < 3 && q == hipSuccess; ++k)
                if (ctx->side[k]) q = hipStreamQuery(ctx->side[k]);
            if (q == hipSuccess) {
                if (*flag == want) break;
                ctx->last_error = "batch round finished without publishing its completion flag";
                return JOLT_ERR_HIP;
            }
            if (q != hipErrorNotReady) { ctx->last_error = std::string("batch round: ") + hipGetErrorString(q); return JOLT_ERR_HIP; }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    std::memcpy(out, ctx->h_round, count * sizeof(Fr));
    ctx->d_round_count = count;  // the same sums sit in d_round (written by the publishing workgroups)
    return JOLT_OK;
}

static void member_aux(const jolt_member* m, jolt_fr_t* aux) {
    if (!aux) return;
    Fr z = Fr::zero();
    if (m->has_split_eq() && m->bound < m->rounds) {
        fr_to_abi(&aux[0], m->current_scalar);
        fr_to_abi(&aux[1], m->w[m->rounds - m->bound - 1]);
    } else {
        fr_to_abi(&aux[0], z);
        fr_to_abi(&aux[1], z);
    }
    fr_to_abi(&aux[2], z);
}

extern "C" int32_t jolt_member_prove_round(jolt_member* m, const jolt_fr_t* bind, jolt_fr_t* evals_out, size_t n_evals, jolt_fr_t* aux_out) {
    (void)jolt_internal_engine_quiesce(m ? m->ctx : nullptr);
    if (!m || !evals_out) return JOLT_ERR_INVALID_ARG;
    jolt_ctx* ctx = m->ctx;
    if (n_evals != jolt_internal_member_n_evals(m)) return JOLT_ERR_SIZE_MISMATCH;
    Fr b;
    const Fr* bp = nullptr;
    if (bind) {
        b = fr_from_abi(bind);
        JOLT_REQUIRE(ctx, fr_is_canonical(b), "bind challenge is not a canonical Fr");
        bp = &b;
    }
    JOLT_TRY(group_enqueue(ctx, &m, 1, &bp));
    member_aux(m, aux_out);
    return round_wait(ctx, n_evals, evals_out);
}

// ------------------------------------------------------------------------------------------------------------------
// Persistent round engine (engine_kernel.hip.h): host side
// ------------------------------------------------------------------------------------------------------------------
struct jolt_engine {
    bool active = false;
    std::vector<jolt_member*> members;
    int n_rounds = 0, round = 0;
    uint32_t binds_posted = 0;
    size_t total = 0;  // round sums per round
    uint64_t seq0 = 0;
    EngDesc* h_desc = nullptr;  // pinned staging
    EngDesc* d_desc = nullptr;
    EngCtl* h_ctl = nullptr;    // pinned, device-mapped
    EngSync* d_sync = nullptr;
    Fr* d_partials = nullptr;
    // policy (environment: JOLT_ENGINE=1 enables, JOLT_ENGINE_PAIRS=<n> sets the size at which a batch switches over).
    // Off by default: measured within +-1 % of the per-round kernels at T = 2^20 -- the late rounds
    // are bound by the serial multiply chain and the host<->device handshake, not by launch overhead.
    bool enabled = false, trace = false;
    size_t max_pairs = 1024;
    uint32_t task_blocks = 32, max_blocks = kEngMaxBlocks;
    // JOLT_ENGINE_TRACE: host-side split per round (ns): posted -> sums seen (device + PCIe), sums seen -> next post (host)
    std::vector<long long> t_device, t_host;
    long long t_last_seen = 0;
};

static jolt_engine* engine_get(jolt_ctx* ctx) {
    if (ctx->engine) return ctx->engine;
    jolt_engine* e = new (std::nothrow) jolt_engine();
    if (!e) return nullptr;
    const char* en = std::getenv("JOLT_ENGINE");
    e->enabled = en && en[0] == '1';
    const char* mp = std::getenv("JOLT_ENGINE_PAIRS");
    if (mp && std::atoll(mp) > 0) e->max_pairs = (size_t)std::atoll(mp);
    const char* tb = std::getenv("JOLT_ENGINE_TASK_BLOCKS");
    if (tb && std::atoi(tb) > 0) e->task_blocks = (uint32_t)std::min(64, std::atoi(tb));
    const char* mb = std::getenv("JOLT_ENGINE_MAX_BLOCKS");
    if (mb && std::atoi(mb) > 0) e->max_blocks = (uint32_t)std::min(kEngMaxBlocks, std::atoi(mb));
    const char* tr = std::getenv("JOLT_ENGINE_TRACE");
    e->trace = tr && tr[0] == '1';
    bool ok = hipHostMalloc((void**)&e->h_desc, sizeof(EngDesc), hipHostMallocDefault) == hipSuccess &&
              hipMalloc((void**)&e->d_desc, sizeof(EngDesc)) == hipSuccess &&
              hipHostMalloc((void**)&e->h_ctl, sizeof(EngCtl), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
              hipMalloc((void**)&e->d_sync, sizeof(EngSync)) == hipSuccess &&
              hipMalloc((void**)&e->d_partials, (size_t)kEngMaxSlots * kEngMaxBlocks * sizeof(Fr)) == hipSuccess;
    if (!ok) e->enabled = false;  // the per-round kernels still work
    ctx->engine = e;
    return e;
}

void jolt_internal_engine_free(jolt_ctx* ctx) {
    jolt_engine* e = ctx->engine;
    if (!e) return;
    if (e->h_desc) (void)hipHostFree(e->h_desc);
    if (e->d_desc) (void)hipFree(e->d_desc);
    if (e->h_ctl) (void)hipHostFree(e->h_ctl);
    if (e->d_sync) (void)hipFree(e->d_sync);
    if (e->d_partials) (void)hipFree(e->d_partials);
    delete e;
    ctx->engine = nullptr;
}

int32_t jolt_internal_engine_quiesce(jolt_ctx* ctx) {
    if (ctx) JOLT_TRY(jolt_internal_join_side_writers(ctx));  // every entry point that touches tables passes through here
    jolt_engine* e = ctx ? ctx->engine : nullptr;
    if (!e || !e->active) return JOLT_OK;
    __atomic_store_n(&e->h_ctl->abort, (uint64_t)1, __ATOMIC_RELEASE);
    e->active = false;
    JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return JOLT_OK;
}

// Post bind number `binds_posted` into its mailbox: challenge first, then both sequence copies.
static void engine_post(jolt_engine* e, const Fr& c) {
    EngMail& mb = e->h_ctl->mail[e->binds_posted];
    for (int k = 0; k < 8; ++k) mb.challenge[k] = c.l[k];
    __atomic_thread_fence(__ATOMIC_RELEASE);
    const uint64_t seq = (uint64_t)e->binds_posted + 1;
    __atomic_store_n(&mb.seq_b, seq, __ATOMIC_RELEASE);
    __atomic_store_n(&mb.seq_a, seq, __ATOMIC_RELEASE);
    e->binds_posted += 1;
}

// How many rounds the engine could take over from here (0 = not eligible).
static int engine_eligible(jolt_ctx* ctx, jolt_engine* e, jolt_member* const* members, size_t n, const Fr* const* binds) {
    if (!e || !e->enabled || n == 0 || n > (size_t)kEngMaxMembers) return 0;
    const bool has_bind = binds && binds[0];
    size_t tables = 0, slots = 0, tasks = 0;
    int rounds = -1;
    for (size_t i = 0; i < n; ++i) {
        const jolt_member* m = members[i];
        if (m->order != JOLT_ORDER_LOW_TO_HIGH || m->lazy_width || m->kind == jolt_member::kSplitEqBooleanity || m->eq_weighted) return 0;
        if ((binds && binds[i] != nullptr) != has_bind) return 0;
        if (has_bind && !(*binds[i] == *binds[0])) return 0;
        size_t len = has_bind ? m->len / 2 : m->len;
        if (len < 2 || len / 2 > e->max_pairs) return 0;
        int r = 0;
        while (((size_t)1 << r) < len) ++r;
        if (((size_t)1 << r) != len) return 0;
        if ((size_t)r != m->rounds - m->bound - (has_bind ? 1 : 0)) return 0;  // the member must run to its end
        if (rounds < 0) rounds = r;
        if (r != rounds) return 0;
        for (const jolt_table* t : m->tables) if (t->len != m->len || t->ints) return 0;
        size_t ne = jolt_internal_member_n_evals(m);
        tables += m->tables.size();
        slots += ne;
        tasks += m->kind == jolt_member::kExpr ? ne : 1;
        if (m->kind == jolt_member::kSplitEqUniform && (m->uni_F < 2 || m->uni_F > 4 || m->uni_V > (uint32_t)kMaxGroups)) return 0;
        if (m->kind == jolt_member::kExpr && (ne > 8 || !m->all_tables_used)) return 0;  // the fused bind writes through the summand's table references
    }
    if (rounds < 3 || rounds > kEngMaxRounds) return 0;  // not worth a launch for one or two rounds
    if (tables > (size_t)kEngMaxTables || slots > (size_t)kEngMaxSlots || tasks > (size_t)kEngMaxTasks || slots > ctx->round_cap) return 0;
    return rounds;
}

static int32_t engine_start(jolt_ctx* ctx, jolt_engine* e, jolt_member* const* members, size_t n, const Fr* const* binds, int rounds) {
    const bool has_bind = binds && binds[0];
    EngDesc& D = *e->h_desc;
    std::memset(&D, 0, sizeof(D));
    D.n_members = (int32_t)n;
    D.n_rounds = rounds;
    D.first_has_bind = has_bind ? 1 : 0;
    uint32_t tab = 0, slot = 0, task = 0;
    std::vector<size_t> work;
    for (size_t i = 0; i < n; ++i) {
        jolt_member* m = members[i];
        EngMember& M = D.m[i];
        M.kind = m->kind;
        M.n_tables = (uint32_t)m->tables.size();
        M.tab_off = tab;
        M.ne = (uint32_t)jolt_internal_member_n_evals(m);
        M.slot = slot;
        M.skip_one = m->skip_one ? 1u : 0u;
        M.desc = m->d_desc;
        for (jolt_table* t : m->tables) {
            EngTable& T = D.t[tab++];
            // capacities for the ping-pong: the first bind writes len/2 entries into the alternate buffer, the second
            // len/4 into the other one (which a borrowed view does not own yet)
            JOLT_TRY(jolt_internal_table_ensure_alt(t, t->len / 2));
            const int first = t->cur < 0 ? 0 : 1 - t->cur;
            const int second = 1 - first;
            if (t->cap[second] < std::max<size_t>(t->len / 4, 1)) {
                if (t->cur == second) { ctx->last_error = "round engine: table buffer smaller than its contents"; return JOLT_ERR_INVALID_ARG; }
                if (t->buf[second]) { jolt_internal_dev_free(ctx, t->buf[second]); t->buf[second] = nullptr; t->cap[second] = 0; }
                JOLT_TRY(jolt_internal_dev_alloc(ctx, std::max<size_t>(t->len / 4, 1) * sizeof(Fr), (void**)&t->buf[second]));
                t->cap[second] = std::max<size_t>(t->len / 4, 1);
            }
            T.src = t->data();
            T.buf[0] = t->buf[0];
            T.buf[1] = t->buf[1];
            T.first_out = (uint32_t)first;
            T.len0 = (uint32_t)t->len;
        }
        const size_t pairs0 = (has_bind ? m->len / 2 : m->len) / 2;
        if (m->kind == jolt_member::kExpr) {
            for (uint32_t t = 0; t < M.ne; ++t) {
                D.task[task] = EngTask{(uint32_t)i, t, 0, 0, slot + t, 1, (uint32_t)(pairs0 * std::max<uint32_t>(1, m->desc.n_groups))};
                D.slot_task[slot + t] = task;
                work.push_back(pairs0 * std::max<uint32_t>(1, m->desc.n_groups));
                task++;
            }
        } else {
            M.V = m->kind == jolt_member::kSplitEqUniform ? m->uni_V : 1;
            M.F = m->kind == jolt_member::kSplitEqUniform ? m->uni_F : 2;
            for (uint32_t v = 0; v < M.V && m->kind == jolt_member::kSplitEqUniform; ++v) {
                M.coeff[v] = m->uni_prescaled ? Fr::one() : m->uni_coeff[v];
                M.coeff_one[v] = (m->uni_prescaled || m->uni_coeff[v] == Fr::one()) ? 1u : 0u;
            }
            // E_out / E_in schedule (member_note_bind's bookkeeping, replayed ahead of time)
            size_t bound = m->bound, in_bits = m->e_in_bits, out_bits = m->e_out_bits;
            auto step = [&]() {
                size_t ci = m->rounds - bound - 1;
                if (m->rounds / 2 < ci && in_bits > 0) in_bits -= 1;
                else if (0 < ci && out_bits > 0) out_bits -= 1;
                bound += 1;
            };
            if (has_bind) step();
            for (int k = 0; k < rounds; ++k) {
                if (out_bits >= m->e_out_cache.size() || in_bits >= m->e_in_cache.size()) { ctx->last_error = "round engine: split-eq cache"; return JOLT_ERR_INVALID_ARG; }
                M.e_out[k] = m->e_out_cache[out_bits]->data();
                M.e_in[k] = m->e_in_cache[in_bits]->data();
                M.in_bits[k] = (int32_t)in_bits;
                step();
            }
            for (uint32_t a = 0; a < M.ne; ++a) D.slot_task[slot + a] = task;
            D.task[task] = EngTask{(uint32_t)i, 0, 0, 0, slot, M.ne, (uint32_t)(pairs0 * M.V)};
            work.push_back(pairs0 * M.V);
            task++;
        }
        slot += M.ne;
    }
    D.n_tables = (int32_t)tab;
    D.n_tasks = (int32_t)task;
    D.n_slots = (int32_t)slot;
    // workgroups per task: ~2 items per thread in the first engine round, 1..32 each, kEngMaxBlocks in total
    std::vector<uint32_t> nb(task);
    uint32_t total_blocks = 0;
    for (uint32_t k = 0; k < task; ++k) {
        nb[k] = eng_active_chunks((uint32_t)work[k], e->task_blocks, 0);
        total_blocks += nb[k];
    }
    while (total_blocks > e->max_blocks && total_blocks > task) {
        total_blocks = 0;
        for (uint32_t k = 0; k < task; ++k) { nb[k] = std::max<uint32_t>(1, nb[k] / 2); total_blocks += nb[k]; }
    }
    uint32_t fb = 0;
    for (uint32_t k = 0; k < task; ++k) { D.task[k].first_block = fb; D.task[k].n_blocks = nb[k]; fb += nb[k]; }
    // control block: clear the mailboxes; the pending challenge (if any) is posted before the launch
    std::memset(e->h_ctl, 0, sizeof(EngCtl));
    e->binds_posted = 0;
    if (has_bind) engine_post(e, *binds[0]);
    __atomic_thread_fence(__ATOMIC_RELEASE);
    void* d_ctl = nullptr;
    void *d_round = nullptr, *d_flag = nullptr;
    JOLT_HIP_TRY(ctx, hipHostGetDevicePointer(&d_ctl, e->h_ctl, 0));
    JOLT_HIP_TRY(ctx, hipHostGetDevicePointer(&d_round, ctx->h_round, 0));
    JOLT_HIP_TRY(ctx, hipHostGetDevicePointer(&d_flag, ctx->h_flag, 0));
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(e->d_desc, e->h_desc, sizeof(EngDesc), hipMemcpyHostToDevice, ctx->stream));
    JOLT_HIP_TRY(ctx, hipMemsetAsync(e->d_sync, 0, sizeof(EngSync), ctx->stream));
    if (e->trace) {
        const uint32_t one = 1;
        JOLT_HIP_TRY(ctx, hipMemcpyAsync(&e->d_sync->trace, &one, sizeof(one), hipMemcpyHostToDevice, ctx->stream));
    }
    e->seq0 = ctx->seq + 1;
    hipLaunchKernelGGL(k_round_engine, dim3(total_blocks), dim3(kBlock), 0, ctx->stream, (const EngDesc*)e->d_desc, (const EngCtl*)d_ctl, e->d_sync, e->d_partials,
                       (Fr*)d_round, (uint64_t*)d_flag, e->seq0);
    JOLT_HIP_TRY(ctx, hipGetLastError());
    e->members.assign(members, members + n);
    e->n_rounds = rounds;
    e->round = 0;
    e->total = slot;
    e->active = true;
    return JOLT_OK;
}

// One engine round.  Returns JOLT_OK with the sums, or `*gone = true` when the engine is no longer running (it gave up
// waiting, or was never started): the caller then runs the round through the per-round kernels -- nothing of this round
// has been applied to the host-side state yet.
static int32_t engine_round(jolt_ctx* ctx, jolt_engine* e, const Fr* const* binds, jolt_fr_t* out, bool* gone, const std::function<void()>* overlap) {
    *gone = false;
    const bool has_bind = binds && binds[0];
    auto now_ns = []() { return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const long long t_post = e->trace ? now_ns() : 0;
    if (e->trace && e->round > 0) e->t_host.push_back(t_post - e->t_last_seen);
    if (e->round > 0) engine_post(e, *binds[0]);  // the challenge of the previous round
    if (overlap && *overlap) (*overlap)();        // host work that does not need the sums
    const uint64_t want = e->seq0 + (uint64_t)e->round;
    volatile uint64_t* flag = ctx->h_flag;
    uint64_t spins = 0;
    while (*flag != want) {
        if (++spins > (1ull << 20)) {
            spins = 0;
            hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) {
                if (*flag == want) break;
                e->active = false;  // the engine left (spin limit): not an error, the per-round kernels take over
                *gone = true;
                return JOLT_OK;
            }
            if (q != hipErrorNotReady) { e->active = false; ctx->last_error = std::string("round engine: ") + hipGetErrorString(q); return JOLT_ERR_HIP; }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (e->trace) { e->t_last_seen = now_ns(); e->t_device.push_back(e->t_last_seen - t_post); }
    std::memcpy(out, ctx->h_round, e->total * sizeof(Fr));
    ctx->d_round_count = 0;  // the engine publishes to the host only
    ctx->seq = want;
    // host-side bookkeeping of the bind the engine applied at the start of this round
    if (has_bind) {
        for (jolt_member* m : e->members) {
            JOLT_TRY(member_note_bind(m, *binds[0]));
            for (jolt_table* t : m->tables) {
                t->cur = t->cur < 0 ? 0 : 1 - t->cur;
                t->len /= 2;
            }
        }
    }
    e->round += 1;
    if (e->round == e->n_rounds) {
        e->active = false;  // the kernel returns by itself after its last round
        if (e->trace) {     // per-round phase timestamps of block 0 / the publishing block, in shader cycles
            static EngSync snap;
            JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            JOLT_HIP_TRY(ctx, hipMemcpy(&snap, e->d_sync, sizeof(EngSync), hipMemcpyDeviceToHost));
            std::fprintf(stderr, "[engine] host view (us): posted->seen");
            for (long long v : e->t_device) std::fprintf(stderr, " %.1f", v / 1e3);
            std::fprintf(stderr, " | seen->next post");
            for (long long v : e->t_host) std::fprintf(stderr, " %.1f", v / 1e3);
            std::fprintf(stderr, "\n");
            e->t_device.clear();
            e->t_host.clear();
            for (int r = 0; r < e->n_rounds; ++r) {
                const uint64_t* st = snap.stamps[r];
                const uint64_t prev = r ? snap.stamps[r - 1][5] : st[0];
                std::fprintf(stderr, "[engine] round %2d  wait %6lld  setup %6lld  task %6lld  ticket %6lld  (other block) %6lld  reduce+publish %6lld  (cycles)\n", r,
                             (long long)(st[0] - prev), (long long)(st[1] - st[0]), (long long)(st[2] - st[1]), (long long)(st[3] - st[2]),
                             (long long)(st[4] - st[3]), (long long)(st[5] - st[4]));
            }
        }
    }
    return JOLT_OK;
}

extern "C" int32_t jolt_round_group_prove(jolt_ctx* ctx, jolt_member* const* members, size_t n, const jolt_fr_t* const* binds, jolt_fr_t* evals_out,
                                          size_t cap) {
    return jolt_internal_round_group_prove(ctx, members, n, binds, evals_out, cap, nullptr);
}

int32_t jolt_internal_round_group_prove(jolt_ctx* ctx, jolt_member* const* members, size_t n, const jolt_fr_t* const* binds, jolt_fr_t* evals_out,
                                        size_t cap, const std::function<void()>* overlap) {
    if (!ctx || (!members && n) || !evals_out) return JOLT_ERR_INVALID_ARG;
    size_t total = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!members[i] || members[i]->ctx != ctx) return JOLT_ERR_INVALID_ARG;
        total += jolt_internal_member_n_evals(members[i]);
    }
    if (total > cap) return JOLT_ERR_SIZE_MISMATCH;
    std::vector<Fr> bstore(n);
    std::vector<const Fr*> bptr(n, nullptr);
    for (size_t i = 0; i < n; ++i) {
        if (binds && binds[i]) {
            bstore[i] = fr_from_abi(binds[i]);
            JOLT_REQUIRE(ctx, fr_is_canonical(bstore[i]), "bind challenge is not a canonical Fr");
            bptr[i] = &bstore[i];
        }
    }
    // late rounds: the persistent round engine (no launches per round)
    jolt_engine* eng = engine_get(ctx);
    if (eng && eng->active) {
        bool same = eng->members.size() == n && bptr[0] != nullptr;
        for (size_t i = 0; same && i < n; ++i) same = eng->members[i] == members[i] && bptr[i] && *bptr[i] == *bptr[0];
        if (!same) JOLT_TRY(jolt_internal_engine_quiesce(ctx));
    } else if (eng) {
        int r = engine_eligible(ctx, eng, members, n, bptr.data());
        if (r > 0) JOLT_TRY(engine_start(ctx, eng, members, n, bptr.data(), r));
    }
    if (eng && eng->active) {
        bool gone = false;
        JOLT_TRY(engine_round(ctx, eng, bptr.data(), evals_out, &gone, overlap));
        if (!gone) return JOLT_OK;
        overlap = nullptr;  // already ran
    }
    if (!ctx->round_trace) {
        JOLT_TRY(group_enqueue(ctx, members, n, bptr.data()));
        if (overlap && *overlap) (*overlap)();
        return round_wait(ctx, total, evals_out);  // no copy, no stream sync: the last workgroup published the sums
    }
    // JOLT_ROUND_TRACE=1: where a round's host time goes (between calls / enqueue / overlapped host work / waiting for the flag)
    using clk = std::chrono::steady_clock;
    static clk::time_point last_return;
    static double acc[4] = {0, 0, 0, 0};
    static size_t rounds = 0;
    auto t0 = clk::now();
    if (rounds) acc[0] += std::chrono::duration<double, std::micro>(t0 - last_return).count();
    JOLT_TRY(group_enqueue(ctx, members, n, bptr.data()));
    auto t1 = clk::now();
    if (overlap && *overlap) (*overlap)();
    auto t2 = clk::now();
    int32_t st = round_wait(ctx, total, evals_out);
    auto t3 = clk::now();
    acc[1] += std::chrono::duration<double, std::micro>(t1 - t0).count();
    acc[2] += std::chrono::duration<double, std::micro>(t2 - t1).count();
    acc[3] += std::chrono::duration<double, std::micro>(t3 - t2).count();
    last_return = t3;
    if (++rounds % 500 == 0) {
        std::fprintf(stderr, "[jolt round trace] %zu rounds: between calls %.1f us, enqueue %.1f us, overlapped host work %.1f us, wait %.1f us (averages)\n", rounds,
                     acc[0] / rounds, acc[1] / rounds, acc[2] / rounds, acc[3] / rounds);
    }
    return st;
}

// finish_rounds for a whole batch: every table of every member in ceil(tables/40) launches
extern "C" int32_t jolt_round_group_finish(jolt_ctx* ctx, jolt_member* const* members, size_t n, const jolt_fr_t* const* binds) {
    if (!ctx || (!members && n) || !binds) return JOLT_ERR_INVALID_ARG;
    JOLT_TRY(jolt_internal_join_side_writers(ctx));
    for (size_t i = 0; i < n; ++i) {
        if (!members[i] || !binds[i]) return JOLT_ERR_INVALID_ARG;
        Fr b = fr_from_abi(binds[i]);
        JOLT_REQUIRE(ctx, fr_is_canonical(b), "bind challenge is not a canonical Fr");
        // group consecutive members sharing challenge and order
        size_t j = i;
        std::vector<jolt_table*> tabs;
        while (j < n && members[j] && binds[j] && fr_from_abi(binds[j]) == b && members[j]->order == members[i]->order) {
            JOLT_TRY(member_note_bind(members[j], b));
            if (members[j]->lazy_width) {
                JOLT_TRY(jolt_internal_engine_quiesce(ctx));
                JOLT_TRY(lazy_bind_enqueue(members[j], b));
            } else {
                tabs.insert(tabs.end(), members[j]->tables.begin(), members[j]->tables.end());
            }
            ++j;
        }
        JOLT_TRY(jolt_internal_bind(ctx, tabs.data(), tabs.size(), b, members[i]->order));
        i = j - 1;
    }
    return JOLT_OK;
}

extern "C" int32_t jolt_member_finish(jolt_member* m, const jolt_fr_t* bind) {
    if (!m || !bind) return JOLT_ERR_INVALID_ARG;
    Fr b = fr_from_abi(bind);
    JOLT_REQUIRE(m->ctx, fr_is_canonical(b), "bind challenge is not a canonical Fr");
    return member_bind(m, b);
}

extern "C" int32_t jolt_member_final_values(jolt_member* m, jolt_fr_t* out, size_t k) {
    if (!m || !out) return JOLT_ERR_INVALID_ARG;
    if (m->bound != m->rounds) return JOLT_ERR_NOT_FULLY_BOUND;
    size_t need = m->tables.size() + (m->has_split_eq() ? 1 : 0);
    if (k != need) return JOLT_ERR_SIZE_MISMATCH;
    jolt_ctx* ctx = m->ctx;
    JOLT_TRY(jolt_internal_ensure_scratch(ctx, 1, need));
    for (size_t i = 0; i < m->tables.size(); ++i)
        JOLT_HIP_TRY(ctx, hipMemcpyAsync(ctx->d_results + i, m->tables[i]->data(), sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream));
    JOLT_TRY(fetch_results(ctx, m->tables.size(), out));
    for (size_t i = 0; i < m->final_unscale.size(); ++i)
        if (!(m->final_unscale[i] == Fr::one())) fr_to_abi(&out[i], mul(fr_from_abi(&out[i]), m->final_unscale[i]));
    if (m->has_split_eq()) fr_to_abi(&out[m->tables.size()], m->current_scalar);
    return JOLT_OK;
}

extern "C" int32_t jolt_round_group_final_values(jolt_ctx* ctx, jolt_member* const* members, size_t n, jolt_fr_t* out, size_t cap) {
    if (!ctx || (!members && n) || !out) return JOLT_ERR_INVALID_ARG;
    size_t total = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!members[i]) return JOLT_ERR_INVALID_ARG;
        if (members[i]->bound != members[i]->rounds) return JOLT_ERR_NOT_FULLY_BOUND;
        total += members[i]->tables.size();
    }
    if (total > cap) return JOLT_ERR_SIZE_MISMATCH;
    JOLT_TRY(jolt_internal_ensure_scratch(ctx, 1, total + 8));
    size_t k = 0;
    for (size_t i = 0; i < n; ++i)
        for (jolt_table* t : members[i]->tables)
            JOLT_HIP_TRY(ctx, hipMemcpyAsync(ctx->d_results + k++, t->data(), sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream));
    JOLT_TRY(fetch_results(ctx, total, out));
    k = 0;
    for (size_t i = 0; i < n; ++i) {
        const jolt_member* m = members[i];
        for (size_t j = 0; j < m->final_unscale.size(); ++j)
            if (!(m->final_unscale[j] == Fr::one())) fr_to_abi(&out[k + j], mul(fr_from_abi(&out[k + j]), m->final_unscale[j]));
        k += m->tables.size();
    }
    return JOLT_OK;
}

// Field views of a member's tables for the setup-time claim helper: tables that are still integer columns are promoted into temporaries (the caller frees them)
static int32_t member_tables_as_fr(jolt_ctx* ctx, const jolt_member* m, TablePtrs& tp, std::vector<void*>& temps) {
    for (size_t i = 0; i < kMaxBatchTables; ++i) tp.p[i] = nullptr;
    for (size_t i = 0; i < m->tables.size(); ++i) {
        const jolt_table* t = m->tables[i];
        if (!t->ints) { tp.p[i] = t->data(); continue; }
        Fr* buf = nullptr;
        JOLT_TRY(jolt_internal_dev_alloc(ctx, std::max<size_t>(t->len, 1) * sizeof(Fr), (void**)&buf));
        temps.push_back(buf);
        hipLaunchKernelGGL(k_from_u64, dim3(sweep_grid(ctx, t->len)), dim3(kBlock), 0, ctx->stream, reinterpret_cast<const uint64_t*>(t->ints), buf, t->len);
        JOLT_HIP_TRY(ctx, hipGetLastError());
        tp.p[i] = buf;
    }
    return JOLT_OK;
}

extern "C" int32_t jolt_member_input_claim(jolt_member* m, jolt_fr_t* out) {
    (void)jolt_internal_engine_quiesce(m ? m->ctx : nullptr);
    if (!m || !out) return JOLT_ERR_INVALID_ARG;
    jolt_ctx* ctx = m->ctx;
    int grid = sweep_grid(ctx, m->len);
    JOLT_TRY(jolt_internal_ensure_scratch(ctx, (size_t)grid, 8));
    if (m->kind == jolt_member::kExpr && !m->eq_weighted) {
        TablePtrs tp;
        std::vector<void*> temps;
        int32_t st = member_tables_as_fr(ctx, m, tp, temps);
        if (st == JOLT_OK) {
            hipLaunchKernelGGL(k_member_claim, dim3(grid), dim3(kBlock), 0, ctx->stream, (const MemberDesc*)m->d_desc, tp, m->len, ctx->d_partials);
            if (hipGetLastError() != hipSuccess) st = JOLT_ERR_HIP;
        }
        if (st == JOLT_OK) st = reduce_into_results(ctx, grid, 1, 0);
        if (st == JOLT_OK) st = fetch_results(ctx, 1, out);
        for (void* p : temps) jolt_internal_dev_free(ctx, p);
        return st;
    }
    if (m->eq_weighted) {  // sum_x scale * eq(w[..remaining], x) * q(x): the inner descriptor with the dense eq table as one more factor
        size_t rem = m->rounds - m->bound;
        jolt_table* eq = nullptr;
        JOLT_TRY(eq_build(ctx, m->w.data(), rem, m->current_scalar, 8, nullptr, &eq));
        MemberDesc md = m->desc;  // groups g: factors [f0, f1) ++ one new factor {eq}
        const uint32_t G = md.n_groups, eq_tab = (uint32_t)m->tables.size();
        int32_t st = JOLT_OK;
        if (md.n_factors + G > (uint32_t)kMaxFactors || md.n_lc + G > (uint32_t)kMaxLc || eq_tab + 1 > (uint32_t)kMaxBatchTables) st = JOLT_ERR_UNSUPPORTED;
        MemberDesc nd;
        std::memset(&nd, 0, sizeof(nd));
        if (st == JOLT_OK) {
            nd.n_groups = G;
            uint32_t nf = 0, nl = 0;
            for (uint32_t g = 0; g < G; ++g) {
                nd.grp_fac_off[g] = nf;
                for (uint32_t f = md.grp_fac_off[g]; f < md.grp_fac_off[g + 1]; ++f) {
                    nd.fac_lc_off[nf] = nl;
                    nd.fac_has_const[nf] = md.fac_has_const[f];
                    nd.fac_const[nf] = md.fac_const[f];
                    for (uint32_t k = md.fac_lc_off[f]; k < md.fac_lc_off[f + 1]; ++k) {
                        nd.lc_tab[nl] = md.lc_tab[k]; nd.lc_one[nl] = md.lc_one[k]; nd.lc_coeff[nl] = md.lc_coeff[k]; nd.lc_owner[nl] = 0;
                        nl++;
                    }
                    nf++;
                }
                nd.fac_lc_off[nf] = nl;  // the eq factor
                nd.lc_tab[nl] = eq_tab; nd.lc_one[nl] = 1; nd.lc_coeff[nl] = Fr::one();
                nl++;
                nf++;
            }
            nd.grp_fac_off[G] = nf;
            nd.fac_lc_off[nf] = nl;
            nd.n_factors = nf;
            nd.n_lc = nl;
        }
        MemberDesc* dd = nullptr;
        if (st == JOLT_OK) st = jolt_internal_dev_alloc(ctx, sizeof(MemberDesc), (void**)&dd);
        if (st == JOLT_OK && hipMemcpyAsync(dd, &nd, sizeof(nd), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) st = JOLT_ERR_HIP;
        if (st == JOLT_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = JOLT_ERR_HIP;
        std::vector<void*> temps;
        if (st == JOLT_OK) {
            TablePtrs tp;
            st = member_tables_as_fr(ctx, m, tp, temps);
            tp.p[eq_tab] = eq->data();
            if (st == JOLT_OK) {
                hipLaunchKernelGGL(k_member_claim, dim3(grid), dim3(kBlock), 0, ctx->stream, (const MemberDesc*)dd, tp, m->len, ctx->d_partials);
                if (hipGetLastError() != hipSuccess) st = JOLT_ERR_HIP;
            }
        }
        if (st == JOLT_OK) st = reduce_into_results(ctx, grid, 1, 0);
        if (st == JOLT_OK) st = fetch_results(ctx, 1, out);
        (void)hipStreamSynchronize(ctx->stream);
        for (void* p : temps) jolt_internal_dev_free(ctx, p);
        if (dd) jolt_internal_dev_free(ctx, dd);
        jolt_table_free(ctx, eq);
        return st;
    }
    if (m->kind == jolt_member::kSplitEqBooleanity) { ctx->last_error = "input_claim helper: not provided for the booleanity member (its claim is the zero check's)"; return JOLT_ERR_UNSUPPORTED; }
    // split-eq members: sum_x scale * eq(w[..remaining], x) * (product terms) with the dense eq table (claim helper only)
    size_t rem = m->rounds - m->bound;
    jolt_table* eq = nullptr;
    JOLT_TRY(eq_build(ctx, m->w.data(), rem, m->current_scalar, 8, nullptr, &eq));
    MemberDesc md;
    std::memset(&md, 0, sizeof(md));
    const uint32_t V = m->kind == jolt_member::kSplitEqUniform ? m->uni_V : 1, F = m->kind == jolt_member::kSplitEqUniform ? m->uni_F : 2;
    if ((size_t)V * (F + 1) > (size_t)kMaxLc || m->tables.size() + 1 > (size_t)kMaxBatchTables) { jolt_table_free(ctx, eq); return JOLT_ERR_UNSUPPORTED; }
    md.n_groups = V; md.n_factors = V * (F + 1); md.n_lc = V * (F + 1);
    for (uint32_t v = 0; v <= V; ++v) md.grp_fac_off[v] = v * (F + 1);
    for (uint32_t f = 0; f <= md.n_factors; ++f) md.fac_lc_off[f] = f;
    for (uint32_t v = 0; v < V; ++v) {
        uint32_t base = v * (F + 1);
        Fr c = m->kind == jolt_member::kSplitEqUniform && !m->uni_prescaled ? m->uni_coeff[v] : Fr::one();
        md.lc_tab[base] = 0; md.lc_coeff[base] = c; md.lc_one[base] = c == Fr::one() ? 1u : 0u;  // table 0 = dense eq
        for (uint32_t k = 0; k < F; ++k) { md.lc_tab[base + 1 + k] = 1 + v * F + k; md.lc_one[base + 1 + k] = 1; md.lc_coeff[base + 1 + k] = Fr::one(); }
    }
    MemberDesc* dd = nullptr;
    JOLT_TRY(jolt_internal_dev_alloc(ctx, sizeof(MemberDesc), (void**)&dd));
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(dd, &md, sizeof(md), hipMemcpyHostToDevice, ctx->stream));
    JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    TablePtrs tp;
    for (size_t i = 0; i < kMaxBatchTables; ++i) tp.p[i] = nullptr;
    tp.p[0] = eq->data();
    std::vector<jolt_table*> temp;  // index-encoded selector columns: gathered dense for this helper only (setup-time, untimed)
    if (m->lazy_width) {
        const jolt_onehot* src = m->onehot;
        const size_t per_poly = (size_t)m->lazy_width * src->k;
        for (size_t k = 0; k < m->tables.size(); ++k) {
            jolt_table* t = nullptr;
            int32_t st = jolt_internal_table_new(ctx, m->len, &t);
            if (st != JOLT_OK) { for (jolt_table* x : temp) jolt_table_free(ctx, x); jolt_internal_dev_free(ctx, dd); jolt_table_free(ctx, eq); return st; }
            temp.push_back(t);
            OneHotDense o;
            for (int i = 0; i < kMaxBatchTables; ++i) o.out[i] = i == 0 ? t->data() : nullptr;
            hipLaunchKernelGGL(k_onehot_materialize, dim3((unsigned)((m->len + kBlock - 1) / kBlock), 1), dim3(kBlock), 0, ctx->stream,
                               (const Fr*)m->d_branch[m->branch_cur], per_poly, (const uint8_t*)src->idx, src->cycles, m->lazy_width, src->k, k, o, src->wide);
            tp.p[1 + k] = t->data();
        }
    } else {
        for (size_t k = 0; k < m->tables.size(); ++k) tp.p[1 + k] = m->tables[k]->data();
    }
    hipLaunchKernelGGL(k_member_claim, dim3(grid), dim3(kBlock), 0, ctx->stream, (const MemberDesc*)dd, tp, m->len, ctx->d_partials);
    int32_t s = hipGetLastError() == hipSuccess ? JOLT_OK : JOLT_ERR_HIP;
    if (s == JOLT_OK) s = reduce_into_results(ctx, grid, 1, 0);
    if (s == JOLT_OK) s = fetch_results(ctx, 1, out);
    (void)hipStreamSynchronize(ctx->stream);
    for (jolt_table* x : temp) jolt_table_free(ctx, x);
    jolt_internal_dev_free(ctx, dd);
    jolt_table_free(ctx, eq);
    return s;
}

extern "C" int32_t jolt_member_destroy(jolt_member* m) {
    (void)jolt_internal_engine_quiesce(m ? m->ctx : nullptr);
    if (!m) return JOLT_OK;
    jolt_ctx* ctx = m->ctx;
    // no synchronisation: everything the member owns goes back to the context's pool and is reused in stream order
    for (jolt_table* t : m->tables) jolt_table_free(ctx, t);
    for (jolt_table* t : m->e_out_cache) jolt_table_free(ctx, t);
    for (jolt_table* t : m->e_in_cache) jolt_table_free(ctx, t);
    if (m->d_desc) jolt_internal_dev_free(m->ctx, m->d_desc);
    if (m->d_small) jolt_internal_dev_free(m->ctx, m->d_small);
    delete m->h_small;
    for (void* p : m->promoted) jolt_internal_dev_free(m->ctx, p);
    for (int k = 0; k < 2; ++k) if (m->d_branch[k]) jolt_internal_dev_free(m->ctx, m->d_branch[k]);
    if (m->d_base) jolt_internal_dev_free(m->ctx, m->d_base);
    if (m->d_pair) jolt_internal_dev_free(m->ctx, m->d_pair);
    delete m;
    return JOLT_OK;
}
