// jolt_amd/csrc/comm.hip -- RCCL communicator for the hypercube-sharded prover (one process per GPU, xGMI underneath).
//
// The sharded round loop (batch.hip) exchanges a few hundred bytes per sumcheck round -- every rank's partial round sums,
// all-gathered and added mod r on each rank (RCCL has no prime-field reduction) -- and, once per batch, the tables that
// remain when the shards switch to the redundant tail.  Going through torch.distributed from Python costs more per
// exchange than the round kernels themselves at the late rounds, so the data path calls RCCL directly: rendezvous stays
// with torch.distributed (it broadcasts the ncclUniqueId), the collectives are enqueued on the context's own stream.
//
// RCCL is resolved at run time with dlopen, so libjolt_hip.so has no link-time dependency on it and still loads in the CPU-only
// container.  The caller names the library: it must be the RCCL built against the HIP runtime THIS library uses (the system one);
// a process that imported torch also holds torch's private HIP/HSA runtime and librccl, and streams do not cross runtimes.
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>

#include "desc.hpp"
#include "member.hpp"
#include "poly_kernels.hip.h"

using namespace jolt;

namespace {
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    std::string error;
};

// how many loaded objects of this process carry `needle` in their file name
struct LoadedCount {
    const char* needle;
    int n;
};
static int count_loaded_cb(struct dl_phdr_info* info, size_t, void* data) {
    LoadedCount* c = (LoadedCount*)data;
    if (info->dlpi_name && std::strstr(info->dlpi_name, c->needle)) c->n++;
    return 0;
}
static int count_loaded(const char* needle) {
    LoadedCount c{needle, 0};
    dl_iterate_phdr(count_loaded_cb, &c);
    return c.n;
}

// Which RCCL.  A process holds ONE HIP runtime or this communicator refuses to exist: RCCL enqueues on the context's stream, and streams do not cross runtimes.  A
// process that imported torch BEFORE this library was loaded runs both on torch's bundled libamdhip64 (same SONAME: the loader resolves our NEEDED entry to it) and
// already holds torch's librccl with its own librocm_smi64 -- that copy is then the one to use: loading the system librccl next to it brings a second librocm_smi64
// whose C++ globals interpose with the first's, and the process aborts at exit with "double free or corruption" in ~std::map<amd::smi::DevInfoTypes, ..>
// (profiles/r05_teardown_abort_backtrace.txt: the round-4 teardown abort).  So: an RCCL already in the process wins; otherwise the named / system one is loaded.
RcclApi g_rccl;
RcclApi* rccl_api(const char* path) {
    RcclApi& api = g_rccl;
    if (api.handle) return &api;
    if (count_loaded("libamdhip64") > 1) {
        api.error = "two HIP runtimes are loaded in this process (torch was imported AFTER libjolt_hip.so was loaded): import torch first, or keep torch out of the process";
        return nullptr;
    }
    for (const char* c : {"librccl.so.1", "librccl.so"}) {
        api.handle = dlopen(c, RTLD_NOW | RTLD_NOLOAD);  // matches the SONAME of an object that is already loaded, whatever path it came from
        if (api.handle) break;
    }
    if (!api.handle && count_loaded("librocm_smi64") > count_loaded("librocm_smi64.so.1")) {
        // no RCCL in the process yet, but a bundled rocm_smi under another name: the system librccl would bring librocm_smi64.so.1 beside it -- the crash above
        api.error = "a bundled librocm_smi64 is loaded without its librccl: loading the system RCCL beside it would duplicate rocm_smi's globals";
        return nullptr;
    }
    const char* candidates[] = {path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* c : candidates) {
        if (api.handle) break;
        if (!c || !*c) continue;
        api.handle = dlopen(c, RTLD_NOW | RTLD_LOCAL);
        if (!api.handle) api.error = dlerror();
    }
    if (!api.handle) return nullptr;
    api.get_unique_id = (decltype(api.get_unique_id))dlsym(api.handle, "ncclGetUniqueId");
    api.comm_init_rank = (decltype(api.comm_init_rank))dlsym(api.handle, "ncclCommInitRank");
    api.comm_destroy = (decltype(api.comm_destroy))dlsym(api.handle, "ncclCommDestroy");
    api.all_gather = (decltype(api.all_gather))dlsym(api.handle, "ncclAllGather");
    api.error_string = (decltype(api.error_string))dlsym(api.handle, "ncclGetErrorString");
    if (!api.get_unique_id || !api.comm_init_rank || !api.comm_destroy || !api.all_gather || !api.error_string) {
        api.error = "librccl lacks an expected symbol";
        dlclose(api.handle);
        api.handle = nullptr;
        return nullptr;
    }
    return &api;
}
const char* rccl_error() { return g_rccl.error.empty() ? "dlopen librccl.so.1 failed" : g_rccl.error.c_str(); }
}  // namespace

struct jolt_comm {
    jolt_ctx* ctx = nullptr;
    RcclApi* api = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    // staging for the small per-round exchange: pinned host <-> device, grown on demand
    char *h_send = nullptr, *h_recv = nullptr, *d_send = nullptr, *d_recv = nullptr;
    size_t cap = 0;  // bytes per rank
    uint64_t* h_flag = nullptr;  // pinned, device-mapped: the publish kernel stores `seq` after the gathered bytes landed
    const void* round_sums_host = nullptr;  // set by jolt_comm_gather_round_sums: the host copy of what ctx->d_round holds
    uint64_t seq = 0;
};

// d_recv -> device-mapped pinned host memory, then the sequence number: the host spins on the flag instead of paying a
// stream synchronisation per sumcheck round (same completion scheme as the round kernels, ctx.hpp)
static __global__ __launch_bounds__(256) void k_publish_gathered(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t words,
                                                                 uint64_t* __restrict__ flag, uint64_t seq) {
    for (size_t i = threadIdx.x; i < words; i += 256) dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

#define JOLT_NCCL_TRY(c, expr)                                                                   \
    do {                                                                                         \
        ncclResult_t r_ = (expr);                                                                \
        if (r_ != ncclSuccess) {                                                                 \
            (c)->ctx->last_error = std::string(#expr ": ") + (c)->api->error_string(r_);         \
            return JOLT_ERR_HIP;                                                                 \
        }                                                                                        \
    } while (0)

// Rank 0 creates the id; the caller broadcasts the 128 bytes to the other ranks (torch.distributed) before jolt_comm_create.
extern "C" int32_t jolt_comm_unique_id(const char* rccl_path, uint8_t out[128]) {
    if (!out) return JOLT_ERR_INVALID_ARG;
    RcclApi* api = rccl_api(rccl_path);
    if (!api) return JOLT_ERR_UNSUPPORTED;
    ncclUniqueId id;
    static_assert(sizeof(id) == 128, "ncclUniqueId size");
    if (api->get_unique_id(&id) != ncclSuccess) return JOLT_ERR_HIP;
    std::memcpy(out, &id, sizeof(id));
    return JOLT_OK;
}

// Collective over all `world` ranks (ncclCommInitRank synchronises them).
extern "C" int32_t jolt_comm_create(jolt_ctx* ctx, const char* rccl_path, const uint8_t unique_id[128], int32_t rank, int32_t world,
                                    jolt_comm** out) {
    if (!ctx || !unique_id || !out || world < 1 || rank < 0 || rank >= world) return JOLT_ERR_INVALID_ARG;
    RcclApi* api = rccl_api(rccl_path);
    if (!api) {
        ctx->last_error = std::string("RCCL not available: ") + rccl_error();
        return JOLT_ERR_UNSUPPORTED;
    }
    jolt_comm* c = new (std::nothrow) jolt_comm();
    if (!c) return JOLT_ERR_OOM;
    c->ctx = ctx;
    c->api = api;
    c->rank = rank;
    c->world = world;
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); delete c; return JOLT_ERR_HIP; }
    ncclResult_t r = api->comm_init_rank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        ctx->last_error = std::string("ncclCommInitRank: ") + api->error_string(r);
        delete c;
        return JOLT_ERR_HIP;
    }
    *out = c;
    return JOLT_OK;
}

extern "C" int32_t jolt_comm_destroy(jolt_comm* c) {
    if (!c) return JOLT_OK;
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm) (void)c->api->comm_destroy(c->comm);
    if (c->h_send) (void)hipHostFree(c->h_send);
    if (c->h_recv) (void)hipHostFree(c->h_recv);
    if (c->h_flag) (void)hipHostFree(c->h_flag);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    delete c;
    return JOLT_OK;
}

extern "C" int32_t jolt_comm_world(const jolt_comm* c, int32_t* rank, int32_t* world) {
    if (!c) return JOLT_ERR_INVALID_ARG;
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return JOLT_OK;
}

static int32_t comm_reserve(jolt_comm* c, size_t bytes) {
    if (bytes <= c->cap) return JOLT_OK;
    jolt_ctx* ctx = c->ctx;
    JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (c->h_send) (void)hipHostFree(c->h_send);
    if (c->h_recv) (void)hipHostFree(c->h_recv);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    c->h_send = c->h_recv = c->d_send = c->d_recv = nullptr;
    c->cap = 0;
    size_t cap = (std::max<size_t>(4096, bytes * 2) + 3) & ~(size_t)3;
    JOLT_HIP_TRY(ctx, hipHostMalloc((void**)&c->h_send, cap, hipHostMallocDefault));
    JOLT_HIP_TRY(ctx, hipHostMalloc((void**)&c->h_recv, cap * c->world, hipHostMallocMapped | hipHostMallocCoherent));
    if (!c->h_flag) {
        JOLT_HIP_TRY(ctx, hipHostMalloc((void**)&c->h_flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
        *c->h_flag = 0;
    }
    JOLT_HIP_TRY(ctx, hipMalloc((void**)&c->d_send, cap));
    JOLT_HIP_TRY(ctx, hipMalloc((void**)&c->d_recv, cap * c->world));
    c->cap = cap;
    return JOLT_OK;
}

// All-gather of a small host payload (the per-round partial sums): pinned staging -> device -> ncclAllGather -> pinned ->
// host, all on the context stream, one synchronisation.  gathered = world blocks of `bytes`, in rank order.
extern "C" int32_t jolt_comm_all_gather_host(jolt_comm* c, const void* local, size_t bytes, void* gathered) {
    if (!c || (!local && bytes) || (!gathered && bytes)) return JOLT_ERR_INVALID_ARG;
    if (!bytes) return JOLT_OK;
    jolt_ctx* ctx = c->ctx;
    JOLT_TRY(comm_reserve(c, bytes));
    // the round sums of the last batch round already sit in device memory (written next to the host copy by the publishing
    // workgroups): send from there and skip the staging copy
    const void* send = c->d_send;
    static const bool force_stage = std::getenv("JOLT_COMM_STAGE") != nullptr;
    if (!force_stage && local == c->round_sums_host && ctx->d_round && ctx->d_round_count * sizeof(Fr) == bytes) {
        send = ctx->d_round;
    } else {
        std::memcpy(c->h_send, local, bytes);
        JOLT_HIP_TRY(ctx, hipMemcpyAsync(c->d_send, c->h_send, bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    JOLT_NCCL_TRY(c, c->api->all_gather(send, c->d_recv, bytes, ncclUint8, c->comm, ctx->stream));
    const size_t total = bytes * c->world;
    if (total % 4 == 0 && total <= (1u << 16)) {  // the per-round payload: publish + spin (no stream synchronisation)
        void *d_host = nullptr, *d_flag = nullptr;
        JOLT_HIP_TRY(ctx, hipHostGetDevicePointer(&d_host, c->h_recv, 0));
        JOLT_HIP_TRY(ctx, hipHostGetDevicePointer(&d_flag, c->h_flag, 0));
        const uint64_t want = ++c->seq;
        hipLaunchKernelGGL(k_publish_gathered, dim3(1), dim3(256), 0, ctx->stream, (const uint32_t*)c->d_recv, (uint32_t*)d_host, total / 4, (uint64_t*)d_flag,
                           want);
        JOLT_HIP_TRY(ctx, hipGetLastError());
        volatile uint64_t* flag = c->h_flag;
        uint64_t spins = 0;
        while (*flag != want) {
            if (++spins > (1ull << 22)) {  // tens of ms without the flag: consult the runtime (a peer may simply be late)
                spins = 0;
                hipError_t q = hipStreamQuery(ctx->stream);
                if (q == hipSuccess) {
                    if (*flag == want) break;
                    ctx->last_error = "all-gather finished without publishing its completion flag";
                    return JOLT_ERR_HIP;
                }
                if (q != hipErrorNotReady) { ctx->last_error = std::string("all-gather: ") + hipGetErrorString(q); return JOLT_ERR_HIP; }
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    } else {
        JOLT_HIP_TRY(ctx, hipMemcpyAsync(c->h_recv, c->d_recv, total, hipMemcpyDeviceToHost, ctx->stream));
        JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    std::memcpy(gathered, c->h_recv, total);
    return JOLT_OK;
}

// All-gather between device buffers (the tables handed to the redundant tail), asynchronous on the context stream.
extern "C" int32_t jolt_comm_all_gather_device(jolt_comm* c, const void* d_local, size_t bytes, void* d_gathered) {
    if (!c || (!d_local && bytes) || (!d_gathered && bytes)) return JOLT_ERR_INVALID_ARG;
    if (!bytes) return JOLT_OK;
    JOLT_NCCL_TRY(c, c->api->all_gather(d_local, d_gathered, bytes, ncclUint8, c->comm, c->ctx->stream));
    return JOLT_OK;
}

// jolt_gather_fn for jolt_host_batch_run: user = jolt_comm*
extern "C" int32_t jolt_comm_gather_round_sums(void* user, const jolt_fr_t* local, size_t count, jolt_fr_t* gathered) {
    jolt_comm* c = static_cast<jolt_comm*>(user);
    if (!c) return JOLT_ERR_INVALID_ARG;
    c->round_sums_host = local;  // these are the sums of the round that just completed: ctx->d_round mirrors them on the device
    int32_t s = jolt_comm_all_gather_host(c, local, count * sizeof(jolt_fr_t), gathered);
    c->round_sums_host = nullptr;
    c->ctx->d_round_count = 0;  // the device mirror is consumed: the next gather must not send it again by accident
    return s;
}

// ------------------------------------------------------------------------------------------------------------------
// Hand-over to the redundant tail: when a shard's tables are down to `entries` values each, every rank packs them,
// the packs are all-gathered, and the tail tables are rebuilt with the rank as the top variables:
//   tail[t][r * entries + j] = table t of rank r, entry j      (LowToHigh binding keeps the top variables for last)
// ------------------------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(kBlock) void k_pack_tables(TablePtrs tp, int n_tables, size_t entries, Fr* __restrict__ dst) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (size_t)n_tables * entries) return;
    size_t t = i / entries, j = i % entries;
    st_fr(dst + i, ld_fr(tp.p[t] + j));
}
static __global__ __launch_bounds__(kBlock) void k_tail_interleave(const Fr* __restrict__ gathered, size_t world, size_t n_tables, size_t entries,
                                                                   Fr* __restrict__ dst) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;  // index into dst: (t, r, j)
    if (i >= world * n_tables * entries) return;
    size_t j = i % entries, r = (i / entries) % world, t = i / (entries * world);
    st_fr(dst + i, ld_fr(gathered + (r * n_tables + t) * entries + j));
}

extern "C" int32_t jolt_round_group_pack_tables(jolt_ctx* ctx, jolt_member* const* members, size_t n, size_t entries, jolt_table* dst) {
    if (!ctx || (!members && n) || !dst || !entries) return JOLT_ERR_INVALID_ARG;
    size_t total = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!members[i]) return JOLT_ERR_INVALID_ARG;
        if (members[i]->len != entries) return JOLT_ERR_SIZE_MISMATCH;
        if (members[i]->lazy_width) return JOLT_ERR_UNSUPPORTED;  // index-encoded columns have no table to pack yet
        total += members[i]->tables.size();
    }
    if (dst->cur < 0 || dst->len < total * entries) return JOLT_ERR_SIZE_MISMATCH;
    JOLT_TRY(jolt_internal_engine_quiesce(ctx));
    std::vector<const Fr*> ptrs;
    for (size_t i = 0; i < n; ++i)
        for (jolt_table* t : members[i]->tables) ptrs.push_back(t->data());
    for (size_t base = 0; base < ptrs.size(); base += kMaxBatchTables) {
        int k = (int)std::min<size_t>(kMaxBatchTables, ptrs.size() - base);
        TablePtrs tp;
        for (int i = 0; i < kMaxBatchTables; ++i) tp.p[i] = i < k ? ptrs[base + i] : nullptr;
        size_t work = (size_t)k * entries;
        hipLaunchKernelGGL(k_pack_tables, dim3((unsigned)((work + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, tp, k, entries,
                           dst->data() + base * entries);
    }
    JOLT_HIP_TRY(ctx, hipGetLastError());
    return JOLT_OK;
}

extern "C" int32_t jolt_tail_interleave(jolt_ctx* ctx, const jolt_table* gathered, size_t world, size_t n_tables, size_t entries, jolt_table* dst) {
    if (!ctx || !gathered || !dst || !world || !entries) return JOLT_ERR_INVALID_ARG;
    size_t total = world * n_tables * entries;
    if (gathered->len < total || dst->len < total || dst->cur < 0) return JOLT_ERR_SIZE_MISMATCH;
    if (!total) return JOLT_OK;
    hipLaunchKernelGGL(k_tail_interleave, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, (const Fr*)gathered->data(), world,
                       n_tables, entries, dst->data());
    JOLT_HIP_TRY(ctx, hipGetLastError());
    return JOLT_OK;
}

// all-gather of a whole table: gathered (world * local->len entries) <- every rank's `local`
extern "C" int32_t jolt_comm_all_gather_table(jolt_comm* c, const jolt_table* local, size_t n, jolt_table* gathered) {
    if (!c || !local || !gathered) return JOLT_ERR_INVALID_ARG;
    if (n > local->len || gathered->len < n * (size_t)c->world || gathered->cur < 0) return JOLT_ERR_SIZE_MISMATCH;
    return jolt_comm_all_gather_device(c, local->data(), n * sizeof(Fr), gathered->data());
}
