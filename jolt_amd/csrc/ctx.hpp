// jolt_amd/csrc/ctx.hpp -- context, table and member objects behind the opaque C-ABI handles of include/jolt_hip.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/jolt_hip.h"
#include "field.hip.h"

using jolt::Fq;
using jolt::Fr;

static_assert(sizeof(Fr) == sizeof(jolt_fr_t), "Fr must be layout-compatible with jolt_fr_t");

struct jolt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 256;
    size_t max_lds_per_block = 65536;  // hipDeviceProp::sharedMemPerBlock (gfx950: 160 KiB)
    std::string last_error;
    // reduction scratch: per-block partial sums, final results (device) and their pinned host mirror
    Fr* d_partials = nullptr;
    size_t partials_cap = 0;  // in Fr
    Fr* d_results = nullptr;
    Fr* h_results = nullptr;  // pinned
    size_t results_cap = 0;   // in Fr
    void* d_desc_scratch = nullptr;
    // batch-round completion without a copy + stream sync: the last workgroup of a round writes the round sums straight
    // into host-mapped pinned memory and then publishes a sequence number the host spins on
    Fr* h_round = nullptr;          // pinned, device-mapped (fine-grained)
    uint64_t* h_flag = nullptr;     // pinned, device-mapped
    Fr* d_round = nullptr;          // device copy of the last round's sums (what a sharded prover all-gathers)
    size_t d_round_count = 0;       // sums of the last completed round held by d_round (0: none)
    uint32_t* d_counters = nullptr; // [0..31] per-member tickets, [32] group ticket
    uint64_t seq = 0;
    size_t round_cap = 0;
    // independent kernels of one batch round run concurrently: the main stream plus three side streams, forked after the
    // round's bind launches (completion is tracked by the in-kernel tickets, not by stream order)
    hipStream_t side[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr;
    // a side stream that ran a table-WRITING kernel (fused bind) records ev_join[k]; the main stream waits for it before the next
    // operation that touches tables (the host only knows that kernel took its tickets, not that its write-back completed)
    hipEvent_t ev_join[3] = {nullptr, nullptr, nullptr};
    bool join_pending[3] = {false, false, false};
    // MSM lanes: lane 0 runs on `stream`, lanes 1..3 on side[0..2]; each has a grow-only device workspace and a pinned
    // host buffer for the window sums, so independent MSMs (the HyperKZG level commitments) overlap
    void* msm_ws[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t msm_ws_cap[4] = {0, 0, 0, 0};
    void* msm_host[4] = {nullptr, nullptr, nullptr, nullptr};
    // the batch of short MSMs of jolt_internal_msm_many (msm.hip): its own stream, workspace and pinned window sums, beside the four lanes
    hipStream_t msm_batch_stream = nullptr;
    hipStream_t hint_stream = nullptr;  // lowest priority: the opening hint's class sums (pcs.hip jolt_grid_hint_begin), created at first use
    void* msm_batch_ws = nullptr;
    size_t msm_batch_ws_cap = 0;
    void* msm_batch_host = nullptr;
    size_t msm_batch_host_cap = 0;
    bool msm_batch = true;  // JOLT_MSM_BATCH=0: every short MSM on its own (A/B)
    // a pair of fixed-base MSMs over one sort (msm_fixed.hip): the first result's reduction runs here, under the second pass's bucket sums
    hipStream_t msm_aux_stream = nullptr;
    hipEvent_t ev_aux[4][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    bool msm_tables_pending = false;  // between jolt_msm_g1_tables_begin and _finish: the side lanes hold MSMs in flight
    void* msm_pending_one = nullptr;  // an MSM begun by jolt_internal_msm_one_begin and not yet collected (msm.hip)
    bool msm_pair_overlap = true;  // JOLT_MSM_PAIR_OVERLAP=0: reduction between the two passes (A/B)
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    // jolt_msm_profile_buckets: HIP events around the fixed-base MSM's dominant kernel (k_fx_buckets_ordered) ON THE STREAM IT IS LAUNCHED ON, and where the launch's
    // count of non-zero digits (= mixed additions) lives on the device -- the `roofline_msm` object of bench.py
    hipStream_t copy_stream = nullptr;   // jolt_rows_upload_begin: H2D copies beside the main stream's kernels
    hipEvent_t ev_copy_fork = nullptr;
    bool fx_profile = false;
    hipEvent_t ev_fx[2] = {nullptr, nullptr};
    const uint32_t* fx_profile_info = nullptr;  // device: info[1] = non-zero digits of the profiled launch
    bool fx_profile_valid = false;
    // Sort token (JOLT_MSM_STAGGER): the partition / sort phase of a fixed-base MSM is HBM-bound and its bucket sums are bound by
    // integer multiply-adds, so concurrent lanes only gain when one lane's sort runs under ANOTHER lane's bucket sums.  Equal MSMs
    // enqueued together (the three witness MSMs of an opening) would run their sorts at the same time; each sort phase therefore
    // waits for the previous MSM's sort phase (whatever its lane) and records the next event of this ring when it is done.
    hipEvent_t ev_sort[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    unsigned sort_seq = 0;        // sort phases recorded so far
    int sort_last_lane = -1;      // lane of the last recorded sort phase (no wait needed on the same stream)
    // persistent round engine for the late rounds of a batch (engine_kernel.hip.h); owned by capi.hip
    struct jolt_engine* engine = nullptr;
    // uniform split-eq members switch from (product, pair) work items to one item per pair at this many pairs
    // (JOLT_UNIFORM_ROWS_PAIRS overrides; tests lower it to run the row-major kernels at small sizes)
    size_t uniform_rows_pairs = (size_t)1 << 16;
    size_t grid_floor = 3;        // adaptive grid: workgroups per CU before the items-per-thread rule adds more (JOLT_GRID_FLOOR; 1..4 equal at 2^20, 3 is 4 % ahead at 2^22)
    size_t grid_mult = 0;         // workgroups per CU of a round-sum kernel (JOLT_GRID_MULT); 0 = by size (round_grid)
    size_t fuse_ratio = 2;        // an expression member's pending bind is fused into its round kernel when multiplies per pair <= ratio x tables (JOLT_FUSE_RATIO)
    size_t tail_pairs = 16384;    // rounds with at most this many pairs use the tail kernel (JOLT_TAIL_PAIRS; 4096..65536 measure within 2 %)
    bool fuse_tail = false;       // JOLT_FUSE_TAIL=1: pending binds of expr members are applied inside the tail kernel too
    bool msm_lds_attr_set = false;
    bool msm_fx_attr_set = false;
    bool grid_hint_attr_set = false;  // k_grid_onehot_sum's dynamic-LDS limit raised on this device (pcs.hip: background class sums reserve LDS to stay at one wave per SIMD)
    bool rows_many_attr_set = false;  // k_rows_to_ints_many's dynamic-LDS limit raised on this device (onehot.hip)
    int msm_lanes = 4;            // MSM lanes used by jolt_internal_msm_many (JOLT_MSM_LANES=1: every MSM on the main stream, for standalone kernel durations)
    int msm_fx_partition = 2;     // JOLT_FX_PARTITION=1: one-pass segment scatter (A/B of the two coalesced passes in msm_fixed.hip)
    bool msm_uniform_scalars = false;     // set around MSMs whose scalars are UNIFORM field elements (quotients of a random linear combination, the witness polynomials of an opening): the digit sort
                                          // may then size its regions from the digit model (capacity sort, msm_fixed.hip 2d).  Level commitments do not set it: the folds of a sparse or
                                          // few-valued polynomial overflow the regions and would pay the capacity passes AND the exact sort
    bool msm_full_width_scalars = false;  // set by a caller around MSMs whose scalars are uniform field elements (the level commitments of an opening): lets mid-length ones use the mid table set
    bool msm_fx_soa = true;          // fixed-base MSM: no key array, split entries in the partition / segment sort (JOLT_FX_SOA=0: 8-byte entries, for an A/B)
    bool msm_fx_grid_reduce = true;  // fixed-base MSM: bucket reduction by rows and columns (JOLT_FX_REDUCE=0: running sums, for an A/B)
    // JOLT_MSM_CU_SPLIT=k (experiment, off by default): the fixed-base MSM's HBM-bound phases (digits .. bucket order) and its bucket reduction run on
    // streams confined to k compute units of every group of 8, its bucket sums on streams confined to the other 8 - k, so that one MSM's
    // sort runs UNDER another MSM's bucket sums (the bucket kernel fills the VGPR file of every CU it can reach: nothing co-resides with it)
    int msm_cu_split = 0;
    hipStream_t sort_stream[4] = {nullptr, nullptr, nullptr, nullptr}, bucket_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_phase[4][4] = {};
    bool msm_stagger = false;     // JOLT_MSM_STAGGER=1: serialise the sort phases of concurrent fixed-base MSMs (see ev_sort)
    int msm_fx_reduce_div = 24;   // buckets per thread of the fixed-base bucket reduction (JOLT_FX_REDUCE_DIV)
    bool msm_fx_lform = true;     // JOLT_FX_LFORM=0: window tables in standard form, word-form XYZZ accumulators (A/B of fq_limb.hip.h)
    bool msm_fx_stage = true;     // JOLT_FX_STAGE=0: segment sort scatters straight to global memory (A/B of the LDS-staged segment)
    bool msm_fixed = true;        // JOLT_MSM_FIXED=0: ignore window-precomputed bases (A/B of msm_fixed.hip)
    bool msm_lds_sort = true;     // MSM counting sort with per-workgroup LDS histograms (JOLT_MSM_LDS_SORT=0: global atomics per key)
    bool round_trace = false;     // JOLT_ROUND_TRACE=1: print where the host time of a batch round goes
    bool serial_streams = false;  // JOLT_SERIAL_STREAMS=1: a round's kernels on one stream (standalone kernel durations under rocprof)
    bool lazy_lds = true;  // index-encoded members past the first bind: branch tables staged in LDS (JOLT_LAZY_LDS=0: global gathers)
    // Device-memory pool for tables, member descriptors and per-call temporaries (jolt_internal_dev_alloc / _free): a proof builds
    // and drops dozens of T-sized derived tables (eq / eq+1 / LT expansions, linear-leaf fusions, bind scratch), and hipMalloc /
    // hipFree cost 0.1-1 ms each and synchronise the device.  Freed blocks are kept per size class and handed out again WITHOUT a
    // synchronisation: every consumer enqueues on the context's main stream (or on a side stream forked from it after the
    // allocation), so reuse is stream-ordered.  JOLT_POOL=0 disables caching (every free is a hipFree).
    bool pool_enabled = true;
    std::unordered_map<void*, size_t> pool_live;               // block -> size class (bytes)
    std::unordered_map<size_t, std::vector<void*>> pool_free;  // size class -> cached blocks
    size_t pool_cached_bytes = 0, pool_live_bytes = 0, pool_peak_bytes = 0;
};

// Stop a running round engine (if any) so that other work may use the stream / the members' tables.
int32_t jolt_internal_engine_quiesce(jolt_ctx* ctx);
int32_t jolt_internal_join_side_writers(jolt_ctx* ctx);

struct jolt_table {
    jolt_ctx* ctx = nullptr;
    Fr* buf[2] = {nullptr, nullptr};  // owned; LowToHigh binds ping-pong between them
    size_t cap[2] = {0, 0};
    int cur = 0;                      // -1: the evaluations are the borrowed `view` (never written)
    size_t len = 0;
    const Fr* view = nullptr;         // borrowed source of a member that does not own its tables
    size_t view_len = 0;
    // != nullptr (only inside members made by jolt_member_create_lc_small, until their first bind): the evaluations are the unpromoted u64 entries of a
    // resident witness column -- data() is null in that state; round 0 and the first bind read the integers (small_round.hip.h)
    const void* ints = nullptr;
    const void* ints_src = nullptr;   // what jolt_member_reset restores `ints` to
    Fr* data() const { return cur < 0 ? const_cast<Fr*>(view) : buf[cur]; }
};

#define JOLT_HIP_TRY(ctx, expr)                                                                       \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            if (ctx) (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(e_);           \
            return e_ == hipErrorOutOfMemory ? JOLT_ERR_OOM : JOLT_ERR_HIP;                           \
        }                                                                                             \
    } while (0)

#define JOLT_TRY(expr)                     \
    do {                                   \
        int32_t s_ = (expr);               \
        if (s_ != JOLT_OK) return s_;      \
    } while (0)

#define JOLT_REQUIRE(ctx, cond, msg)                            \
    do {                                                        \
        if (!(cond)) {                                          \
            if (ctx) (ctx)->last_error = (msg);                 \
            return JOLT_ERR_INVALID_ARG;                        \
        }                                                       \
    } while (0)

static inline Fr fr_from_abi(const jolt_fr_t* p) {
    Fr r;
    std::memcpy(&r, p, sizeof(Fr));
    return r;
}
static inline void fr_to_abi(jolt_fr_t* p, const Fr& v) { std::memcpy(p, &v, sizeof(Fr)); }
static inline bool fr_low_limbs_zero(const Fr& v) { return (v.l[0] | v.l[1] | v.l[2] | v.l[3]) == 0; }
static inline bool fr_is_canonical(const Fr& v) {
    Fr d;
    return jolt::sub_p(d, v) != 0;
}

// internal helpers implemented in capi.hip
int32_t jolt_internal_dev_alloc(jolt_ctx* ctx, size_t bytes, void** out);
void jolt_internal_dev_free(jolt_ctx* ctx, void* p);
int32_t jolt_internal_pool_trim(jolt_ctx* ctx);
int32_t jolt_internal_ensure_scratch(jolt_ctx* ctx, size_t partials, size_t results);
int32_t jolt_internal_table_new(jolt_ctx* ctx, size_t len, jolt_table** out);
int32_t jolt_internal_table_ensure_alt(jolt_table* t, size_t need);
