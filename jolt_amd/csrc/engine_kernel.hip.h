// jolt_amd/csrc/engine_kernel.hip.h -- persistent "round engine" for the late, latency-bound sumcheck rounds.
//
// Once every member of a batch is down to a few thousand pairs, a round's arithmetic takes a few microseconds while
// the host pays ~10 HIP API calls (bind launches, stream fork, one launch per member class) for it: measured 50-115 us
// per round, i.e. ~40 % of a 2^20 proof spent in the last 13 of 20 rounds.  The engine is ONE kernel that stays resident
// for all remaining rounds of the batch (ProveRounds::prove_round contract unchanged, crates/jolt-sumcheck/src/prover.rs:52-72):
//     round k:  wait for challenge k-1 in host-mapped memory  ->  round sums of every member, the pending bind fused
//               into the loads (one task per workgroup range)  ->  last workgroup reduces the partials and publishes
//               the sums + sequence number to host-mapped memory  ->  next round
// The host still assembles the round polynomial and runs the transcript (Fiat-Shamir needs the sums before the next
// challenge exists); it just never launches anything.  Same arithmetic, same order-independent modular sums as the
// per-round kernels -> bit-identical results.
//
// Safety: every spin loop gives up after kEngSpinLimit polls (or when the host posts an abort), so a host that stops
// driving the rounds cannot leave the GPU spinning; the host then falls back to the per-round kernels (table state is
// consistent at every round boundary: binds are out of place).
#pragma once
#include <type_traits>

#include "sumcheck_kernels.hip.h"

namespace jolt {

constexpr int kEngMaxMembers = 8;
constexpr int kEngMaxTables = 64;
constexpr int kEngMaxRounds = 26;
constexpr int kEngMaxTasks = 32;
constexpr int kEngMaxSlots = 32;
constexpr int kEngMaxBlocks = 256;            // <= one workgroup per CU: the whole grid is always co-resident
constexpr uint32_t kEngSpinLimit = 1u << 15;  // polls of host memory (~2-3 us each) before the engine gives up: ~0.1 s

struct EngTable {
    const Fr* src;       // evaluations before the first engine bind
    Fr* buf[2];          // ping-pong targets of the engine's binds
    uint32_t first_out;  // buf index written by the first engine bind
    uint32_t len0;       // entries before the first engine bind
};
struct EngMember {
    int32_t kind;  // jolt_member::Kind
    uint32_t n_tables, tab_off, ne, slot, skip_one;
    const MemberDesc* desc;  // kExpr
    uint32_t V, F;           // kSplitEqUniform
    uint32_t coeff_one[kMaxGroups];
    Fr coeff[kMaxGroups];
    // split-eq members: the E_out / E_in tables in force when engine round k is evaluated
    const Fr* e_out[kEngMaxRounds];
    const Fr* e_in[kEngMaxRounds];
    int32_t in_bits[kEngMaxRounds];
};
// one unit of phase-2 work: (member, evaluation point) for expression members, the whole member otherwise
struct EngTask {
    uint32_t member, point, first_block, n_blocks, slot, n_acc;
    uint32_t items0;  // work items in the first engine round (halves every round): decides how many workgroups stay
};
struct EngDesc {
    int32_t n_members, n_tables, n_rounds, first_has_bind, n_tasks, n_slots;
    uint32_t slot_task[kEngMaxSlots];  // owning task of every result slot
    EngMember m[kEngMaxMembers];
    EngTable t[kEngMaxTables];
    EngTask task[kEngMaxTasks];
};
// host-mapped pinned memory written by the host: one 64-byte mailbox per bind.  The host fills `challenge`, then stores
// `seq_a = seq_b = index + 1` (release); a workgroup polls the whole line and accepts it when both copies match, so a
// challenge costs ONE PCIe read round trip once it has been posted.  `abort` != 0 makes every workgroup leave.
struct alignas(64) EngMail {
    uint64_t seq_a;
    uint32_t challenge[8];
    uint64_t seq_b;
    uint64_t pad;
};
static_assert(sizeof(EngMail) == 64, "mailbox is one cache line");
struct EngCtl {
    EngMail mail[kEngMaxRounds + 1];
    uint64_t abort;
};
// device memory, zeroed before the launch
constexpr int kEngStamps = 6;  // per round: challenge seen, pointers set, task done, ticket taken | publishing block: start, published
struct EngSync {
    uint32_t abort, done_count, trace, pad[5];
    uint64_t stamps[kEngMaxRounds][kEngStamps];  // shader-clock timestamps (JOLT_ENGINE_TRACE=1), block 0 / the publishing block
};

__device__ __forceinline__ bool eng_aborted(EngSync* s) { return __hip_atomic_load(&s->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; }

// workgroups of a task that still have work in `round` (the others have left the kernel): ~2 items per thread
__host__ __device__ __forceinline__ uint32_t eng_active_chunks(uint32_t items0, uint32_t n_blocks, int round) {
    uint32_t items = items0 >> round;
    uint32_t c = (items + 2 * kBlock - 1) / (2 * kBlock);
    return c < 1 ? 1 : (c > n_blocks ? n_blocks : c);
}

template <int NE>
__device__ __forceinline__ void eng_block_reduce(Fr (&acc)[NE], Fr* __restrict__ partials, uint32_t slot, uint32_t chunk) {
    __shared__ Fr sm[kBlock / 64][NE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int t = 0; t < NE; ++t) {
            Fr o;
#pragma unroll
            for (int k = 0; k < 8; ++k) o.l[k] = __shfl_xor(acc[t].l[k], off, 64);
            acc[t] = add(acc[t], o);
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t < NE; ++t) sm[wave][t] = acc[t];
    }
    __syncthreads();
    if (threadIdx.x < NE) {
        Fr s = sm[0][threadIdx.x];
        for (int w = 1; w < kBlock / 64; ++w) s = add(s, sm[w][threadIdx.x]);
        st_fr(partials + (size_t)(slot + threadIdx.x) * kEngMaxBlocks + chunk, s);
    }
    __syncthreads();
}

// The (lo, hi) pair at pair index y.  fused: `in` has not been bound with the round's challenge yet -- read the four
// entries 4y..4y+3 and bind on the fly; the table's owner stores the two bound values into `out` for the next round.
__device__ __forceinline__ void eng_load_pair(const Fr* __restrict__ in, Fr* __restrict__ out, bool fused, bool owner, size_t y, const Fr& r, bool shifted,
                                              Fr& lo, Fr& hi) {
    if (fused) {
        Fr a0 = ld_fr(in + 4 * y), a1 = ld_fr(in + 4 * y + 1), a2 = ld_fr(in + 4 * y + 2), a3 = ld_fr(in + 4 * y + 3);
        lo = bind_pair_rt(a0, a1, r, shifted);
        hi = bind_pair_rt(a2, a3, r, shifted);
        if (owner) { st_fr(out + 2 * y, lo); st_fr(out + 2 * y + 1, hi); }
    } else {
        lo = ld_fr(in + 2 * y);
        hi = ld_fr(in + 2 * y + 1);
    }
}

// `partials`: kEngMaxSlots * kEngMaxBlocks field elements; results / flag: host-mapped pinned memory; the flag of engine
// round k is seq0 + k.  One workgroup range per task; the pending bind is fused into the task's loads (the next round
// starts only after the host has seen this round's sums, i.e. after every workgroup took its ticket, so the bound
// tables need no grid barrier).
static __global__ __launch_bounds__(kBlock) void k_round_engine(const EngDesc* __restrict__ desc, const EngCtl* ctl, EngSync* sync, Fr* __restrict__ partials,
                                                               Fr* results, uint64_t* flag, uint64_t seq0) {
    __shared__ const Fr* s_in[kEngMaxTables];  // this round's input evaluations (not yet bound when the round has a bind)
    __shared__ Fr* s_out[kEngMaxTables];       // where the bound evaluations go
    __shared__ Fr s_r;
    __shared__ uint32_t s_flag;
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const int n_tables = desc->n_tables, n_rounds = desc->n_rounds, n_tasks = desc->n_tasks, n_slots = desc->n_slots;
    int my_task = 0;  // static for the whole run; every block belongs to exactly one task
    for (int k = 0; k < n_tasks; ++k)
        if (b >= desc->task[k].first_block && b < desc->task[k].first_block + desc->task[k].n_blocks) my_task = k;
    const EngTask tk = desc->task[my_task];
    const EngMember& M = desc->m[tk.member];
    const uint32_t chunk = b - tk.first_block;
    uint32_t binds_done = 0;
    const bool trace = sync->trace != 0;
#define ENG_STAMP(k) do { if (trace && tid == 0 && b == 0) sync->stamps[round][k] = __builtin_readcyclecounter(); } while (0)
    for (int round = 0; round < n_rounds; ++round) {
        const bool fused = round > 0 || desc->first_has_bind;
        Fr r = Fr::zero();
        // workgroups beyond the task's remaining work leave for good (work only shrinks); the others stride by `nb`
        const uint32_t nb = eng_active_chunks(tk.items0, tk.n_blocks, round);
        if (chunk >= nb) return;
        if (fused) {
            // ---- wait for the challenge: poll this bind's mailbox in host memory (one line, validated by its two sequence copies)
            if (tid < 16) {
                const volatile uint32_t* mail = reinterpret_cast<const volatile uint32_t*>(&ctl->mail[binds_done]);
                const uint32_t want = binds_done + 1;
                uint32_t ok = 1, spins = 0, word = 0;
                for (;;) {
                    word = mail[tid];  // 16 lanes x 4 bytes = the whole line in one request
                    const uint32_t a_lo = __shfl(word, 0, 64), b_lo = __shfl(word, 10, 64);
                    if (a_lo == want && b_lo == want) break;
                    if ((++spins & 63) == 0) {
                        uint64_t ab = __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (ab != 0 || spins > kEngSpinLimit || eng_aborted(sync)) { ok = 0; break; }
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (ok && tid >= 2 && tid < 10) s_r.l[tid - 2] = word;
                if (tid == 0) {
                    if (!ok) __hip_atomic_store(&sync->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_flag = ok;
                }
            }
            __syncthreads();
            if (!s_flag) return;
            __threadfence();  // acquire: the tables bound by the other workgroups in the previous round
            r = s_r;
        }
        ENG_STAMP(0);
        if ((int)tid < n_tables) {
            const EngTable& T = desc->t[tid];
            s_in[tid] = binds_done == 0 ? T.src : T.buf[(T.first_out + binds_done - 1) & 1];
            s_out[tid] = T.buf[(T.first_out + binds_done) & 1];
        }
        __syncthreads();
        ENG_STAMP(1);
        const bool shifted = (r.l[0] | r.l[1] | r.l[2] | r.l[3]) == 0;
        const uint32_t half = (desc->t[M.tab_off].len0 >> (binds_done + (fused ? 1 : 0))) >> 1;  // pairs evaluated this round
        const Fr* const* tin = s_in + M.tab_off;
        Fr* const* tout = s_out + M.tab_off;
        if (M.kind == 0) {
            const MemberDesc* __restrict__ d = M.desc;
            const uint32_t ng = d->n_groups;
            const uint32_t point = (M.skip_one && tk.point >= 1) ? tk.point + 1 : tk.point;
            const bool writer = tk.point == 0;
            Fr acc[1] = {Fr::zero()};
            const uint32_t items = half * ng;
            for (uint32_t i = chunk * kBlock + tid; i < items; i += nb * kBlock) {
                const uint32_t g = i / half, y = i - g * half;
                Fr prod = Fr::one();
                const uint32_t f0 = d->grp_fac_off[g], f1 = d->grp_fac_off[g + 1];
                for (uint32_t f = f0; f < f1; ++f) {
                    Fr lo = d->fac_has_const[f] ? d->fac_const[f] : Fr::zero(), hi = lo;
                    for (uint32_t k = d->fac_lc_off[f]; k < d->fac_lc_off[f + 1]; ++k) {
                        const uint32_t ti = d->lc_tab[k];
                        Fr x, z;
                        eng_load_pair(tin[ti], tout[ti], fused, writer && d->lc_owner[k], y, r, shifted, x, z);
                        if (!d->lc_one[k]) {
                            Fr c = d->lc_coeff[k];
                            x = mul(x, c);
                            z = mul(z, c);
                        }
                        lo = add(lo, x);
                        hi = add(hi, z);
                    }
                    Fr step = sub(hi, lo), v = lo;
                    for (uint32_t q = 0; q < point; ++q) v = add(v, step);
                    prod = f == f0 ? v : mul(prod, v);
                }
                acc[0] = add(acc[0], prod);
            }
            eng_block_reduce<1>(acc, partials, tk.slot, chunk);
        } else if (M.kind == 1) {
            const Fr* __restrict__ e_out = M.e_out[round];
            const Fr* __restrict__ e_in = M.e_in[round];
            const int in_bits = M.in_bits[round];
            const uint32_t mask = (1u << in_bits) - 1;
            Fr acc[2] = {Fr::zero(), Fr::zero()};
            for (uint32_t row = chunk * kBlock + tid; row < half; row += nb * kBlock) {
                Fr e = mul(ld_fr(e_out + (row >> in_bits)), ld_fr(e_in + (row & mask)));
                Fr a_lo, a_hi, b_lo, b_hi;
                eng_load_pair(tin[0], tout[0], fused, true, row, r, shifted, a_lo, a_hi);
                eng_load_pair(tin[1], tout[1], fused, true, row, r, shifted, b_lo, b_hi);
                acc[0] = add(acc[0], mul(e, mul(a_lo, b_lo)));
                acc[1] = add(acc[1], mul(e, mul(sub(a_hi, a_lo), sub(b_hi, b_lo))));
            }
            eng_block_reduce<2>(acc, partials, tk.slot, chunk);
        } else {
            const Fr* __restrict__ e_out = M.e_out[round];
            const Fr* __restrict__ e_in = M.e_in[round];
            const int in_bits = M.in_bits[round];
            const uint32_t mask = (1u << in_bits) - 1;
            const uint32_t items = half * M.V;
            auto run = [&](auto fc) {
                constexpr int F = decltype(fc)::value;
                Fr acc[F];
#pragma unroll
                for (int t = 0; t < F; ++t) acc[t] = Fr::zero();
                for (uint32_t i = chunk * kBlock + tid; i < items; i += nb * kBlock) {
                    const uint32_t v = i / half, row = i - v * half;
                    Fr lo[F], hi[F];
#pragma unroll
                    for (int k = 0; k < F; ++k) eng_load_pair(tin[v * F + k], tout[v * F + k], fused, true, row, r, shifted, lo[k], hi[k]);
                    Fr w = mul(ld_fr(e_out + (row >> in_bits)), ld_fr(e_in + (row & mask)));
                    if (!M.coeff_one[v]) w = mul(w, M.coeff[v]);
                    lo[0] = mul(lo[0], w);
                    hi[0] = mul(hi[0], w);
                    Fr q[F];
                    uniform_item<F>(lo, hi, q);
#pragma unroll
                    for (int t = 0; t < F; ++t) acc[t] = add(acc[t], q[t]);
                }
                eng_block_reduce<F>(acc, partials, tk.slot, chunk);
            };
            if (M.F == 2) run(std::integral_constant<int, 2>{});
            else if (M.F == 3) run(std::integral_constant<int, 3>{});
            else run(std::integral_constant<int, 4>{});
        }
        if (fused) binds_done += 1;
        ENG_STAMP(2);
        // ---- completion: the last block to arrive sums the partials (16 lanes per slot) and publishes the round
        if (tid == 0) {
            uint32_t expected = 0;
            for (int k = 0; k < n_tasks; ++k) expected += eng_active_chunks(desc->task[k].items0, desc->task[k].n_blocks, round);
            __threadfence();  // release: partials and bound tables
            uint32_t t = atomicAdd(&sync->done_count, 1u);
            s_flag = (t == expected - 1) ? 1u : 0u;
        }
        __syncthreads();
        ENG_STAMP(3);
        if (s_flag) {
            __threadfence();
            if (trace && tid == 0) sync->stamps[round][4] = __builtin_readcyclecounter();
            for (int base = 0; base < n_slots; base += kBlock / 16) {
                const int sl = base + (int)(tid >> 4), l = tid & 15;
                Fr s = Fr::zero();
                if (sl < n_slots) {
                    const EngTask& st = desc->task[desc->slot_task[sl]];
                    const uint32_t cnt = eng_active_chunks(st.items0, st.n_blocks, round);
                    for (uint32_t c = l; c < cnt; c += 16) s = add(s, ld_fr(partials + (size_t)sl * kEngMaxBlocks + c));
                }
#pragma unroll
                for (int off = 8; off >= 1; off >>= 1) {
                    Fr o;
#pragma unroll
                    for (int k = 0; k < 8; ++k) o.l[k] = __shfl_xor(s.l[k], off, 64);
                    s = add(s, o);
                }
                if (sl < n_slots && l == 0) st_fr(results + sl, s);
            }
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_store(&sync->done_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(flag, seq0 + (uint64_t)round, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // after the round sums
                if (trace) sync->stamps[round][5] = __builtin_readcyclecounter();
            }
        }
    }
#undef ENG_STAMP
}

}  // namespace jolt
