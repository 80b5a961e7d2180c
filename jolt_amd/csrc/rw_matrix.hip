// jolt_amd/csrc/rw_matrix.hip -- the sparse (K x T) read-write matrix of RAM read/write checking on the device.
//
// Replaces CycleMajorMatrix / AddressMajorMatrix and the round messages of RamReadWriteKernel
// (crates/jolt-kernels/src/optimized/rw_matrix.rs:1-690, optimized/ram_read_write.rs:58-330).  Summand
//   eq(tau_low, j) * ra(k,j) * (val(k,j) + gamma * (val(k,j) + inc(j)))   over (address || cycle), bound low-to-high, cycles first;
// ra / val are never materialised at K x T: ONE entry per RAM access, implicit coefficients recovered from prev / next checkpoints.
//
// The reference walks sorted row pairs with a two-pointer merge per group (rayon over groups).  On the device every ENTRY is a
// thread: entries stay sorted by (row, col) as one 64-bit key, an entry finds its partner (same column, sibling row) by binary
// search, and the merged output position of the bind is a closed form of two prefix sums (matched evens / produced outputs):
//     pos(even a)        = out[gs] + (a - gs) + (lb_odd(a.col) - os) - (M[a] - M[gs])
//     pos(unmatched odd) = out[gs] + (b - os) + (lb_even(b.col) - gs) - (M[lb_even(b.col)] - M[gs])
// (gs / os = first even / odd entry of the pair group, M = exclusive count of matched even entries) -- the order the reference's
// merge produces, so the next round can search again.  Gather bound; ~7 multiplies per entry and round.
// Address rounds (all rows 0 after the cycle phase: one entry per touched address): partners are array neighbours, outputs keep
// their order, checkpoints come from the bound val_init column (rw_matrix.rs:599-637).
#include <algorithm>
#include <vector>

#include "ctx.hpp"
#include "ints.hpp"
#include "onehot.hpp"
#include "poly_kernels.hip.h"

using namespace jolt;

int32_t jolt_internal_table_new(jolt_ctx* ctx, size_t len, jolt_table** out);
int32_t jolt_internal_bind(jolt_ctx* ctx, jolt_table* const* tables, size_t k, const Fr& r, int32_t order);
int32_t jolt_internal_eq_levels(jolt_ctx* ctx, const Fr* r, size_t n, const Fr& scale, std::vector<jolt_table*>* levels);
int32_t jolt_internal_ensure_scratch(jolt_ctx* ctx, size_t partials, size_t results);

namespace {

constexpr uint64_t kNoAccess = 0xFFFFFFFFFFFFFFFFull;  // ram_trace.rs:22

// one state of the matrix (structure of arrays); cycle phase: key = row << 32 | col, checkpoints raw u64; address phase: key = col,
// checkpoints promoted to Fr (CycleMajorEntry / AddressMajorEntry, rw_matrix.rs:26-55)
struct RwArrays {
    uint64_t* key;
    uint64_t *prev_u, *next_u;
    Fr *prev_f, *next_f;
    Fr *val, *ra;
    Fr* wa;  // registers mode only: the rd_wa coefficient column (ra then holds gamma * rs1_ra + gamma^2 * rs2_ra)
};

// The first index whose key is >= x, found from a position that is known to be CLOSE to it (the entry itself: the sibling row and the group's bounds lie within one
// group of entries, 2 .. 256 long): gallop away from `hint`, then bisect the bracket -- 2 .. 8 probes in the neighbourhood of the entry instead of
// log2 n = 23 across the whole array, three times per entry and round.
__device__ __forceinline__ uint32_t lower_bound_near(const uint64_t* __restrict__ a, uint32_t n, uint64_t x, uint32_t hint) {
    uint32_t lo, hi;
    if (a[hint] < x) {  // the bound lies after hint
        uint32_t prev = hint, step = 1;
        hi = n;
        for (;;) {
            const uint32_t probe = prev + step;
            if (probe >= n || probe < prev) break;
            if (a[probe] < x) { prev = probe; step <<= 1; }
            else { hi = probe; break; }
        }
        lo = prev + 1;
    } else {  // at or before hint
        uint32_t next = hint, step = 1;
        lo = 0;
        for (;;) {
            if (next < step) break;
            const uint32_t probe = next - step;
            if (a[probe] >= x) { next = probe; step <<= 1; }
            else { lo = probe + 1; break; }
        }
        hi = next;
    }
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ Fr bind_pair(const Fr& lo, const Fr& hi, const Fr& r, int shifted) {
    return shifted ? jolt::bind_pair<true>(lo, hi, r) : jolt::bind_pair<false>(lo, hi, r);
}
__device__ __forceinline__ Fr slope_term(const Fr& val, const Fr& inc, const Fr& gamma) { return add(val, mul(gamma, add(inc, val))); }

// ---- construction: one entry per access, compacted in cycle order (ram_read_write.rs:291-306) --------------------------------
__global__ __launch_bounds__(kBlock) void k_rw_access_flags(const uint64_t* __restrict__ addresses, uint32_t cycles, uint64_t* __restrict__ flags) {
    uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    if (j < cycles) flags[j] = addresses[j] != kNoAccess ? 1ull : 0ull;
}
__global__ __launch_bounds__(kBlock) void k_rw_build(const uint64_t* __restrict__ addresses, const uint64_t* __restrict__ pre, const uint64_t* __restrict__ post,
                                                     uint32_t cycles, const uint64_t* __restrict__ pos, RwArrays o) {
    uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= cycles || addresses[j] == kNoAccess) return;
    const uint32_t p = (uint32_t)pos[j];
    o.key[p] = ((uint64_t)j << 32) | (uint32_t)addresses[j];
    o.prev_u[p] = pre[j];
    o.next_u[p] = post[j];
    st_fr(o.val + p, fr_from_u64(pre[j]));
    st_fr(o.ra + p, Fr::one());
}

// ---- registers mode (optimized/registers_read_write/sparse.rs:406-466): <= 3 cells per cycle, sorted by register -------------------------
// rs2 folds into rs1's cell (ra = gamma + gamma^2), rd into either read's cell (wa = 1, next = the written value)
struct RegCells {
    uint32_t col[3];
    uint64_t prev[3], next[3], val[3];
    uint32_t ra_kind[3];  // 0: none, 1: rs1, 2: rs2, 3: both
    uint32_t wa[3];
    uint32_t len;
};
__device__ __forceinline__ RegCells reg_cells(uint32_t rs1, uint64_t rs1_val, uint32_t rs2, uint64_t rs2_val, uint32_t rd, uint64_t rd_pre, uint64_t rd_post) {
    RegCells c;
    c.len = 0;
    if (rs1 != kColdIdx) { c.col[0] = rs1; c.prev[0] = rs1_val; c.next[0] = rs1_val; c.val[0] = rs1_val; c.ra_kind[0] = 1; c.wa[0] = 0; c.len = 1; }
    if (rs2 != kColdIdx) {
        if (c.len && c.col[0] == rs2) c.ra_kind[0] = 3;
        else { const uint32_t k = c.len++; c.col[k] = rs2; c.prev[k] = rs2_val; c.next[k] = rs2_val; c.val[k] = rs2_val; c.ra_kind[k] = 2; c.wa[k] = 0; }
    }
    if (rd != kColdIdx) {
        uint32_t f = 0;
        while (f < c.len && c.col[f] != rd) ++f;
        if (f < c.len) { c.wa[f] = 1; c.next[f] = rd_post; }
        else { const uint32_t k = c.len++; c.col[k] = rd; c.prev[k] = rd_pre; c.next[k] = rd_post; c.val[k] = rd_pre; c.ra_kind[k] = 0; c.wa[k] = 1; }
    }
    return c;
}
__global__ __launch_bounds__(kBlock) void k_reg_count(const uint8_t* __restrict__ idx, uint32_t wide, uint32_t cycles, uint64_t* __restrict__ flags) {
    const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= cycles) return;
    const uint32_t rs1 = hot_load(hot_col(idx, 0, wide), j, wide), rs2 = hot_load(hot_col(idx, cycles, wide), j, wide), rd = hot_load(hot_col(idx, 2 * (size_t)cycles, wide), j, wide);
    uint32_t count = rs1 != kColdIdx ? 1u : 0u;  // RegisterCycleRow::entry_count (sparse.rs:392-404)
    if (rs2 != kColdIdx && rs2 != rs1) count += 1;
    if (rd != kColdIdx && rd != rs1 && rd != rs2) count += 1;
    flags[j] = count;
}
__global__ __launch_bounds__(kBlock) void k_reg_build(const uint8_t* __restrict__ idx, uint32_t wide, const uint64_t* __restrict__ rs1_val, const uint64_t* __restrict__ rs2_val,
                                                      const uint64_t* __restrict__ rd_pre, const uint64_t* __restrict__ rd_post, uint32_t cycles, const uint64_t* __restrict__ pos,
                                                      Fr gamma, Fr gamma2, RwArrays o) {
    const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= cycles) return;
    const uint32_t rs1 = hot_load(hot_col(idx, 0, wide), j, wide), rs2 = hot_load(hot_col(idx, cycles, wide), j, wide), rd = hot_load(hot_col(idx, 2 * (size_t)cycles, wide), j, wide);
    RegCells c = reg_cells(rs1, rs1_val[j], rs2, rs2_val[j], rd, rd_pre[j], rd_post[j]);
    uint32_t ord[3] = {0, 1, 2};  // sort by register; len <= 3
    for (uint32_t a = 0; a < c.len; ++a)
        for (uint32_t b = a + 1; b < c.len; ++b)
            if (c.col[ord[b]] < c.col[ord[a]]) { const uint32_t t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    const uint32_t p0 = (uint32_t)pos[j];
    for (uint32_t a = 0; a < c.len; ++a) {
        const uint32_t k = ord[a], p = p0 + a;
        o.key[p] = ((uint64_t)j << 32) | c.col[k];
        o.prev_u[p] = c.prev[k];
        o.next_u[p] = c.next[k];
        st_fr(o.val + p, fr_from_u64(c.val[k]));
        const Fr ra = c.ra_kind[k] == 0 ? Fr::zero() : (c.ra_kind[k] == 1 ? gamma : (c.ra_kind[k] == 2 ? gamma2 : add(gamma, gamma2)));
        st_fr(o.ra + p, ra);
        st_fr(o.wa + p, c.wa[k] ? Fr::one() : Fr::zero());
    }
}

// ---- exclusive scan of 64-bit counters (two packed 32-bit sums): block scan, scan of the block sums, add ---------------------------
constexpr int kScanItems = 4;
__global__ __launch_bounds__(kBlock) void k_scan_blocks(const uint64_t* __restrict__ in, uint32_t n, uint64_t* __restrict__ out, uint64_t* __restrict__ block_sums) {
    __shared__ uint64_t sm[kBlock];
    const uint32_t base = (blockIdx.x * kBlock + threadIdx.x) * kScanItems;
    uint64_t v[kScanItems], local = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        v[k] = base + k < n ? in[base + k] : 0ull;
        local += v[k];
    }
    sm[threadIdx.x] = local;
    __syncthreads();
    for (int off = 1; off < kBlock; off <<= 1) {
        uint64_t t = (int)threadIdx.x >= off ? sm[threadIdx.x - off] : 0ull;
        __syncthreads();
        sm[threadIdx.x] += t;
        __syncthreads();
    }
    uint64_t run = sm[threadIdx.x] - local;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == kBlock - 1) block_sums[blockIdx.x] = sm[threadIdx.x];
}
// one workgroup: exclusive scan of the block sums in place; total[0] = the grand total
__global__ __launch_bounds__(kBlock) void k_scan_sums(uint64_t* __restrict__ sums, uint32_t nblocks, uint64_t* __restrict__ total) {
    __shared__ uint64_t sm[kBlock];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += kBlock) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t v = i < nblocks ? sums[i] : 0ull;
        sm[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < kBlock; off <<= 1) {
            uint64_t t = (int)threadIdx.x >= off ? sm[threadIdx.x - off] : 0ull;
            __syncthreads();
            sm[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nblocks) sums[i] = carry + sm[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == kBlock - 1) carry += sm[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) total[0] = carry;
}
__global__ __launch_bounds__(kBlock) void k_scan_add(uint64_t* __restrict__ out, uint32_t n, const uint64_t* __restrict__ block_sums) {
    const uint32_t base = (blockIdx.x * kBlock + threadIdx.x) * kScanItems;
    const uint64_t add_ = block_sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
        if (base + k < n) out[base + k] += add_;
}

// ---- cycle phase -----------------------------------------------------------------------------------------------------------------
// partner[i] = lower bound of (sibling row, col) | matched << 31 is NOT enough (31 bits of index may be needed): two arrays.
// sib_lb[i] = lower bound of this entry's column in the sibling row; flags[i] = produces-an-output | (even && matched) << 32
__global__ __launch_bounds__(kBlock) void k_rw_cycle_match(const uint64_t* __restrict__ key, uint32_t n, uint32_t* __restrict__ sib_lb, uint32_t* __restrict__ matched,
                                                           uint64_t* __restrict__ flags) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = key[i], want = k ^ (1ull << 32);
    const uint32_t lb = lower_bound_near(key, n, want, i);
    const bool m = lb < n && key[lb] == want;
    const bool even = ((k >> 32) & 1) == 0;
    sib_lb[i] = lb;
    matched[i] = m ? 1u : 0u;
    flags[i] = ((even || !m) ? 1ull : 0ull) | ((even && m) ? (1ull << 32) : 0ull);
}

// CycleMajorMatrix::quadratic_coefficients (rw_matrix.rs:287-325) with CycleMajorEntry::quadratic_evals (:116-145) per pair:
// partial sums of head(pair) * [q(0), q_inf]; a matched pair is evaluated by its EVEN entry
template <bool REG>
__global__ __launch_bounds__(kBlock) void k_rw_cycle_round(RwArrays a, uint32_t n, const uint32_t* __restrict__ sib_lb, const uint32_t* __restrict__ matched,
                                                           const Fr* __restrict__ inc, const Fr* __restrict__ e_out, const Fr* __restrict__ e_in, int in_bits, Fr gamma,
                                                           Fr* __restrict__ partials) {
    Fr acc[2] = {Fr::zero(), Fr::zero()};
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint64_t k = a.key[i];
        const uint32_t row = (uint32_t)(k >> 32);
        const bool even = (row & 1) == 0, m = matched[i] != 0;
        if (!even && m) continue;
        const uint32_t pair = row >> 1;
        const Fr inc0 = ld_fr(inc + 2 * (size_t)pair), inc1 = sub(ld_fr(inc + 2 * (size_t)pair + 1), inc0);
        const Fr ra = ld_fr(a.ra + i), val = ld_fr(a.val + i);
        Fr q0, q1;
        if constexpr (REG) {  // SparseEntry::accumulate_pair_evals (registers_read_write/sparse.rs:283-325): ra_t * val_t + wa_t * (val_t + inc_t)
            const Fr wa = ld_fr(a.wa + i);
            if (even) {
                Fr ra_slope, wa_slope, val_slope;
                if (m) {
                    const uint32_t o = sib_lb[i];
                    ra_slope = sub(ld_fr(a.ra + o), ra);
                    wa_slope = sub(ld_fr(a.wa + o), wa);
                    val_slope = sub(ld_fr(a.val + o), val);
                } else {
                    ra_slope = neg(ra);
                    wa_slope = neg(wa);
                    val_slope = sub(fr_from_u64(a.next_u[i]), val);
                }
                q0 = add(mul(ra, val), mul(wa, add(val, inc0)));
                q1 = add(mul(ra_slope, val_slope), mul(wa_slope, add(val_slope, inc1)));
            } else {
                const Fr val_slope = sub(val, fr_from_u64(a.prev_u[i]));
                q0 = Fr::zero();
                q1 = add(mul(ra, val_slope), mul(wa, add(val_slope, inc1)));
            }
        } else if (even) {
            Fr ra_slope, val_slope;
            if (m) {
                const uint32_t o = sib_lb[i];
                ra_slope = sub(ld_fr(a.ra + o), ra);
                val_slope = sub(ld_fr(a.val + o), val);
            } else {
                ra_slope = neg(ra);
                val_slope = sub(fr_from_u64(a.next_u[i]), val);
            }
            q0 = mul(ra, slope_term(val, inc0, gamma));
            q1 = mul(ra_slope, slope_term(val_slope, inc1, gamma));
        } else {
            q0 = Fr::zero();
            q1 = mul(ra, slope_term(sub(val, fr_from_u64(a.prev_u[i])), inc1, gamma));
        }
        const Fr head = mul(ld_fr(e_out + (pair >> in_bits)), ld_fr(e_in + (pair & ((1u << in_bits) - 1))));
        acc[0] = add(acc[0], mul(head, q0));
        acc[1] = add(acc[1], mul(head, q1));
    }
    block_reduce_store<2>(acc, partials);
}

// CycleMajorMatrix::bind (rw_matrix.rs:268-285): every entry writes (at most) one merged entry at its merge rank
template <bool REG>
__global__ __launch_bounds__(kBlock) void k_rw_cycle_bind(RwArrays a, uint32_t n, const uint32_t* __restrict__ sib_lb, const uint32_t* __restrict__ matched,
                                                          const uint64_t* __restrict__ scan, Fr r, int shifted, RwArrays o) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = a.key[i];
    const uint32_t row = (uint32_t)(k >> 32), col = (uint32_t)k;
    const bool even = (row & 1) == 0, m = matched[i] != 0;
    if (!even && m) return;
    const uint64_t gkey = (uint64_t)(row & ~1u) << 32;
    const uint32_t gs = lower_bound_near(a.key, n, gkey, i), os = lower_bound_near(a.key, n, gkey | (1ull << 32), i);
    const uint64_t s_gs = gs < n ? scan[gs] : 0;  // gs < n always (this entry belongs to the group)
    const uint32_t out_base = (uint32_t)s_gs, m_gs = (uint32_t)(s_gs >> 32);
    const uint32_t lb = sib_lb[i];
    uint32_t pos;
    if (even) pos = out_base + (i - gs) + (lb - os) - ((uint32_t)(scan[i] >> 32) - m_gs);
    else pos = out_base + (i - os) + (lb - gs) - ((lb < n ? (uint32_t)(scan[lb] >> 32) : 0u) - m_gs);
    // lb for an unmatched odd entry lies in [gs, os]; scan[os] exists (os <= i < n)
    const Fr ra = ld_fr(a.ra + i), val = ld_fr(a.val + i);
    Fr nra, nval;
    uint64_t np, nn;
    auto lerp = [&](const Fr& lo, const Fr& hi) { return bind_pair(lo, hi, r, shifted); };
    if (even && m) {  // rw_matrix.rs:72-83
        nra = lerp(ra, ld_fr(a.ra + lb));
        nval = lerp(val, ld_fr(a.val + lb));
        np = a.prev_u[i];
        nn = a.next_u[lb];
    } else if (even) {  // :84-94: implicit odd side: ra = 0, val = the even entry's next checkpoint
        nra = lerp(ra, Fr::zero());
        nval = lerp(val, fr_from_u64(a.next_u[i]));
        np = a.prev_u[i];
        nn = a.next_u[i];
    } else {  // :95-105: implicit even side: ra = 0, val = the odd entry's prev checkpoint
        nra = lerp(Fr::zero(), ra);
        nval = lerp(fr_from_u64(a.prev_u[i]), val);
        np = a.prev_u[i];
        nn = a.next_u[i];
    }
    o.key[pos] = ((uint64_t)(row >> 1) << 32) | col;
    o.prev_u[pos] = np;
    o.next_u[pos] = nn;
    st_fr(o.ra + pos, nra);
    st_fr(o.val + pos, nval);
    if constexpr (REG) {  // the write coefficient binds like the read coefficient: a missing side is 0
        const Fr wa = ld_fr(a.wa + i);
        st_fr(o.wa + pos, (even && m) ? lerp(wa, ld_fr(a.wa + lb)) : (even ? lerp(wa, Fr::zero()) : lerp(Fr::zero(), wa)));
    }
}

// CycleMajorMatrix::into_address_major (rw_matrix.rs:327-337): rows are all 0; key = col, checkpoints promoted
__global__ __launch_bounds__(kBlock) void k_rw_to_address_major(RwArrays a, uint32_t n) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    a.key[i] = (uint32_t)a.key[i];
    st_fr(a.prev_f + i, fr_from_u64(a.prev_u[i]));
    st_fr(a.next_f + i, fr_from_u64(a.next_u[i]));
}

// ---- address phase (one row): partners are neighbours ---------------------------------------------------------------------------
__device__ __forceinline__ bool adr_partner(const uint64_t* __restrict__ key, uint32_t n, uint32_t i, uint32_t& other) {
    const uint64_t c = key[i];
    if ((c & 1) == 0) { other = i + 1; return i + 1 < n && key[i + 1] == c + 1; }
    other = i - 1;
    return i > 0 && key[i - 1] == c - 1;
}
__global__ __launch_bounds__(kBlock) void k_rw_address_flags(const uint64_t* __restrict__ key, uint32_t n, uint64_t* __restrict__ flags) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    uint32_t o;
    const bool m = adr_partner(key, n, i, o);
    flags[i] = ((key[i] & 1) == 0 || !m) ? 1ull : 0ull;
}
// AddressMajorMatrix::address_round_evals (rw_matrix.rs:641-676, per pair :388-419): [s(0), s(2)]; eq and inc are cycle-bound scalars
__global__ __launch_bounds__(kBlock) void k_rw_address_round(RwArrays a, uint32_t n, const Fr* __restrict__ val_init, const Fr* __restrict__ inc_scalar, Fr eq_eval,
                                                             Fr gamma, Fr* __restrict__ partials) {
    Fr acc[2] = {Fr::zero(), Fr::zero()};
    const Fr inc = ld_fr(inc_scalar);
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        uint32_t o;
        const bool m = adr_partner(a.key, n, i, o);
        const uint64_t c = a.key[i];
        const bool even = (c & 1) == 0;
        if (!even && m) continue;
        const Fr ra = ld_fr(a.ra + i), val = ld_fr(a.val + i);
        Fr s0, s2;
        if (even) {
            Fr ra2, val2;
            if (m) {
                const Fr ora = ld_fr(a.ra + o), oval = ld_fr(a.val + o);
                ra2 = sub(add(ora, ora), ra);
                val2 = sub(add(oval, oval), val);
            } else {
                const Fr cp = ld_fr(val_init + (c | 1));  // the odd column's checkpoint
                ra2 = neg(ra);
                val2 = sub(add(cp, cp), val);
            }
            s0 = mul(mul(eq_eval, ra), slope_term(val, inc, gamma));
            s2 = mul(mul(eq_eval, ra2), slope_term(val2, inc, gamma));
        } else {
            const Fr cp = ld_fr(val_init + (c & ~1ull));  // the even column's checkpoint
            s0 = Fr::zero();
            s2 = mul(mul(eq_eval, add(ra, ra)), slope_term(sub(add(val, val), cp), inc, gamma));
        }
        acc[0] = add(acc[0], s0);
        acc[1] = add(acc[1], s2);
    }
    block_reduce_store<2>(acc, partials);
}
// AddressMajorMatrix::bind (rw_matrix.rs:599-637, per pair :342-383); val_init is bound by the caller afterwards
__global__ __launch_bounds__(kBlock) void k_rw_address_bind(RwArrays a, uint32_t n, const Fr* __restrict__ val_init, const uint64_t* __restrict__ scan, Fr r, int shifted,
                                                            RwArrays o) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    uint32_t p;
    const bool m = adr_partner(a.key, n, i, p);
    const uint64_t c = a.key[i];
    const bool even = (c & 1) == 0;
    if (!even && m) return;
    const uint32_t pos = (uint32_t)scan[i];
    auto lerp = [&](const Fr& lo, const Fr& hi) { return bind_pair(lo, hi, r, shifted); };
    const Fr ra = ld_fr(a.ra + i), val = ld_fr(a.val + i), pv = ld_fr(a.prev_f + i), nx = ld_fr(a.next_f + i);
    Fr nra, nval, npv, nnx;
    if (even && m) {
        nra = lerp(ra, ld_fr(a.ra + p));
        nval = lerp(val, ld_fr(a.val + p));
        npv = lerp(pv, ld_fr(a.prev_f + p));
        nnx = lerp(nx, ld_fr(a.next_f + p));
    } else if (even) {
        const Fr cp = ld_fr(val_init + (c | 1));
        nra = lerp(ra, Fr::zero());
        nval = lerp(val, cp);
        npv = lerp(pv, cp);
        nnx = lerp(nx, cp);
    } else {
        const Fr cp = ld_fr(val_init + (c & ~1ull));
        nra = lerp(Fr::zero(), ra);
        nval = lerp(cp, val);
        npv = lerp(cp, pv);
        nnx = lerp(cp, nx);
    }
    o.key[pos] = c >> 1;
    st_fr(o.ra + pos, nra);
    st_fr(o.val + pos, nval);
    st_fr(o.prev_f + pos, npv);
    st_fr(o.next_f + pos, nnx);
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// host object
// ------------------------------------------------------------------------------------------------------------------
struct jolt_rw_matrix {
    jolt_ctx* ctx = nullptr;
    size_t log_t = 0, log_k = 0, round = 0;  // round = number of challenges ingested
    uint32_t n = 0, cap = 0;
    void* block = nullptr;  // one pool block holding both SoA states + scratch
    RwArrays st[2];
    int cur = 0;
    uint32_t *sib_lb = nullptr, *matched = nullptr;
    uint64_t *flags = nullptr, *scan = nullptr, *block_sums = nullptr, *total = nullptr;
    uint64_t* h_total = nullptr;  // pinned
    bool match_valid = false;
    bool registers = false;                 // registers read/write checking: two coefficient columns, dense K-sized address phase on the host
    bool hold_row = false;                  // a rank's LOCAL matrix of a sharded prover: after its last cycle round the single row stays cycle-major (jolt_rw_matrix_export_row)
    std::vector<Fr> reg_ra, reg_wa, reg_val;  // the address-phase state (ReadWriteKernel::{ra, wa, val}, registers_read_write/mod.rs:162-169)
    Fr inc_scalar;
    jolt_table *inc = nullptr, *val_init = nullptr;
    Fr gamma;
    // GruenSplitEqPolynomial::new(tau_low, LowToHigh) host half (split_eq.rs:187-363)
    std::vector<Fr> w;
    Fr current_scalar;
    size_t out_len = 0, in_len = 0, e_out_bits = 0, e_in_bits = 0;
    std::vector<jolt_table*> e_out_cache, e_in_cache;
};

static int32_t rw_scan(jolt_rw_matrix* m, uint32_t n) {  // m->scan = exclusive scan of m->flags; *m->total = sum
    jolt_ctx* ctx = m->ctx;
    const uint32_t per = kBlock * kScanItems, nblocks = std::max<uint32_t>(1, (n + per - 1) / per);
    hipLaunchKernelGGL(k_scan_blocks, dim3(nblocks), dim3(kBlock), 0, ctx->stream, (const uint64_t*)m->flags, n, m->scan, m->block_sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kBlock), 0, ctx->stream, m->block_sums, nblocks, m->total);
    hipLaunchKernelGGL(k_scan_add, dim3(nblocks), dim3(kBlock), 0, ctx->stream, m->scan, n, (const uint64_t*)m->block_sums);
    JOLT_HIP_TRY(ctx, hipGetLastError());
    return JOLT_OK;
}
static int32_t rw_read_total(jolt_rw_matrix* m, uint32_t* out) {
    jolt_ctx* ctx = m->ctx;
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(m->h_total, m->total, 8, hipMemcpyDeviceToHost, ctx->stream));
    JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *out = (uint32_t)*m->h_total;
    return JOLT_OK;
}
static unsigned rw_grid(uint32_t n) { return std::max<unsigned>(1, (n + kBlock - 1) / kBlock); }
// workgroups per CU of the round kernels' grid-stride loops (JOLT_RW_GRID_MULT; 4 = one resident set at 4 waves per SIMD)
static uint32_t rw_round_mult() {
    static const uint32_t m = [] {
        const char* e = std::getenv("JOLT_RW_GRID_MULT");
        const int v = e ? std::atoi(e) : 0;
        return (uint32_t)(v > 0 && v <= 64 ? v : 4);
    }();
    return m;
}

extern "C" int32_t jolt_rw_matrix_destroy(jolt_rw_matrix* m) {
    if (!m) return JOLT_OK;
    jolt_ctx* ctx = m->ctx;
    if (m->block) jolt_internal_dev_free(ctx, m->block);
    if (m->h_total) (void)hipHostFree(m->h_total);
    if (m->inc) jolt_table_free(ctx, m->inc);
    if (m->val_init) jolt_table_free(ctx, m->val_init);
    for (jolt_table* t : m->e_out_cache) jolt_table_free(ctx, t);
    for (jolt_table* t : m->e_in_cache) jolt_table_free(ctx, t);
    delete m;
    return JOLT_OK;
}

// the access columns either on the host (uploaded here) or already in HBM (`resident`: device pointers, `resident_cap` accesses counted by the caller)
static int32_t rw_create_impl(jolt_ctx* ctx, const uint64_t* addresses, const uint64_t* pre_values, const uint64_t* post_values, bool resident, uint32_t resident_cap,
                              size_t cycles, const jolt_table* inc, const jolt_table* val_init, const jolt_fr_t* tau_low, const jolt_fr_t* gamma, jolt_rw_matrix** out) {
    if (!ctx || !addresses || !pre_values || !post_values || !inc || !val_init || !tau_low || !gamma || !out) return JOLT_ERR_INVALID_ARG;
    if (cycles < 2 || (cycles & (cycles - 1)) || cycles > ((size_t)1 << 31) || inc->len != cycles) return JOLT_ERR_SIZE_MISMATCH;
    const size_t K = val_init->len;
    if (K == 0 || (K & (K - 1)) || K > ((size_t)1 << 32)) return JOLT_ERR_SIZE_MISMATCH;
    for (size_t j = 0; j < cycles && !resident; ++j)  // RamAccessColumns::validate_addresses (ram_trace.rs:119-131)
        if (addresses[j] != kNoAccess && addresses[j] >= K) { ctx->last_error = "RAM address outside the address space"; return JOLT_ERR_INVALID_ARG; }
    jolt_rw_matrix* m = new (std::nothrow) jolt_rw_matrix();
    if (!m) return JOLT_ERR_OOM;
    m->ctx = ctx;
    while (((size_t)1 << m->log_t) < cycles) m->log_t++;
    while (((size_t)1 << m->log_k) < K) m->log_k++;
    m->gamma = fr_from_abi(gamma);
    m->w.resize(m->log_t);
    for (size_t i = 0; i < m->log_t; ++i) m->w[i] = fr_from_abi(&tau_low[i]);
    m->current_scalar = Fr::one();
    int32_t s = fr_is_canonical(m->gamma) ? JOLT_OK : JOLT_ERR_INVALID_ARG;
    for (size_t i = 0; i < m->log_t && s == JOLT_OK; ++i) if (!fr_is_canonical(m->w[i])) s = JOLT_ERR_INVALID_ARG;
    // split_eq.rs:214-236: head = w[..n-1], out_point = head[..n/2], in_point = the rest; evals_cached tables per prefix length
    const size_t split = m->log_t / 2, head_len = m->log_t - 1;
    m->out_len = std::min(split, head_len);
    m->in_len = head_len - m->out_len;
    m->e_out_bits = m->out_len;
    m->e_in_bits = m->in_len;
    if (s == JOLT_OK) s = jolt_internal_eq_levels(ctx, m->w.data(), m->out_len, Fr::one(), &m->e_out_cache);
    if (s == JOLT_OK) s = jolt_internal_eq_levels(ctx, m->w.data() + m->out_len, m->in_len, Fr::one(), &m->e_in_cache);
    if (s == JOLT_OK) s = jolt_table_clone(ctx, inc, &m->inc);
    if (s == JOLT_OK) s = jolt_table_clone(ctx, val_init, &m->val_init);
    // device layout: uploads (3 x cycles u64) + two states of `cap` entries + scratch
    const uint32_t T = (uint32_t)cycles;
    uint32_t cap = resident_cap;
    for (size_t j = 0; j < cycles && !resident; ++j) cap += addresses[j] != kNoAccess;
    m->cap = std::max<uint32_t>(cap, 1);
    const size_t scan_len = std::max<size_t>(T, m->cap);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    size_t o_up = take(resident ? 256 : 3 * cycles * 8);
    size_t o_st[2][8];
    for (int b = 0; b < 2; ++b) {
        o_st[b][0] = take((size_t)m->cap * 8); o_st[b][1] = take((size_t)m->cap * 8); o_st[b][2] = take((size_t)m->cap * 8);
        o_st[b][3] = take((size_t)m->cap * 32); o_st[b][4] = take((size_t)m->cap * 32); o_st[b][5] = take((size_t)m->cap * 32); o_st[b][6] = take((size_t)m->cap * 32);
        o_st[b][7] = take(256);  // wa: registers mode only (rw_create_registers sizes its own block)
    }
    const size_t o_sib = take((size_t)m->cap * 4), o_match = take((size_t)m->cap * 4), o_flags = take(scan_len * 8), o_scan = take(scan_len * 8),
                 o_bs = take(((scan_len + kBlock * kScanItems - 1) / (kBlock * kScanItems) + 1) * 8), o_total = take(256);
    if (s == JOLT_OK) s = jolt_internal_dev_alloc(ctx, off, &m->block);
    if (s == JOLT_OK && hipHostMalloc((void**)&m->h_total, 64, hipHostMallocDefault) != hipSuccess) s = JOLT_ERR_HIP;
    if (s != JOLT_OK) { jolt_rw_matrix_destroy(m); return s; }
    char* base = (char*)m->block;
    for (int b = 0; b < 2; ++b) {
        m->st[b].key = (uint64_t*)(base + o_st[b][0]); m->st[b].prev_u = (uint64_t*)(base + o_st[b][1]); m->st[b].next_u = (uint64_t*)(base + o_st[b][2]);
        m->st[b].prev_f = (Fr*)(base + o_st[b][3]); m->st[b].next_f = (Fr*)(base + o_st[b][4]); m->st[b].val = (Fr*)(base + o_st[b][5]); m->st[b].ra = (Fr*)(base + o_st[b][6]);
        m->st[b].wa = (Fr*)(base + o_st[b][7]);
    }
    m->sib_lb = (uint32_t*)(base + o_sib); m->matched = (uint32_t*)(base + o_match); m->flags = (uint64_t*)(base + o_flags); m->scan = (uint64_t*)(base + o_scan);
    m->block_sums = (uint64_t*)(base + o_bs); m->total = (uint64_t*)(base + o_total);
    const uint64_t *d_addr = addresses, *d_pre = pre_values, *d_post = post_values;
    hipError_t e = hipSuccess;
    if (!resident) {
        uint64_t* up = (uint64_t*)(base + o_up);
        d_addr = up; d_pre = up + cycles; d_post = up + 2 * cycles;
        e = hipMemcpyAsync(up, addresses, cycles * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(up + cycles, pre_values, cycles * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(up + 2 * cycles, post_values, cycles * 8, hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_rw_access_flags, dim3(rw_grid(T)), dim3(kBlock), 0, ctx->stream, (const uint64_t*)d_addr, T, m->flags);
        e = hipGetLastError();
    }
    if (e == hipSuccess) s = rw_scan(m, T);
    if (e == hipSuccess && s == JOLT_OK) {
        hipLaunchKernelGGL(k_rw_build, dim3(rw_grid(T)), dim3(kBlock), 0, ctx->stream, (const uint64_t*)d_addr, (const uint64_t*)d_pre, (const uint64_t*)d_post, T,
                           (const uint64_t*)m->scan, m->st[0]);
        e = hipGetLastError();
    }
    if (e == hipSuccess && !resident) e = hipStreamSynchronize(ctx->stream);  // the host arrays may be short-lived
    if (e != hipSuccess || s != JOLT_OK) {
        if (e != hipSuccess) { ctx->last_error = std::string("rw matrix: ") + hipGetErrorString(e); s = JOLT_ERR_HIP; }
        jolt_rw_matrix_destroy(m);
        return s;
    }
    m->n = cap;
    *out = m;
    return JOLT_OK;
}
extern "C" int32_t jolt_rw_matrix_create(jolt_ctx* ctx, const uint64_t* addresses, const uint64_t* pre_values, const uint64_t* post_values, size_t cycles,
                                         const jolt_table* inc, const jolt_table* val_init, const jolt_fr_t* tau_low, const jolt_fr_t* gamma, jolt_rw_matrix** out) {
    return rw_create_impl(ctx, addresses, pre_values, post_values, false, 0, cycles, inc, val_init, tau_low, gamma, out);
}
// accesses and out-of-range addresses of a resident column: counters[0], counters[1]
// (grid-stride, one atomic pair per WORKGROUP: one per wavefront of a cycles / 256 grid was 65 536 atomics on the same two words at T = 2^22 -- 0.75 ms for a 32 MiB read)
__global__ __launch_bounds__(kBlock) void k_rw_count_accesses(const uint64_t* __restrict__ addresses, uint32_t cycles, uint64_t K, uint32_t* __restrict__ counters) {
    __shared__ uint32_t s_hit[kBlock / 64], s_bad[kBlock / 64];
    uint32_t hit = 0, bad = 0;
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < cycles; j += gridDim.x * kBlock) {
        const uint64_t a = addresses[j];
        hit += a != kNoAccess ? 1u : 0u;
        bad += (a != kNoAccess && a >= K) ? 1u : 0u;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        hit += __shfl_xor(hit, off, 64);
        bad += __shfl_xor(bad, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { s_hit[threadIdx.x >> 6] = hit; s_bad[threadIdx.x >> 6] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t h = 0, b = 0;
        for (int w = 0; w < kBlock / 64; ++w) { h += s_hit[w]; b += s_bad[w]; }
        if (h) atomicAdd(&counters[0], h);
        if (b) atomicAdd(&counters[1], b);
    }
}
// The same member over access columns that are ALREADY in HBM (three u64 jolt_ints of `cycles` entries: the witness is uploaded once per
// trace, not once per proof): no host pass over the columns, one 8-byte read-back for the entry count.
extern "C" int32_t jolt_rw_matrix_create_resident(jolt_ctx* ctx, const jolt_ints* addresses, const jolt_ints* pre_values, const jolt_ints* post_values,
                                                  const jolt_table* inc, const jolt_table* val_init, const jolt_fr_t* tau_low, const jolt_fr_t* gamma, jolt_rw_matrix** out) {
    if (!ctx || !addresses || !pre_values || !post_values || !val_init || !out) return JOLT_ERR_INVALID_ARG;
    if (addresses->kind != JOLT_INT_U64 || pre_values->kind != JOLT_INT_U64 || post_values->kind != JOLT_INT_U64) return JOLT_ERR_INVALID_ARG;
    const size_t cycles = addresses->count;
    if (pre_values->count != cycles || post_values->count != cycles) return JOLT_ERR_SIZE_MISMATCH;
    if (cycles < 2 || cycles > ((size_t)1 << 31)) return JOLT_ERR_SIZE_MISMATCH;
    uint32_t* d_counters = nullptr;
    JOLT_TRY(jolt_internal_dev_alloc(ctx, 256, (void**)&d_counters));
    uint32_t h_counters[2] = {0, 0};
    hipError_t e = hipMemsetAsync(d_counters, 0, 8, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_rw_count_accesses, dim3(std::min<uint32_t>(rw_grid((uint32_t)cycles), (uint32_t)ctx->num_cus * 8)), dim3(kBlock), 0, ctx->stream, (const uint64_t*)addresses->data, (uint32_t)cycles,
                           (uint64_t)val_init->len, d_counters);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h_counters, d_counters, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    jolt_internal_dev_free(ctx, d_counters);
    if (e != hipSuccess) { ctx->last_error = std::string("rw matrix: ") + hipGetErrorString(e); return JOLT_ERR_HIP; }
    if (h_counters[1]) { ctx->last_error = "RAM address outside the address space"; return JOLT_ERR_INVALID_ARG; }
    return rw_create_impl(ctx, (const uint64_t*)addresses->data, (const uint64_t*)pre_values->data, (const uint64_t*)post_values->data, true, h_counters[0], cycles, inc,
                          val_init, tau_low, gamma, out);
}

// RamReadWriteKernel::ingest (ram_read_write.rs:104-143)
static int32_t rw_ingest(jolt_rw_matrix* m, const Fr& r) {
    jolt_ctx* ctx = m->ctx;
    const int shifted = fr_low_limbs_zero(r) ? 1 : 0;
    RwArrays &a = m->st[m->cur], &o = m->st[1 - m->cur];
    if (m->round < m->log_t) {
        if (m->n) {
            if (!m->match_valid) {
                hipLaunchKernelGGL(k_rw_cycle_match, dim3(rw_grid(m->n)), dim3(kBlock), 0, ctx->stream, (const uint64_t*)a.key, m->n, m->sib_lb, m->matched, m->flags);
                JOLT_HIP_TRY(ctx, hipGetLastError());
            }
            JOLT_TRY(rw_scan(m, m->n));
            if (m->registers)
                hipLaunchKernelGGL(k_rw_cycle_bind<true>, dim3(rw_grid(m->n)), dim3(kBlock), 0, ctx->stream, a, m->n, (const uint32_t*)m->sib_lb, (const uint32_t*)m->matched,
                                   (const uint64_t*)m->scan, r, shifted, o);
            else
                hipLaunchKernelGGL(k_rw_cycle_bind<false>, dim3(rw_grid(m->n)), dim3(kBlock), 0, ctx->stream, a, m->n, (const uint32_t*)m->sib_lb, (const uint32_t*)m->matched,
                                   (const uint64_t*)m->scan, r, shifted, o);
            JOLT_HIP_TRY(ctx, hipGetLastError());
            JOLT_TRY(rw_read_total(m, &m->n));
            m->cur = 1 - m->cur;
        }
        m->match_valid = false;
        JOLT_TRY(jolt_internal_bind(ctx, &m->inc, 1, r, JOLT_ORDER_LOW_TO_HIGH));
        {   // GruenSplitEqPolynomial::bind (split_eq.rs:334-363)
            const size_t nvar = m->log_t;
            size_t current_index = nvar - m->round;
            const Fr p = m->w[current_index - 1], prod = mul(p, r);
            m->current_scalar = mul(m->current_scalar, add(add(sub(sub(Fr::one(), p), r), prod), prod));
            current_index -= 1;
            if (nvar / 2 < current_index && m->e_in_bits > 0) m->e_in_bits -= 1;
            else if (0 < current_index && m->e_out_bits > 0) m->e_out_bits -= 1;
        }
        if (m->round == m->log_t - 1 && m->hold_row) {
            // a sharded prover's local matrix: the row is exported as it is and merged with the other ranks' rows (jolt_rw_matrix_create_merged)
        } else if (m->round == m->log_t - 1 && m->registers) {
            // SparseEntries::into_dense (registers_read_write/sparse.rs:532-561): the single remaining row scattered into K-sized arrays on the HOST
            // ("small fixed K": the address rounds cost O(K) = 128 pairs, mod.rs:30-33); eq and inc are scalars from here on
            const size_t K = (size_t)1 << m->log_k;
            m->reg_ra.assign(K, Fr::zero());
            m->reg_wa.assign(K, Fr::zero());
            m->reg_val.assign(K, Fr::zero());
            const uint32_t n = m->n;
            if (n > K) { ctx->last_error = "registers matrix: more cells than registers after the cycle rounds"; return JOLT_ERR_INVALID_ARG; }
            std::vector<uint64_t> key(n);
            std::vector<Fr> ra(n), wa(n), val(n);
            const RwArrays& c = m->st[m->cur];
            // the destinations are locals: collect the first error and ALWAYS synchronise before leaving the scope (a copy may still be in flight)
            hipError_t first = hipSuccess;
            auto copy = [&](void* dst, const void* src, size_t bytes) {
                const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
                if (first == hipSuccess) first = e;
            };
            if (n) {
                copy(key.data(), c.key, (size_t)n * 8);
                copy(ra.data(), c.ra, (size_t)n * sizeof(Fr));
                copy(wa.data(), c.wa, (size_t)n * sizeof(Fr));
                copy(val.data(), c.val, (size_t)n * sizeof(Fr));
            }
            copy(&m->inc_scalar, m->inc->data(), sizeof(Fr));
            const hipError_t sync = hipStreamSynchronize(ctx->stream);
            JOLT_HIP_TRY(ctx, first);
            JOLT_HIP_TRY(ctx, sync);
            for (uint32_t i = 0; i < n; ++i) {
                const uint32_t col = (uint32_t)key[i];
                if ((key[i] >> 32) != 0 || col >= K) { ctx->last_error = "registers matrix: a cell outside the single bound row"; return JOLT_ERR_INVALID_ARG; }
                m->reg_ra[col] = ra[i]; m->reg_wa[col] = wa[i]; m->reg_val[col] = val[i];
            }
        } else if (m->round == m->log_t - 1 && m->n) {
            hipLaunchKernelGGL(k_rw_to_address_major, dim3(rw_grid(m->n)), dim3(kBlock), 0, ctx->stream, m->st[m->cur], m->n);
            JOLT_HIP_TRY(ctx, hipGetLastError());
        }
    } else if (m->registers) {  // bind_pairs over the three dense arrays (mod.rs:262-265)
        for (std::vector<Fr>* t : {&m->reg_ra, &m->reg_wa, &m->reg_val}) {
            const size_t half = t->size() / 2;
            for (size_t y = 0; y < half; ++y) (*t)[y] = add((*t)[2 * y], mul(r, sub((*t)[2 * y + 1], (*t)[2 * y])));
            t->resize(half);
        }
    } else {
        if (m->n) {
            hipLaunchKernelGGL(k_rw_address_flags, dim3(rw_grid(m->n)), dim3(kBlock), 0, ctx->stream, (const uint64_t*)a.key, m->n, m->flags);
            JOLT_HIP_TRY(ctx, hipGetLastError());
            JOLT_TRY(rw_scan(m, m->n));
            hipLaunchKernelGGL(k_rw_address_bind, dim3(rw_grid(m->n)), dim3(kBlock), 0, ctx->stream, a, m->n, (const Fr*)m->val_init->data(), (const uint64_t*)m->scan, r,
                               shifted, o);
            JOLT_HIP_TRY(ctx, hipGetLastError());
            JOLT_TRY(rw_read_total(m, &m->n));
            m->cur = 1 - m->cur;
        }
        JOLT_TRY(jolt_internal_bind(ctx, &m->val_init, 1, r, JOLT_ORDER_LOW_TO_HIGH));
    }
    m->round += 1;
    return JOLT_OK;
}

// ProveRounds::prove_round (ram_read_write.rs:196-214), device half.  Cycle rounds: evals_out = (q(0), q_inf) of the quadratic factor
// and aux_out = {current_scalar, w[current_index - 1], 0} for gruen_poly_deg_3; address rounds: evals_out = (s(0), s(2)), s(1) from
// the claim (UnivariatePoly::from_evals_and_hint), aux_out = 0.
extern "C" int32_t jolt_rw_matrix_prove_round(jolt_rw_matrix* m, const jolt_fr_t* bind, jolt_fr_t* evals_out, jolt_fr_t* aux_out) {
    if (!m || !evals_out) return JOLT_ERR_INVALID_ARG;
    jolt_ctx* ctx = m->ctx;
    if (m->registers) { ctx->last_error = "a registers handle: use jolt_registers_rw_prove_round (this entry point ignores the wa column)"; return JOLT_ERR_INVALID_ARG; }
    if (bind) {
        Fr r = fr_from_abi(bind);
        JOLT_REQUIRE(ctx, fr_is_canonical(r), "bind challenge is not a canonical Fr");
        if (m->round >= m->log_t + m->log_k) { ctx->last_error = "rw matrix already fully bound"; return JOLT_ERR_INVALID_ARG; }
        JOLT_TRY(rw_ingest(m, r));
    }
    if (m->round >= m->log_t + m->log_k) { ctx->last_error = "prove_round on a fully bound rw matrix"; return JOLT_ERR_INVALID_ARG; }
    const int grid = (int)std::max<uint32_t>(1, std::min<uint32_t>((m->n + kBlock - 1) / kBlock, (uint32_t)ctx->num_cus * rw_round_mult()));
    JOLT_TRY(jolt_internal_ensure_scratch(ctx, (size_t)grid * 2 + 8, 8));
    RwArrays& a = m->st[m->cur];
    Fr zero = Fr::zero();
    if (m->round < m->log_t) {
        if (m->n && !m->match_valid) {
            hipLaunchKernelGGL(k_rw_cycle_match, dim3(rw_grid(m->n)), dim3(kBlock), 0, ctx->stream, (const uint64_t*)a.key, m->n, m->sib_lb, m->matched, m->flags);
            JOLT_HIP_TRY(ctx, hipGetLastError());
            m->match_valid = true;
        }
        hipLaunchKernelGGL(k_rw_cycle_round<false>, dim3(grid), dim3(kBlock), 0, ctx->stream, a, m->n, (const uint32_t*)m->sib_lb, (const uint32_t*)m->matched,
                           (const Fr*)m->inc->data(), (const Fr*)m->e_out_cache[m->e_out_bits]->data(), (const Fr*)m->e_in_cache[m->e_in_bits]->data(), (int)m->e_in_bits,
                           m->gamma, ctx->d_partials);
        if (aux_out) {
            fr_to_abi(&aux_out[0], m->current_scalar);
            fr_to_abi(&aux_out[1], m->w[m->log_t - m->round - 1]);
            fr_to_abi(&aux_out[2], zero);
        }
    } else {
        hipLaunchKernelGGL(k_rw_address_round, dim3(grid), dim3(kBlock), 0, ctx->stream, a, m->n, (const Fr*)m->val_init->data(), (const Fr*)m->inc->data(),
                           m->current_scalar, m->gamma, ctx->d_partials);
        if (aux_out) for (int k = 0; k < 3; ++k) fr_to_abi(&aux_out[k], zero);
    }
    JOLT_HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(kBlock), 0, ctx->stream, (const Fr*)ctx->d_partials, grid, 2, ctx->d_results);
    JOLT_HIP_TRY(ctx, hipGetLastError());
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(ctx->h_results, ctx->d_results, 2 * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(evals_out, ctx->h_results, 2 * sizeof(Fr));
    return JOLT_OK;
}

extern "C" int32_t jolt_rw_matrix_finish(jolt_rw_matrix* m, const jolt_fr_t* bind) {
    if (!m || !bind) return JOLT_ERR_INVALID_ARG;
    Fr r = fr_from_abi(bind);
    JOLT_REQUIRE(m->ctx, fr_is_canonical(r), "bind challenge is not a canonical Fr");
    if (m->round + 1 != m->log_t + m->log_k) { m->ctx->last_error = "finish_rounds before the last round"; return JOLT_ERR_INVALID_ARG; }
    return rw_ingest(m, r);
}

// RamReadWriteOutputClaims (ram_read_write.rs:222-240) + the bound cycle-eq factor (validate_derived_tables, :245-266): {ra, val, inc, eq}
extern "C" int32_t jolt_rw_matrix_final_values(jolt_rw_matrix* m, jolt_fr_t* out) {
    if (!m || !out) return JOLT_ERR_INVALID_ARG;
    jolt_ctx* ctx = m->ctx;
    if (m->registers) { ctx->last_error = "a registers handle: use jolt_registers_rw_final_values (no val_init column here)"; return JOLT_ERR_INVALID_ARG; }
    if (m->round != m->log_t + m->log_k) return JOLT_ERR_NOT_FULLY_BOUND;
    Fr vals[3];
    if (m->n) {  // AddressMajorMatrix::final_values (rw_matrix.rs:680-688): at most one entry remains
        JOLT_HIP_TRY(ctx, hipMemcpyAsync(&vals[0], m->st[m->cur].ra, sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
        JOLT_HIP_TRY(ctx, hipMemcpyAsync(&vals[1], m->st[m->cur].val, sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    } else {
        vals[0] = Fr::zero();
        JOLT_HIP_TRY(ctx, hipMemcpyAsync(&vals[1], m->val_init->data(), sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    }
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(&vals[2], m->inc->data(), sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 3; ++k) fr_to_abi(&out[k], vals[k]);
    fr_to_abi(&out[3], m->current_scalar);
    return JOLT_OK;
}

extern "C" int32_t jolt_rw_matrix_len(const jolt_rw_matrix* m, size_t* entries) {
    if (!m || !entries) return JOLT_ERR_INVALID_ARG;
    *entries = m->n;
    return JOLT_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// The sharded form (one process per GPU).  The cycles of a trace are dealt to the ranks in contiguous blocks and the cycle variables bind low to
// high, so the first log T_local cycle rounds touch only a rank's own cells: rank g runs them on a LOCAL matrix over its block (the ordinary constructors with the low
// log T_local coordinates of the point; its two round sums, scaled by eq(w_hi, g), add up over the ranks).  After them every rank holds ONE row -- at most one cell per
// address it touched, with its raw checkpoints -- which jolt_rw_matrix_export_row reads out; the rows of all ranks, stacked in rank order, ARE the cycle-major matrix
// of the remaining log G cycle variables (row = rank), and jolt_rw_matrix_create_merged continues from it on every rank: log G cycle rounds, then the address rounds.
//   hold_row: call before the rounds; then the last local bind leaves the row cycle-major (no address-major / dense conversion).
//   bind: ingest a challenge without asking for a round (the last local challenge).
// ------------------------------------------------------------------------------------------------------------------
extern "C" int32_t jolt_rw_matrix_hold_row(jolt_rw_matrix* m) {
    if (!m || m->round != 0) return JOLT_ERR_INVALID_ARG;
    m->hold_row = true;
    return JOLT_OK;
}
extern "C" int32_t jolt_rw_matrix_bind(jolt_rw_matrix* m, const jolt_fr_t* bind) {
    if (!m || !bind) return JOLT_ERR_INVALID_ARG;
    const Fr r = fr_from_abi(bind);
    JOLT_REQUIRE(m->ctx, fr_is_canonical(r), "bind challenge is not a canonical Fr");
    if (m->round >= m->log_t + m->log_k || (m->hold_row && m->round >= m->log_t)) { m->ctx->last_error = "rw matrix already bound"; return JOLT_ERR_INVALID_ARG; }
    return rw_ingest(m, r);
}
// the single row a held matrix is left with after its log_t cycle rounds: *n_out cells (<= cap, else JOLT_ERR_SIZE_MISMATCH) in column order; wa: registers handles
// only (NULL otherwise); inc_out = the bound increment column's one entry, scalar_out = the bound eq factor so far
extern "C" int32_t jolt_rw_matrix_export_row(jolt_rw_matrix* m, size_t cap, uint64_t* cols, uint64_t* prev, uint64_t* next, jolt_fr_t* val, jolt_fr_t* ra, jolt_fr_t* wa,
                                             jolt_fr_t* inc_out, jolt_fr_t* scalar_out, size_t* n_out) {
    if (!m || !cols || !prev || !next || !val || !ra || !inc_out || !scalar_out || !n_out || (m->registers && !wa)) return JOLT_ERR_INVALID_ARG;
    jolt_ctx* ctx = m->ctx;
    if (!m->hold_row || m->round != m->log_t) { ctx->last_error = "export_row: a held matrix after its last cycle round"; return JOLT_ERR_INVALID_ARG; }
    const uint32_t n = m->n;
    *n_out = n;
    if (n > cap) return JOLT_ERR_SIZE_MISMATCH;
    const RwArrays& a = m->st[m->cur];
    std::vector<uint64_t> key(n);
    hipError_t first = hipSuccess;
    auto copy = [&](void* dst, const void* src, size_t bytes) {
        const hipError_t e = bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream) : hipSuccess;
        if (first == hipSuccess) first = e;
    };
    copy(key.data(), a.key, (size_t)n * 8);
    copy(prev, a.prev_u, (size_t)n * 8);
    copy(next, a.next_u, (size_t)n * 8);
    copy(val, a.val, (size_t)n * sizeof(Fr));
    copy(ra, a.ra, (size_t)n * sizeof(Fr));
    if (m->registers) copy(wa, a.wa, (size_t)n * sizeof(Fr));
    Fr inc;
    copy(&inc, m->inc->data(), sizeof(Fr));
    const hipError_t sync = hipStreamSynchronize(ctx->stream);
    JOLT_HIP_TRY(ctx, first);
    JOLT_HIP_TRY(ctx, sync);
    for (uint32_t i = 0; i < n; ++i) {
        if (key[i] >> 32) { ctx->last_error = "export_row: a cell outside the single bound row"; return JOLT_ERR_INVALID_ARG; }
        cols[i] = (uint32_t)key[i];
    }
    fr_to_abi(inc_out, inc);
    fr_to_abi(scalar_out, m->current_scalar);
    return JOLT_OK;
}

// The matrix over the remaining 2^log_rows rows (log_rows >= 1), from cells in (row, col) order: key, raw checkpoints, val, ra (and wa: registers != 0) as
// exported; inc: the 2^log_rows entries of the bound increment column (row order); w: the log_rows remaining coordinates of the cycle point (the HIGH ones);
// scalar: the eq factor bound so far.  RAM (registers == 0): val_init as in jolt_rw_matrix_create; registers: k = 2^log_k <= 256 registers, val_init NULL.
extern "C" int32_t jolt_rw_matrix_create_merged(jolt_ctx* ctx, int32_t registers, size_t log_rows, size_t log_k, size_t n, const uint64_t* rows, const uint64_t* cols,
                                                const uint64_t* prev, const uint64_t* next, const jolt_fr_t* val, const jolt_fr_t* ra, const jolt_fr_t* wa, const jolt_fr_t* inc,
                                                const jolt_table* val_init, const jolt_fr_t* w, const jolt_fr_t* scalar, const jolt_fr_t* gamma, jolt_rw_matrix** out) {
    if (!ctx || !inc || !w || !scalar || !gamma || !out || (n && (!rows || !cols || !prev || !next || !val || !ra))) return JOLT_ERR_INVALID_ARG;
    if ((registers && n && !wa) || (!registers && !val_init)) return JOLT_ERR_INVALID_ARG;
    if (log_rows < 1 || log_rows > 16 || log_k > 32 || n >= ((size_t)1 << 31)) return JOLT_ERR_UNSUPPORTED;
    if (registers && log_k > 8) return JOLT_ERR_SIZE_MISMATCH;
    if (!registers && val_init->len != ((size_t)1 << log_k)) return JOLT_ERR_SIZE_MISMATCH;
    std::vector<uint64_t> key(n);
    for (size_t i = 0; i < n; ++i) {
        if (rows[i] >> log_rows || cols[i] >> log_k) { ctx->last_error = "merged rw matrix: a cell outside the matrix"; return JOLT_ERR_INVALID_ARG; }
        key[i] = (rows[i] << 32) | cols[i];
        if (i && key[i] <= key[i - 1]) { ctx->last_error = "merged rw matrix: cells must come in (row, column) order"; return JOLT_ERR_INVALID_ARG; }
    }
    jolt_rw_matrix* m = new (std::nothrow) jolt_rw_matrix();
    if (!m) return JOLT_ERR_OOM;
    m->ctx = ctx;
    m->registers = registers != 0;
    m->log_t = log_rows;
    m->log_k = log_k;
    m->gamma = fr_from_abi(gamma);
    m->current_scalar = fr_from_abi(scalar);
    m->w.resize(log_rows);
    for (size_t i = 0; i < log_rows; ++i) m->w[i] = fr_from_abi(&w[i]);
    int32_t s = fr_is_canonical(m->gamma) && fr_is_canonical(m->current_scalar) ? JOLT_OK : JOLT_ERR_INVALID_ARG;
    for (size_t i = 0; i < log_rows && s == JOLT_OK; ++i) if (!fr_is_canonical(m->w[i])) s = JOLT_ERR_INVALID_ARG;
    const size_t split = m->log_t / 2, head_len = m->log_t - 1;  // GruenSplitEqPolynomial::new over the remaining point (split_eq.rs:214-236)
    m->out_len = std::min(split, head_len);
    m->in_len = head_len - m->out_len;
    m->e_out_bits = m->out_len;
    m->e_in_bits = m->in_len;
    if (s == JOLT_OK) s = jolt_internal_eq_levels(ctx, m->w.data(), m->out_len, Fr::one(), &m->e_out_cache);
    if (s == JOLT_OK) s = jolt_internal_eq_levels(ctx, m->w.data() + m->out_len, m->in_len, Fr::one(), &m->e_in_cache);
    if (s == JOLT_OK) s = jolt_table_upload(ctx, inc, (size_t)1 << log_rows, &m->inc);
    if (s == JOLT_OK && !registers) s = jolt_table_clone(ctx, val_init, &m->val_init);
    m->cap = (uint32_t)std::max<size_t>(n, 1);
    const size_t scan_len = std::max<size_t>((size_t)1 << log_rows, m->cap);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    size_t o_st[2][8];
    for (int b = 0; b < 2; ++b) {
        o_st[b][0] = take((size_t)m->cap * 8); o_st[b][1] = take((size_t)m->cap * 8); o_st[b][2] = take((size_t)m->cap * 8);
        o_st[b][3] = take(registers ? 256 : (size_t)m->cap * 32); o_st[b][4] = take(registers ? 256 : (size_t)m->cap * 32);
        o_st[b][5] = take((size_t)m->cap * 32); o_st[b][6] = take((size_t)m->cap * 32); o_st[b][7] = take(registers ? (size_t)m->cap * 32 : 256);
    }
    const size_t o_sib = take((size_t)m->cap * 4), o_match = take((size_t)m->cap * 4), o_flags = take(scan_len * 8), o_scan = take(scan_len * 8),
                 o_bs = take(((scan_len + kBlock * kScanItems - 1) / (kBlock * kScanItems) + 1) * 8), o_total = take(256);
    if (s == JOLT_OK) s = jolt_internal_dev_alloc(ctx, off, &m->block);
    if (s == JOLT_OK && hipHostMalloc((void**)&m->h_total, 64, hipHostMallocDefault) != hipSuccess) s = JOLT_ERR_HIP;
    if (s != JOLT_OK) { jolt_rw_matrix_destroy(m); return s; }
    char* base = (char*)m->block;
    for (int b = 0; b < 2; ++b) {
        m->st[b].key = (uint64_t*)(base + o_st[b][0]); m->st[b].prev_u = (uint64_t*)(base + o_st[b][1]); m->st[b].next_u = (uint64_t*)(base + o_st[b][2]);
        m->st[b].prev_f = registers ? nullptr : (Fr*)(base + o_st[b][3]); m->st[b].next_f = registers ? nullptr : (Fr*)(base + o_st[b][4]);
        m->st[b].val = (Fr*)(base + o_st[b][5]); m->st[b].ra = (Fr*)(base + o_st[b][6]); m->st[b].wa = (Fr*)(base + o_st[b][7]);
    }
    m->sib_lb = (uint32_t*)(base + o_sib); m->matched = (uint32_t*)(base + o_match); m->flags = (uint64_t*)(base + o_flags); m->scan = (uint64_t*)(base + o_scan);
    m->block_sums = (uint64_t*)(base + o_bs); m->total = (uint64_t*)(base + o_total);
    hipError_t first = hipSuccess;
    auto up = [&](void* dst, const void* src, size_t bytes) {
        const hipError_t e = bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream) : hipSuccess;
        if (first == hipSuccess) first = e;
    };
    up(m->st[0].key, key.data(), n * 8);
    up(m->st[0].prev_u, prev, n * 8);
    up(m->st[0].next_u, next, n * 8);
    up(m->st[0].val, val, n * sizeof(Fr));
    up(m->st[0].ra, ra, n * sizeof(Fr));
    if (registers) up(m->st[0].wa, wa, n * sizeof(Fr));
    const hipError_t sync = hipStreamSynchronize(ctx->stream);  // the caller's arrays (and `key`) may be short-lived
    if (first != hipSuccess || sync != hipSuccess) {
        ctx->last_error = std::string("merged rw matrix: ") + hipGetErrorString(first != hipSuccess ? first : sync);
        jolt_rw_matrix_destroy(m);
        return JOLT_ERR_HIP;
    }
    m->n = (uint32_t)n;
    *out = m;
    return JOLT_OK;
}

// test hook: the current entries (rows, cols as u64; val, ra, prev, next as Fr -- raw checkpoints promoted in the cycle phase)
extern "C" int32_t jolt_rw_matrix_download(jolt_rw_matrix* m, uint64_t* rows, uint64_t* cols, jolt_fr_t* val, jolt_fr_t* ra, jolt_fr_t* prev, jolt_fr_t* next) {
    if (!m || !rows || !cols || !val || !ra || !prev || !next) return JOLT_ERR_INVALID_ARG;
    jolt_ctx* ctx = m->ctx;
    if (m->registers) { ctx->last_error = "a registers handle: use jolt_registers_rw_download (no prev / next field columns here)"; return JOLT_ERR_INVALID_ARG; }
    const uint32_t n = m->n;
    if (!n) return JOLT_OK;
    const RwArrays& a = m->st[m->cur];
    const bool address_major = m->round >= m->log_t;
    std::vector<uint64_t> key(n), pu(n), nu(n);
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(key.data(), a.key, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(val, a.val, (size_t)n * 32, hipMemcpyDeviceToHost, ctx->stream));
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(ra, a.ra, (size_t)n * 32, hipMemcpyDeviceToHost, ctx->stream));
    if (address_major) {
        JOLT_HIP_TRY(ctx, hipMemcpyAsync(prev, a.prev_f, (size_t)n * 32, hipMemcpyDeviceToHost, ctx->stream));
        JOLT_HIP_TRY(ctx, hipMemcpyAsync(next, a.next_f, (size_t)n * 32, hipMemcpyDeviceToHost, ctx->stream));
    } else {
        JOLT_HIP_TRY(ctx, hipMemcpyAsync(pu.data(), a.prev_u, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
        JOLT_HIP_TRY(ctx, hipMemcpyAsync(nu.data(), a.next_u, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t i = 0; i < n; ++i) {
        rows[i] = address_major ? 0 : key[i] >> 32;
        cols[i] = address_major ? key[i] : (uint32_t)key[i];
        if (!address_major) {
            Fr p = fr_from_u64(pu[i]), q = fr_from_u64(nu[i]);
            fr_to_abi(&prev[i], p);
            fr_to_abi(&next[i], q);
        }
    }
    return JOLT_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Registers read/write checking (stage 4) on the same machinery: optimized/registers_read_write/{mod,sparse,rows}.rs.
// The cycle phase is the sparse matrix above with two coefficient columns (k_rw_cycle_round<true> / k_rw_cycle_bind<true>); after the
// log T cycle rounds at most K = 2^log_k cells are left and the address rounds run over three K-sized arrays on the host.
// ------------------------------------------------------------------------------------------------------------------
extern "C" int32_t jolt_registers_rw_create(jolt_ctx* ctx, const jolt_onehot* regs, const jolt_ints* rs1_val, const jolt_ints* rs2_val, const jolt_ints* rd_pre,
                                            const jolt_ints* rd_post, const jolt_table* inc, const jolt_fr_t* r_cycle, const jolt_fr_t* gamma, jolt_rw_matrix** out) {
    if (!ctx || !regs || !rs1_val || !rs2_val || !rd_pre || !rd_post || !inc || !r_cycle || !gamma || !out) return JOLT_ERR_INVALID_ARG;
    if (regs->n_polys != 3) { ctx->last_error = "registers columns: a hot-index source with the three columns rs1, rs2, rd"; return JOLT_ERR_INVALID_ARG; }
    const size_t cycles = regs->cycles, K = regs->k;
    for (const jolt_ints* c : {rs1_val, rs2_val, rd_pre, rd_post}) {
        if (c->kind != JOLT_INT_U64) return JOLT_ERR_INVALID_ARG;
        if (c->count != cycles) return JOLT_ERR_SIZE_MISMATCH;
    }
    if (cycles < 2 || (cycles & (cycles - 1)) || cycles > ((size_t)1 << 30) || inc->len != cycles) return JOLT_ERR_SIZE_MISMATCH;
    if (K == 0 || (K & (K - 1)) || K > 256) return JOLT_ERR_SIZE_MISMATCH;  // REGISTER_ADDRESS_BITS = 7
    jolt_rw_matrix* m = new (std::nothrow) jolt_rw_matrix();
    if (!m) return JOLT_ERR_OOM;
    m->ctx = ctx;
    m->registers = true;
    while (((size_t)1 << m->log_t) < cycles) m->log_t++;
    while (((size_t)1 << m->log_k) < K) m->log_k++;
    m->gamma = fr_from_abi(gamma);
    m->w.resize(m->log_t);
    for (size_t i = 0; i < m->log_t; ++i) m->w[i] = fr_from_abi(&r_cycle[i]);
    m->current_scalar = Fr::one();
    int32_t s = fr_is_canonical(m->gamma) ? JOLT_OK : JOLT_ERR_INVALID_ARG;
    for (size_t i = 0; i < m->log_t && s == JOLT_OK; ++i) if (!fr_is_canonical(m->w[i])) s = JOLT_ERR_INVALID_ARG;
    const size_t split = m->log_t / 2, head_len = m->log_t - 1;  // GruenSplitEqPolynomial::new (split_eq.rs:214-236)
    m->out_len = std::min(split, head_len);
    m->in_len = head_len - m->out_len;
    m->e_out_bits = m->out_len;
    m->e_in_bits = m->in_len;
    if (s == JOLT_OK) s = jolt_internal_eq_levels(ctx, m->w.data(), m->out_len, Fr::one(), &m->e_out_cache);
    if (s == JOLT_OK) s = jolt_internal_eq_levels(ctx, m->w.data() + m->out_len, m->in_len, Fr::one(), &m->e_in_cache);
    if (s == JOLT_OK) s = jolt_table_clone(ctx, inc, &m->inc);
    const uint32_t T = (uint32_t)cycles;
    // the cell count needs one scan of the per-cycle counts; the states are sized for the worst case of 3 cells per cycle only if the count says so
    const size_t scan_len0 = T;
    uint64_t *d_flags0 = nullptr;
    if (s == JOLT_OK) s = jolt_internal_dev_alloc(ctx, (2 * scan_len0 + ((scan_len0 + kBlock * kScanItems - 1) / (kBlock * kScanItems) + 1) + 32) * 8, (void**)&d_flags0);
    if (s == JOLT_OK && hipHostMalloc((void**)&m->h_total, 64, hipHostMallocDefault) != hipSuccess) s = JOLT_ERR_HIP;
    if (s != JOLT_OK) { if (d_flags0) jolt_internal_dev_free(ctx, d_flags0); jolt_rw_matrix_destroy(m); return s; }
    m->flags = d_flags0; m->scan = d_flags0 + scan_len0; m->block_sums = m->scan + scan_len0; m->total = m->block_sums + ((scan_len0 + kBlock * kScanItems - 1) / (kBlock * kScanItems) + 1);
    hipLaunchKernelGGL(k_reg_count, dim3(rw_grid(T)), dim3(kBlock), 0, ctx->stream, (const uint8_t*)regs->idx, regs->wide, T, m->flags);
    s = hipGetLastError() == hipSuccess ? rw_scan(m, T) : JOLT_ERR_HIP;
    uint32_t cap = 0;
    if (s == JOLT_OK) s = rw_read_total(m, &cap);
    if (s != JOLT_OK) { jolt_internal_dev_free(ctx, d_flags0); m->flags = m->scan = m->block_sums = m->total = nullptr; jolt_rw_matrix_destroy(m); return s; }
    m->cap = std::max<uint32_t>(cap, 1);
    const size_t scan_len = std::max<size_t>(T, m->cap);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    size_t o_st[2][6];
    for (int b = 0; b < 2; ++b) {
        o_st[b][0] = take((size_t)m->cap * 8); o_st[b][1] = take((size_t)m->cap * 8); o_st[b][2] = take((size_t)m->cap * 8);
        o_st[b][3] = take((size_t)m->cap * 32); o_st[b][4] = take((size_t)m->cap * 32); o_st[b][5] = take((size_t)m->cap * 32);
    }
    const size_t o_sib = take((size_t)m->cap * 4), o_match = take((size_t)m->cap * 4), o_flags = take(scan_len * 8), o_scan = take(scan_len * 8),
                 o_bs = take(((scan_len + kBlock * kScanItems - 1) / (kBlock * kScanItems) + 1) * 8), o_total = take(256);
    s = jolt_internal_dev_alloc(ctx, off, &m->block);
    if (s != JOLT_OK) { jolt_internal_dev_free(ctx, d_flags0); m->flags = m->scan = m->block_sums = m->total = nullptr; jolt_rw_matrix_destroy(m); return s; }
    char* base = (char*)m->block;
    for (int b = 0; b < 2; ++b) {
        m->st[b].key = (uint64_t*)(base + o_st[b][0]); m->st[b].prev_u = (uint64_t*)(base + o_st[b][1]); m->st[b].next_u = (uint64_t*)(base + o_st[b][2]);
        m->st[b].val = (Fr*)(base + o_st[b][3]); m->st[b].ra = (Fr*)(base + o_st[b][4]); m->st[b].wa = (Fr*)(base + o_st[b][5]);
        m->st[b].prev_f = nullptr; m->st[b].next_f = nullptr;
    }
    const uint64_t* pos = m->scan;  // the scan of the counts (still in the temporary block)
    hipLaunchKernelGGL(k_reg_build, dim3(rw_grid(T)), dim3(kBlock), 0, ctx->stream, (const uint8_t*)regs->idx, regs->wide, (const uint64_t*)rs1_val->data, (const uint64_t*)rs2_val->data,
                       (const uint64_t*)rd_pre->data, (const uint64_t*)rd_post->data, T, pos, m->gamma, mul(m->gamma, m->gamma), m->st[0]);
    hipError_t e = hipGetLastError();
    jolt_internal_dev_free(ctx, d_flags0);  // stream-ordered: the build kernel above is its last reader
    m->sib_lb = (uint32_t*)(base + o_sib); m->matched = (uint32_t*)(base + o_match); m->flags = (uint64_t*)(base + o_flags); m->scan = (uint64_t*)(base + o_scan);
    m->block_sums = (uint64_t*)(base + o_bs); m->total = (uint64_t*)(base + o_total);
    if (e != hipSuccess) { ctx->last_error = std::string("registers matrix: ") + hipGetErrorString(e); jolt_rw_matrix_destroy(m); return JOLT_ERR_HIP; }
    m->n = cap;
    *out = m;
    return JOLT_OK;
}

// ProveRounds::prove_round (registers_read_write/mod.rs:374-389), device half.  Cycle rounds: evals_out[0..1] = (q(0), leading coefficient) of the
// quadratic inner factor, aux_out = {current_scalar, r_cycle[current_index - 1], 0} for gruen_poly_deg_3; address rounds: evals_out[0..3] =
// s(0), s(1), s(2), s(3) (UnivariatePoly::from_evals; s(0) + s(1) is the claim), aux_out = 0.
extern "C" int32_t jolt_registers_rw_prove_round(jolt_rw_matrix* m, const jolt_fr_t* bind, jolt_fr_t* evals_out, jolt_fr_t* aux_out) {
    if (!m || !evals_out || !m->registers) return JOLT_ERR_INVALID_ARG;
    jolt_ctx* ctx = m->ctx;
    if (bind) {
        Fr r = fr_from_abi(bind);
        JOLT_REQUIRE(ctx, fr_is_canonical(r), "bind challenge is not a canonical Fr");
        if (m->round >= m->log_t + m->log_k) { ctx->last_error = "registers matrix already fully bound"; return JOLT_ERR_INVALID_ARG; }
        JOLT_TRY(rw_ingest(m, r));
    }
    if (m->round >= m->log_t + m->log_k) { ctx->last_error = "prove_round on a fully bound registers matrix"; return JOLT_ERR_INVALID_ARG; }
    const Fr zero = Fr::zero();
    for (int k = 0; k < 4; ++k) fr_to_abi(&evals_out[k], zero);
    if (aux_out) for (int k = 0; k < 3; ++k) fr_to_abi(&aux_out[k], zero);
    if (m->round < m->log_t) {
        const int grid = (int)std::max<uint32_t>(1, std::min<uint32_t>((m->n + kBlock - 1) / kBlock, (uint32_t)ctx->num_cus * rw_round_mult()));
        JOLT_TRY(jolt_internal_ensure_scratch(ctx, (size_t)grid * 2 + 8, 8));
        RwArrays& a = m->st[m->cur];
        if (m->n && !m->match_valid) {
            hipLaunchKernelGGL(k_rw_cycle_match, dim3(rw_grid(m->n)), dim3(kBlock), 0, ctx->stream, (const uint64_t*)a.key, m->n, m->sib_lb, m->matched, m->flags);
            JOLT_HIP_TRY(ctx, hipGetLastError());
            m->match_valid = true;
        }
        hipLaunchKernelGGL(k_rw_cycle_round<true>, dim3(grid), dim3(kBlock), 0, ctx->stream, a, m->n, (const uint32_t*)m->sib_lb, (const uint32_t*)m->matched, (const Fr*)m->inc->data(),
                           (const Fr*)m->e_out_cache[m->e_out_bits]->data(), (const Fr*)m->e_in_cache[m->e_in_bits]->data(), (int)m->e_in_bits, m->gamma, ctx->d_partials);
        JOLT_HIP_TRY(ctx, hipGetLastError());
        hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(kBlock), 0, ctx->stream, (const Fr*)ctx->d_partials, grid, 2, ctx->d_results);
        JOLT_HIP_TRY(ctx, hipGetLastError());
        JOLT_HIP_TRY(ctx, hipMemcpyAsync(ctx->h_results, ctx->d_results, 2 * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
        JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        std::memcpy(evals_out, ctx->h_results, 2 * sizeof(Fr));
        if (aux_out) {
            fr_to_abi(&aux_out[0], m->current_scalar);
            fr_to_abi(&aux_out[1], m->w[m->log_t - m->round - 1]);
        }
        return JOLT_OK;
    }
    // address_round_message (mod.rs:217-252): every (degree + 1) point sampled directly over the K-sized arrays
    Fr ev[4] = {zero, zero, zero, zero};
    const size_t half = m->reg_ra.size() / 2;
    for (size_t y = 0; y < half; ++y) {
        Fr ra_t = m->reg_ra[2 * y], wa_t = m->reg_wa[2 * y], val_t = m->reg_val[2 * y];
        const Fr ra_m = sub(m->reg_ra[2 * y + 1], ra_t), wa_m = sub(m->reg_wa[2 * y + 1], wa_t), val_m = sub(m->reg_val[2 * y + 1], val_t);
        for (int t = 0; t < 4; ++t) {
            ev[t] = add(ev[t], add(mul(wa_t, add(m->inc_scalar, val_t)), mul(ra_t, val_t)));
            ra_t = add(ra_t, ra_m); wa_t = add(wa_t, wa_m); val_t = add(val_t, val_m);
        }
    }
    for (int t = 0; t < 4; ++t) fr_to_abi(&evals_out[t], mul(m->current_scalar, ev[t]));
    return JOLT_OK;
}

// RegistersReadWriteOutputClaims without the operand claims (mod.rs:386-402): {registers_val, rd_wa, gamma * rs1_ra + gamma^2 * rs2_ra, rd_inc, bound eq};
// rs1_ra / rs2_ra are one-hot evaluations of the index columns at the bound point (jolt_onehot_materialize + jolt_evaluate)
extern "C" int32_t jolt_registers_rw_final_values(jolt_rw_matrix* m, jolt_fr_t* out) {
    if (!m || !out || !m->registers) return JOLT_ERR_INVALID_ARG;
    if (m->round != m->log_t + m->log_k) return JOLT_ERR_NOT_FULLY_BOUND;
    if (m->reg_val.size() != 1) return JOLT_ERR_NOT_FULLY_BOUND;
    fr_to_abi(&out[0], m->reg_val[0]);
    fr_to_abi(&out[1], m->reg_wa[0]);
    fr_to_abi(&out[2], m->reg_ra[0]);
    fr_to_abi(&out[3], m->inc_scalar);
    fr_to_abi(&out[4], m->current_scalar);
    return JOLT_OK;
}

// test hook: the current cells of the cycle phase (rows, cols, raw checkpoints as u64; val, ra, wa as Fr)
extern "C" int32_t jolt_registers_rw_download(jolt_rw_matrix* m, uint64_t* rows, uint64_t* cols, jolt_fr_t* val, jolt_fr_t* ra, jolt_fr_t* wa, uint64_t* prev, uint64_t* next) {
    if (!m || !m->registers || !rows || !cols || !val || !ra || !wa || !prev || !next) return JOLT_ERR_INVALID_ARG;
    jolt_ctx* ctx = m->ctx;
    const uint32_t n = m->n;
    if (!n || m->round >= m->log_t) return JOLT_OK;
    const RwArrays& a = m->st[m->cur];
    std::vector<uint64_t> key(n);
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(key.data(), a.key, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(val, a.val, (size_t)n * 32, hipMemcpyDeviceToHost, ctx->stream));
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(ra, a.ra, (size_t)n * 32, hipMemcpyDeviceToHost, ctx->stream));
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(wa, a.wa, (size_t)n * 32, hipMemcpyDeviceToHost, ctx->stream));
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(prev, a.prev_u, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(next, a.next_u, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t i = 0; i < n; ++i) { rows[i] = key[i] >> 32; cols[i] = (uint32_t)key[i]; }
    return JOLT_OK;
}
