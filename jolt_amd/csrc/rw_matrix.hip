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
};

__device__ __forceinline__ uint32_t lower_bound_u64(const uint64_t* __restrict__ a, uint32_t n, uint64_t x) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
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
    const uint32_t lb = lower_bound_u64(key, n, want);
    const bool m = lb < n && key[lb] == want;
    const bool even = ((k >> 32) & 1) == 0;
    sib_lb[i] = lb;
    matched[i] = m ? 1u : 0u;
    flags[i] = ((even || !m) ? 1ull : 0ull) | ((even && m) ? (1ull << 32) : 0ull);
}

// CycleMajorMatrix::quadratic_coefficients (rw_matrix.rs:287-325) with CycleMajorEntry::quadratic_evals (:116-145) per pair:
// partial sums of head(pair) * [q(0), q_inf]; a matched pair is evaluated by its EVEN entry
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
        if (even) {
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
__global__ __launch_bounds__(kBlock) void k_rw_cycle_bind(RwArrays a, uint32_t n, const uint32_t* __restrict__ sib_lb, const uint32_t* __restrict__ matched,
                                                          const uint64_t* __restrict__ scan, Fr r, int shifted, RwArrays o) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = a.key[i];
    const uint32_t row = (uint32_t)(k >> 32), col = (uint32_t)k;
    const bool even = (row & 1) == 0, m = matched[i] != 0;
    if (!even && m) return;
    const uint64_t gkey = (uint64_t)(row & ~1u) << 32;
    const uint32_t gs = lower_bound_u64(a.key, n, gkey), os = lower_bound_u64(a.key, n, gkey | (1ull << 32));
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
    size_t o_st[2][7];
    for (int b = 0; b < 2; ++b) {
        o_st[b][0] = take((size_t)m->cap * 8); o_st[b][1] = take((size_t)m->cap * 8); o_st[b][2] = take((size_t)m->cap * 8);
        o_st[b][3] = take((size_t)m->cap * 32); o_st[b][4] = take((size_t)m->cap * 32); o_st[b][5] = take((size_t)m->cap * 32); o_st[b][6] = take((size_t)m->cap * 32);
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
__global__ __launch_bounds__(kBlock) void k_rw_count_accesses(const uint64_t* __restrict__ addresses, uint32_t cycles, uint64_t K, uint32_t* __restrict__ counters) {
    const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    const uint64_t a = j < cycles ? addresses[j] : kNoAccess;
    const uint64_t hit = __ballot(a != kNoAccess), bad = __ballot(a != kNoAccess && a >= K);
    if ((threadIdx.x & 63) == 0) {
        if (hit) atomicAdd(&counters[0], (uint32_t)__popcll(hit));
        if (bad) atomicAdd(&counters[1], (uint32_t)__popcll(bad));
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
        hipLaunchKernelGGL(k_rw_count_accesses, dim3(rw_grid((uint32_t)cycles)), dim3(kBlock), 0, ctx->stream, (const uint64_t*)addresses->data, (uint32_t)cycles,
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
            hipLaunchKernelGGL(k_rw_cycle_bind, dim3(rw_grid(m->n)), dim3(kBlock), 0, ctx->stream, a, m->n, (const uint32_t*)m->sib_lb, (const uint32_t*)m->matched,
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
        if (m->round == m->log_t - 1 && m->n) {
            hipLaunchKernelGGL(k_rw_to_address_major, dim3(rw_grid(m->n)), dim3(kBlock), 0, ctx->stream, m->st[m->cur], m->n);
            JOLT_HIP_TRY(ctx, hipGetLastError());
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
    if (bind) {
        Fr r = fr_from_abi(bind);
        JOLT_REQUIRE(ctx, fr_is_canonical(r), "bind challenge is not a canonical Fr");
        if (m->round >= m->log_t + m->log_k) { ctx->last_error = "rw matrix already fully bound"; return JOLT_ERR_INVALID_ARG; }
        JOLT_TRY(rw_ingest(m, r));
    }
    if (m->round >= m->log_t + m->log_k) { ctx->last_error = "prove_round on a fully bound rw matrix"; return JOLT_ERR_INVALID_ARG; }
    const int grid = (int)std::max<uint32_t>(1, std::min<uint32_t>((m->n + kBlock - 1) / kBlock, (uint32_t)ctx->num_cus * 4));
    JOLT_TRY(jolt_internal_ensure_scratch(ctx, (size_t)grid * 2 + 8, 8));
    RwArrays& a = m->st[m->cur];
    Fr zero = Fr::zero();
    if (m->round < m->log_t) {
        if (m->n && !m->match_valid) {
            hipLaunchKernelGGL(k_rw_cycle_match, dim3(rw_grid(m->n)), dim3(kBlock), 0, ctx->stream, (const uint64_t*)a.key, m->n, m->sib_lb, m->matched, m->flags);
            JOLT_HIP_TRY(ctx, hipGetLastError());
            m->match_valid = true;
        }
        hipLaunchKernelGGL(k_rw_cycle_round, dim3(grid), dim3(kBlock), 0, ctx->stream, a, m->n, (const uint32_t*)m->sib_lb, (const uint32_t*)m->matched,
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

// test hook: the current entries (rows, cols as u64; val, ra, prev, next as Fr -- raw checkpoints promoted in the cycle phase)
extern "C" int32_t jolt_rw_matrix_download(jolt_rw_matrix* m, uint64_t* rows, uint64_t* cols, jolt_fr_t* val, jolt_fr_t* ra, jolt_fr_t* prev, jolt_fr_t* next) {
    if (!m || !rows || !cols || !val || !ra || !prev || !next) return JOLT_ERR_INVALID_ARG;
    jolt_ctx* ctx = m->ctx;
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
