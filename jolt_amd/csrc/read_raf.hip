// jolt_amd/csrc/read_raf.hip -- the T-scale scans of instruction read+RAF checking (stage 5) on the device.
//
// OptimizedInstructionReadRafKernel (crates/jolt-kernels/src/optimized/instruction_read_raf.rs) runs 16 prefix-suffix phases of 8
// address variables and then log T cycle rounds.  What touches every cycle:
//   * init_phase (:747-900): condensation u[j] *= v_prev[chunk_prev(j)] (:750-758); the fused RAF scan -- per row, by raf_flag, u or
//     u x (left / right operand, identity) added to one of 256 chunk bins (:770-812); init_suffix_tables (:901-971) -- for every lookup
//     table present and each of its suffixes, sum_j u[j] * suffix_mle(low bits of j) binned by chunk;
//   * init_cycle_rounds / pending_*_base (:1140-1232): the combined value column and the ra_i columns (products of the bound-challenge
//     eq tables at the row's chunks) that the cycle rounds sum with eq(r_reduction, .).
// The 8 address rounds of a phase work on 256-entry polynomials (prefix tables from the checkpoints, address_message :973-1050) and
// stay with the reference's host code, as do the checkpoints; the cycle rounds are the eq-weighted product member of capi.hip
// (jolt_member_create_split_eq_lc) over the columns built here; the output flag claims are jolt_onehot_pushforward over the table-index column.
//
// On the device a phase is: key = (table, chunk) per row -> counting sort through LDS (the MSM's kernels) -> ONE WAVEFRONT per
// (table, chunk) bin walks its rows once per accumulator set (RAF sums, then each suffix of the table), products u x small value in
// the small-scalar accumulator (small_scalar.hip.h: mul_u64 / mul_u128 of the reference's scan), lanes folded by shuffles.  No atomics
// on field elements, no per-suffix launch.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ctx.hpp"
#include "msm_kernels.hip.h"
#include "poly_kernels.hip.h"
#include "small_scalar.hip.h"
#include "suffix_mle.hip.h"

using namespace jolt;
using namespace jolt::msmk;

int32_t jolt_internal_table_new(jolt_ctx* ctx, size_t len, jolt_table** out);

constexpr uint32_t kRafChunk = 256;       // CHUNK_SIZE (:72-73)
constexpr uint32_t kRafChunkBits = 8;
constexpr uint32_t kRafSums = 6;          // left, right, identity, shift_half, shift_full, upper_all_ones
constexpr uint32_t kRafMaxTables = 126;   // table indices must fit the packed claim byte (:1166-1171)
constexpr uint8_t kRafNoTable = 0xFF;

struct jolt_read_raf {
    jolt_ctx* ctx = nullptr;
    size_t cycles = 0;
    uint32_t n_tables = 0;
    uint64_t* index = nullptr;  // [2 * cycles]: lookup_index as (lo, hi)
    uint8_t* table = nullptr;   // table index or 0xFF
    uint8_t* raf = nullptr;     // raf_flag
    // A phase's ORDER: its rows sorted by (table, address chunk), the bins' work items, and the lookup index / flag of sorted[p] at position p.  It depends on the
    // witness and the phase alone -- not on any challenge -- so the order of phase p + 1 is built (into the other set) right behind phase p's scan, under the eight
    // address rounds the host runs before it can call again (jolt_read_raf_phase_scan).
    struct Order {
        uint32_t *keys = nullptr, *sorted = nullptr, *hist = nullptr, *offs = nullptr, *cursor = nullptr, *seg_start = nullptr;
        uint64_t* index_sorted = nullptr;
        uint8_t* raf_sorted = nullptr;
        int64_t suffix_len = -1;  // the phase this set holds the order of; -1: none
    } order[2];
    int cur = 0;
    Fr* u_sorted = nullptr;  // u in the current order (k_rr_gather_u): depends on the challenges, gathered inside the scan
    void* h_out = nullptr;   // page-locked read-back block of a scan's sums (the D2H copy then needs no staging by the runtime)
    size_t h_out_cap = 0;
    hipEvent_t ev_out = nullptr;
    bool lds_attr_set = false;
    Fr *bin_raf = nullptr, *d_suffix = nullptr, *d_raf = nullptr;
    uint32_t* d_cfg = nullptr;
    std::vector<uint32_t> cfg_host;  // what d_cfg holds
    size_t suffix_cap = 0, cfg_cap = 0;
    Fr* part = nullptr;             // per work item: the 6 RAF sums, then one sum per suffix of the bin's table
    size_t part_cap = 0;
};

namespace {

__device__ __forceinline__ uint32_t chunk_of(uint64_t lo, uint64_t hi, uint32_t shift) {
    const uint64_t v = shift >= 64 ? hi >> (shift - 64) : (shift == 0 ? lo : (lo >> shift) | (hi << (64 - shift)));
    return (uint32_t)v & (kRafChunk - 1);
}
__device__ __forceinline__ void mask_low(uint64_t& lo, uint64_t& hi, uint32_t len) {  // bits mod 2^len (LookupBits::new)
    if (len >= 128) return;
    if (len >= 64) hi &= len == 64 ? 0ull : ((1ull << (len - 64)) - 1);
    else { hi = 0; lo &= len == 0 ? 0ull : ((1ull << len) - 1); }
}

// keys[j] = bucket * 256 + chunk + 1 (bucket = table index, n_tables for rows without a lookup table)
__global__ __launch_bounds__(kBlock) void k_rr_keys(const uint64_t* __restrict__ index, const uint8_t* __restrict__ table, size_t cycles, uint32_t n_tables,
                                                    uint32_t suffix_len, uint32_t* __restrict__ keys) {
    const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= cycles) return;
    const uint8_t t = table[j];
    const uint32_t bucket = t == kRafNoTable ? n_tables : (uint32_t)t;
    keys[j] = bucket * kRafChunk + chunk_of(index[2 * j], index[2 * j + 1], suffix_len) + 1;
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
// (sum a z) held as an unreduced integer -> its Montgomery field element
__device__ __forceinline__ Fr small_value(const SmallAcc& acc) { return mul(small_redc<FrParams>(acc), Fr::r2()); }

// Work items.  Lookup indices are skewed (small operands, zero, all-ones: a third of a real trace's rows can share one chunk value), so a
// bin is cut into items of kRafSegRows rows, one wavefront each; partial sums are folded per bin afterwards.  (One wavefront per BIN took as
// long as the fullest bin: 9 ms per phase on a trace with 12 % zero indices where uniform indices take 1.3 ms.)
constexpr uint32_t kRafSegRowsDefault = 1024;
// rows per work item (a power of two, 64 .. 4096): JOLT_RAF_SEG_ROWS, for the A/B of shorter per-lane chains against more partial sums to fold
static uint32_t raf_seg_rows() {
    static const uint32_t v = [] {
        const char* e = std::getenv("JOLT_RAF_SEG_ROWS");
        uint32_t r = e ? (uint32_t)std::atoi(e) : kRafSegRowsDefault;
        if (r < 64 || r > 4096 || (r & (r - 1))) r = kRafSegRowsDefault;
        return r;
    }();
    return v;
}
__global__ __launch_bounds__(1024) void k_rr_segments(const uint32_t* __restrict__ hist, uint32_t n_bins, uint32_t kRafSegRows, uint32_t* __restrict__ seg_start) {
    __shared__ uint32_t sm[1024];
    const uint32_t per = (n_bins + 1023) / 1024, lo = min(threadIdx.x * per, n_bins), hi = min(lo + per, n_bins);
    uint32_t local = 0;
    for (uint32_t b = lo; b < hi; ++b) local += (hist[b + 1] + kRafSegRows - 1) / kRafSegRows;
    sm[threadIdx.x] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t v = (int)threadIdx.x >= off ? sm[threadIdx.x - off] : 0;
        __syncthreads();
        sm[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = sm[threadIdx.x] - local;
    for (uint32_t b = lo; b < hi; ++b) {
        seg_start[b] = run;
        run += (hist[b + 1] + kRafSegRows - 1) / kRafSegRows;
    }
    if (threadIdx.x == 1023) seg_start[n_bins] = sm[1023];
}

// The rows of a phase in bin order: position p holds u, the lookup index and the flag of row sorted[p].  One thread per row, so the three gathers of a row
// (the 32-byte u[j] above all) are hidden by occupancy; the scans below then read their rows contiguously.  (Scanning through sorted[] directly left every
// lane waiting on its own chain row id -> u[j] / index[j], 16 rows deep: 0.87 ms per phase against 0.1 + 0.3 ms.)
// Two kernels: the lookup index and the flag belong to the ORDER (witness only: k_rr_gather_rows runs with the sort, ahead of the phase), u to the scan.
__global__ __launch_bounds__(kBlock) void k_rr_gather_rows(const uint32_t* __restrict__ sorted, size_t rows, const uint64_t* __restrict__ index, const uint8_t* __restrict__ raf,
                                                           uint64_t* __restrict__ index_sorted, uint8_t* __restrict__ raf_sorted) {
    const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= rows) return;
    const uint32_t j = sorted[p];
    const uint4 w = *reinterpret_cast<const uint4*>(index + 2 * (size_t)j);
    *reinterpret_cast<uint4*>(index_sorted + 2 * p) = w;
    raf_sorted[p] = raf[j];
}
__global__ __launch_bounds__(kBlock) void k_rr_gather_u(const uint32_t* __restrict__ sorted, size_t rows, const Fr* __restrict__ u, Fr* __restrict__ u_sorted) {
    const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= rows) return;
    st_fr(u_sorted + p, ld_fr(u + sorted[p]));
}

// cfg: [0 .. n_tables] suffix offsets, then the suffix kinds (one u32 each)
template <bool GATHERED>
__global__ __launch_bounds__(kBlock) void k_rr_accumulate(const uint64_t* __restrict__ index /* in bin order */, const uint8_t* __restrict__ raf /* in bin order */,
                                                          const Fr* __restrict__ u /* in bin order */, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offs,
                                                          uint32_t n_tables, uint32_t suffix_len, uint32_t upper_suffix_bits, int canonical,
                                                          const uint32_t* __restrict__ cfg, const uint32_t* __restrict__ seg_start, uint32_t slots, uint32_t kRafSegRows, Fr* __restrict__ part) {
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, n_waves = gridDim.x * (kBlock / 64);
    const uint32_t n_bins = (n_tables + 1) * kRafChunk, n_items = seg_start[n_bins];
    for (uint32_t item = wave; item < n_items; item += n_waves) {
        uint32_t b_lo = 0, b_hi = n_bins;  // the bin with seg_start[bin] <= item < seg_start[bin + 1] (empty bins own no item)
        while (b_hi - b_lo > 1) {
            const uint32_t mid = (b_lo + b_hi) >> 1;
            if (seg_start[mid] <= item) b_lo = mid; else b_hi = mid;
        }
        const uint32_t bin = b_lo, bucket = bin / kRafChunk;
        const uint32_t first = (item - seg_start[bin]) * kRafSegRows;
        const uint32_t cnt = min(hist[bin + 1] - first, kRafSegRows), start = offs[bin + 1] + first;
        Fr* out = part + (size_t)item * slots;
        // ---- RAF scan, operand rows (raf_flag = 0): shift_half, left, right (:785-796)
        {
            Fr shift_half = Fr::zero();
            SmallAcc left = small_zero(), right = small_zero();
            for (uint32_t k = lane; k < cnt; k += 64) {
                const uint32_t j = GATHERED ? start + k : sorted[start + k];
                if (raf[j]) continue;
                const Fr uj = ld_fr(u + j);
                uint64_t lo = index[2 * (size_t)j], hi = index[2 * (size_t)j + 1];
                mask_low(lo, hi, suffix_len);
                const Operands o = uninterleave(lo, hi, suffix_len);
                shift_half = add(shift_half, uj);
                const uint32_t l[4] = {(uint32_t)o.x, (uint32_t)(o.x >> 32), 0u, 0u}, r[4] = {(uint32_t)o.y, (uint32_t)(o.y >> 32), 0u, 0u};
                small_fmadd<2>(left, uj, l);
                small_fmadd<2>(right, uj, r);
            }
            const Fr s0 = wave_sum_fr(small_value(left)), s1 = wave_sum_fr(small_value(right)), s3 = wave_sum_fr(shift_half);
            if (lane == 0) {
                st_fr(out + 0, s0);
                st_fr(out + 1, s1);
                st_fr(out + 3, s3);
            }
        }
        // ---- RAF scan, identity rows (raf_flag = 1): shift_full, identity, upper_all_ones (:776-784, :797-802)
        {
            Fr shift_full = Fr::zero(), upper = Fr::zero();
            SmallAcc identity = small_zero();
            for (uint32_t k = lane; k < cnt; k += 64) {
                const uint32_t j = GATHERED ? start + k : sorted[start + k];
                if (!raf[j]) continue;
                const Fr uj = ld_fr(u + j);
                uint64_t lo = index[2 * (size_t)j], hi = index[2 * (size_t)j + 1];
                mask_low(lo, hi, suffix_len);
                shift_full = add(shift_full, uj);
                const uint32_t m[4] = {(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
                small_fmadd<4>(identity, uj, m);
                if (canonical) {
                    bool all_ones = upper_suffix_bits == 0;
                    if (!all_ones) {  // (suffix_bits >> (suffix_len - upper)) == 2^upper - 1
                        const uint32_t sh = suffix_len - upper_suffix_bits;
                        uint64_t tl = sh >= 64 ? hi >> (sh - 64) : (sh == 0 ? lo : (lo >> sh) | (hi << (64 - sh))), th = sh >= 64 ? 0 : (sh == 0 ? hi : hi >> sh);
                        uint64_t wl = ~0ull, wh = ~0ull;
                        mask_low(wl, wh, upper_suffix_bits);
                        all_ones = tl == wl && th == wh;
                    }
                    if (all_ones) upper = add(upper, uj);
                }
            }
            const Fr s2 = wave_sum_fr(small_value(identity)), s4 = wave_sum_fr(shift_full), s5 = wave_sum_fr(upper);
            if (lane == 0) {
                st_fr(out + 2, s2);
                st_fr(out + 4, s4);
                st_fr(out + 5, s5);
            }
        }
        // ---- suffix accumulators of this bin's table (:901-971)
        if (bucket < n_tables) {
            const uint32_t s_lo = cfg[bucket], s_hi = cfg[bucket + 1];
            for (uint32_t s = s_lo; s < s_hi; ++s) {
                const uint32_t kind = cfg[n_tables + 1 + s];
                SmallAcc acc = small_zero();
                for (uint32_t k = lane; k < cnt; k += 64) {
                    const uint32_t j = GATHERED ? start + k : sorted[start + k];
                    uint64_t lo = index[2 * (size_t)j], hi = index[2 * (size_t)j + 1];
                    mask_low(lo, hi, suffix_len);
                    const uint64_t value = suffix_mle(kind, lo, hi, suffix_len);
                    if (value == 0) continue;
                    const uint32_t m[4] = {(uint32_t)value, (uint32_t)(value >> 32), 0u, 0u};
                    small_fmadd<2>(acc, ld_fr(u + j), m);
                }
                const Fr total = wave_sum_fr(small_value(acc));
                if (lane == 0) st_fr(out + kRafSums + (s - s_lo), total);
            }
        }
    }
}
// per bin: the sums of its items -> bin_raf[bin][6] and suffix_out[(first suffix of the bin's table + s) * 256 + chunk]
__global__ __launch_bounds__(kBlock) void k_rr_fold_items(const Fr* __restrict__ part, const uint32_t* __restrict__ seg_start, uint32_t n_tables, uint32_t slots,
                                                          const uint32_t* __restrict__ cfg, Fr* __restrict__ bin_raf, Fr* __restrict__ suffix_out) {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x, n_bins = (n_tables + 1) * kRafChunk;
    const uint32_t bin = t / slots, q = t % slots;
    if (bin >= n_bins) return;
    const uint32_t bucket = bin / kRafChunk, chunk = bin % kRafChunk;
    Fr acc = Fr::zero();
    for (uint32_t item = seg_start[bin]; item < seg_start[bin + 1]; ++item) acc = add(acc, ld_fr(part + (size_t)item * slots + q));
    if (q < kRafSums) { st_fr(bin_raf + (size_t)bin * kRafSums + q, acc); return; }
    if (bucket >= n_tables) return;
    const uint32_t s = cfg[bucket] + (q - kRafSums);
    if (s < cfg[bucket + 1]) st_fr(suffix_out + (size_t)s * kRafChunk + chunk, acc);
}
// raf_out[q * 256 + chunk] = sum over the buckets of bin_raf[(bucket * 256 + chunk) * 6 + q]
__global__ __launch_bounds__(kBlock) void k_rr_fold_raf(const Fr* __restrict__ bin_raf, uint32_t n_buckets, Fr* __restrict__ raf_out) {
    const uint32_t chunk = threadIdx.x, q = blockIdx.x;
    Fr s = Fr::zero();
    for (uint32_t b = 0; b < n_buckets; ++b) s = add(s, ld_fr(bin_raf + ((size_t)b * kRafChunk + chunk) * kRafSums + q));
    st_fr(raf_out + (size_t)q * kRafChunk + chunk, s);
}

// u[j] *= v[(lookup_index[j] >> shift) & 255]
__global__ __launch_bounds__(kBlock) void k_rr_condense(const uint64_t* __restrict__ index, size_t cycles, const Fr* __restrict__ v, uint32_t shift, Fr* __restrict__ u) {
    __shared__ Fr sv[kRafChunk];
    for (uint32_t k = threadIdx.x; k < kRafChunk; k += kBlock) sv[k] = ld_fr(v + k);
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x; j < cycles; j += stride)
        st_fr(u + j, mul(ld_fr(u + j), sv[chunk_of(index[2 * j], index[2 * j + 1], shift)]));
}

struct CycleOut {
    Fr* ra[16];
};
// combined[j] = table_values[table(j)] + (raf_flag ? raf_identity : raf_interleaved); ra_i[j] = prod over the phases of group i of
// v_tables[phase][chunk_phase(j)] (pending_combined_base / pending_ra_base :1203-1232)
__global__ __launch_bounds__(kBlock) void k_rr_cycle_tables(const uint64_t* __restrict__ index, const uint8_t* __restrict__ table, const uint8_t* __restrict__ raf,
                                                            size_t cycles, const Fr* __restrict__ table_values, Fr raf_interleaved, Fr raf_identity,
                                                            const Fr* __restrict__ v_tables, uint32_t address_bits, uint32_t ra_count, uint32_t phases_per_ra,
                                                            Fr* __restrict__ combined, CycleOut out) {
    const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= cycles) return;
    const uint8_t t = table[j];
    Fr c = raf[j] ? raf_identity : raf_interleaved;
    if (t != kRafNoTable) c = add(c, ld_fr(table_values + t));
    st_fr(combined + j, c);
    const uint64_t lo = index[2 * j], hi = index[2 * j + 1];
    for (uint32_t i = 0; i < ra_count; ++i) {
        uint32_t phase = i * phases_per_ra;
        uint32_t shift = address_bits - (phase + 1) * kRafChunkBits;
        Fr product = ld_fr(v_tables + (size_t)phase * kRafChunk + chunk_of(lo, hi, shift));
        for (uint32_t p = 1; p < phases_per_ra; ++p) {
            ++phase;
            shift -= kRafChunkBits;
            product = mul(product, ld_fr(v_tables + (size_t)phase * kRafChunk + chunk_of(lo, hi, shift)));
        }
        st_fr(out.ra[i] + j, product);
    }
}

}  // namespace

extern "C" int32_t jolt_read_raf_destroy(jolt_ctx* ctx, jolt_read_raf* rr) {
    if (!rr) return JOLT_OK;
    jolt_ctx* c = ctx ? ctx : rr->ctx;
    if (c) (void)hipStreamSynchronize(c->stream);
    void* ptrs[] = {rr->index, rr->table, rr->raf, rr->bin_raf, rr->d_suffix, rr->d_raf, rr->d_cfg, rr->part, rr->u_sorted};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (jolt_read_raf::Order& o : rr->order)
        for (void* p : {(void*)o.keys, (void*)o.sorted, (void*)o.hist, (void*)o.offs, (void*)o.cursor, (void*)o.seg_start, (void*)o.index_sorted, (void*)o.raf_sorted})
            if (p) (void)hipFree(p);
    if (rr->h_out) (void)hipHostFree(rr->h_out);
    if (rr->ev_out) (void)hipEventDestroy(rr->ev_out);
    delete rr;
    return JOLT_OK;
}

extern "C" int32_t jolt_read_raf_create(jolt_ctx* ctx, const uint64_t* lookup_index, const uint8_t* table_index, const uint8_t* raf_flag, size_t cycles, uint32_t n_tables,
                                        jolt_read_raf** out) {
    if (!ctx || !lookup_index || !table_index || !raf_flag || !out || cycles == 0) return JOLT_ERR_INVALID_ARG;
    if (n_tables == 0 || n_tables > kRafMaxTables || cycles >= ((size_t)1 << 31)) return JOLT_ERR_UNSUPPORTED;
    for (size_t j = 0; j < cycles; ++j)
        JOLT_REQUIRE(ctx, table_index[j] == kRafNoTable || table_index[j] < n_tables, "lookup table index out of range");
    jolt_read_raf* rr = new (std::nothrow) jolt_read_raf();
    if (!rr) return JOLT_ERR_OOM;
    rr->ctx = ctx;
    rr->cycles = cycles;
    rr->n_tables = n_tables;
    const size_t n_bins = (size_t)(n_tables + 1) * kRafChunk + 1;
    hipError_t e = hipMalloc((void**)&rr->index, cycles * 16);
    if (e == hipSuccess) e = hipMalloc((void**)&rr->table, cycles);
    if (e == hipSuccess) e = hipMalloc((void**)&rr->raf, cycles);
    if (e == hipSuccess) e = hipMalloc((void**)&rr->u_sorted, cycles * sizeof(Fr));
    for (jolt_read_raf::Order& o : rr->order) {
        if (e == hipSuccess) e = hipMalloc((void**)&o.keys, cycles * 4);
        if (e == hipSuccess) e = hipMalloc((void**)&o.index_sorted, cycles * 16);
        if (e == hipSuccess) e = hipMalloc((void**)&o.raf_sorted, cycles);
        if (e == hipSuccess) e = hipMalloc((void**)&o.sorted, cycles * 4);
        if (e == hipSuccess) e = hipMalloc((void**)&o.hist, n_bins * 4);
        if (e == hipSuccess) e = hipMalloc((void**)&o.offs, n_bins * 4);
        if (e == hipSuccess) e = hipMalloc((void**)&o.cursor, n_bins * 4);
        if (e == hipSuccess) e = hipMalloc((void**)&o.seg_start, (n_bins + 1) * 4);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&rr->ev_out, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void**)&rr->bin_raf, n_bins * kRafSums * sizeof(Fr));
    if (e == hipSuccess) e = hipMalloc((void**)&rr->d_raf, kRafSums * kRafChunk * sizeof(Fr));
    if (e == hipSuccess) e = hipMemcpyAsync(rr->index, lookup_index, cycles * 16, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(rr->table, table_index, cycles, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(rr->raf, raf_flag, cycles, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ctx->last_error = std::string("read-raf rows: ") + hipGetErrorString(e);
        (void)jolt_read_raf_destroy(ctx, rr);
        return e == hipErrorOutOfMemory ? JOLT_ERR_OOM : JOLT_ERR_HIP;
    }
    *out = rr;
    return JOLT_OK;
}

extern "C" int32_t jolt_read_raf_phase_scan(jolt_ctx* ctx, jolt_read_raf* rr, const jolt_table* u, uint32_t suffix_len, uint32_t address_bits, int32_t canonical,
                                            const uint32_t* suffix_offsets, const uint8_t* suffix_kinds, jolt_fr_t* raf_out, jolt_fr_t* suffix_out) {
    if (!ctx || !rr || !u || !suffix_offsets || !raf_out) return JOLT_ERR_INVALID_ARG;
    if (u->len != rr->cycles) return JOLT_ERR_SIZE_MISMATCH;
    if (address_bits > 128 || address_bits % kRafChunkBits || suffix_len + kRafChunkBits > address_bits) return JOLT_ERR_INVALID_ARG;
    const uint32_t n_tables = rr->n_tables, total_suffixes = suffix_offsets[n_tables];
    if (suffix_offsets[0] != 0 || (total_suffixes && (!suffix_kinds || !suffix_out))) return JOLT_ERR_INVALID_ARG;
    std::vector<uint32_t> cfg(n_tables + 1 + total_suffixes);
    for (uint32_t t = 0; t <= n_tables; ++t) {
        if (t && suffix_offsets[t] < suffix_offsets[t - 1]) return JOLT_ERR_INVALID_ARG;
        cfg[t] = suffix_offsets[t];
    }
    for (uint32_t s = 0; s < total_suffixes; ++s) {
        if (suffix_kinds[s] >= kNumSuffixKinds) return JOLT_ERR_UNSUPPORTED;
        cfg[n_tables + 1 + s] = suffix_kinds[s];
    }
    if (cfg.size() > rr->cfg_cap) {
        if (rr->d_cfg) { JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); JOLT_HIP_TRY(ctx, hipFree(rr->d_cfg)); rr->d_cfg = nullptr; rr->cfg_host.clear(); }
        JOLT_HIP_TRY(ctx, hipMalloc((void**)&rr->d_cfg, cfg.size() * 4));
        rr->cfg_cap = cfg.size();
    }
    if ((size_t)total_suffixes * kRafChunk > rr->suffix_cap) {
        if (rr->d_suffix) { JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); JOLT_HIP_TRY(ctx, hipFree(rr->d_suffix)); rr->d_suffix = nullptr; }
        JOLT_HIP_TRY(ctx, hipMalloc((void**)&rr->d_suffix, (size_t)total_suffixes * kRafChunk * sizeof(Fr)));
        rr->suffix_cap = (size_t)total_suffixes * kRafChunk;
    }
    uint32_t max_suffixes = 0;
    for (uint32_t t = 0; t < n_tables; ++t) max_suffixes = std::max(max_suffixes, suffix_offsets[t + 1] - suffix_offsets[t]);
    const uint32_t slots = kRafSums + max_suffixes;
    const uint32_t kRafSegRows = raf_seg_rows();
    const size_t max_items = rr->cycles / kRafSegRows + (size_t)(n_tables + 1) * kRafChunk + 1;
    if (max_items * slots > rr->part_cap) {
        if (rr->part) { JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); JOLT_HIP_TRY(ctx, hipFree(rr->part)); rr->part = nullptr; }
        JOLT_HIP_TRY(ctx, hipMalloc((void**)&rr->part, max_items * slots * sizeof(Fr)));
        rr->part_cap = max_items * slots;
    }
    hipStream_t st = ctx->stream;
    if (cfg != rr->cfg_host) {  // every phase of a proof passes the same suffix lists: uploaded (and waited for: cfg is a local) only when they change
        JOLT_HIP_TRY(ctx, hipMemcpyAsync(rr->d_cfg, cfg.data(), cfg.size() * 4, hipMemcpyHostToDevice, st));
        JOLT_HIP_TRY(ctx, hipStreamSynchronize(st));
        rr->cfg_host = cfg;
    }
    const uint32_t B = (n_tables + 1) * kRafChunk;  // keys 1 .. B
    const size_t T = rr->cycles;
    const size_t lds = ((size_t)B + 1) * 4;
    if (lds > ctx->max_lds_per_block) return JOLT_ERR_UNSUPPORTED;
    static const bool gathered = !(std::getenv("JOLT_RR_GATHER") && std::atoi(std::getenv("JOLT_RR_GATHER")) == 0);
    static const bool ahead = !(std::getenv("JOLT_RR_AHEAD") && std::atoi(std::getenv("JOLT_RR_AHEAD")) == 0);
    // the order of phase `len` into set `o`: keys, counting sort by (table, chunk), the bins' work items, lookup index and flag in that order
    auto build_order = [&](jolt_read_raf::Order& o, uint32_t len) -> int32_t {
        o.suffix_len = -1;
        JOLT_HIP_TRY(ctx, hipMemsetAsync(o.hist, 0, ((size_t)B + 1) * 4, st));
        hipLaunchKernelGGL(k_rr_keys, dim3((unsigned)((T + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, (const uint64_t*)rr->index, (const uint8_t*)rr->table, T, n_tables, len, o.keys);
        const unsigned slices = (unsigned)std::max<size_t>(1, std::min<size_t>((size_t)ctx->num_cus, T / 16384 + 1));
        hipLaunchKernelGGL(k_msm_hist_lds, dim3(slices, 1), dim3(kSortBlock), lds, st, (const uint32_t*)o.keys, T, B, o.hist);
        hipLaunchKernelGGL(k_msm_scan, dim3(1), dim3(kBlock), 0, st, (const uint32_t*)o.hist, o.offs, o.cursor, B, 0x7FFFFFFFu, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u);
        hipLaunchKernelGGL(k_msm_scatter_lds, dim3(slices, 1), dim3(kSortBlock), lds, st, (const uint32_t*)o.keys, T, B, o.cursor, o.sorted);
        hipLaunchKernelGGL(k_rr_segments, dim3(1), dim3(1024), 0, st, (const uint32_t*)o.hist, B, raf_seg_rows(), o.seg_start);
        if (gathered)
            hipLaunchKernelGGL(k_rr_gather_rows, dim3((unsigned)((T + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, (const uint32_t*)o.sorted, T, (const uint64_t*)rr->index,
                               (const uint8_t*)rr->raf, o.index_sorted, o.raf_sorted);
        JOLT_HIP_TRY(ctx, hipGetLastError());
        o.suffix_len = (int64_t)len;
        return JOLT_OK;
    };
    if (!rr->lds_attr_set) {
        (void)hipFuncSetAttribute((const void*)k_msm_hist_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->max_lds_per_block);
        (void)hipFuncSetAttribute((const void*)k_msm_scatter_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->max_lds_per_block);
        (void)hipGetLastError();
        rr->lds_attr_set = true;
    }
    // page-locked read-back block: the 6 x 256 RAF sums, then the suffix sums
    const size_t out_words = (size_t)kRafSums * kRafChunk + (size_t)total_suffixes * kRafChunk;
    if (out_words > rr->h_out_cap) {
        if (rr->h_out) { JOLT_HIP_TRY(ctx, hipStreamSynchronize(st)); JOLT_HIP_TRY(ctx, hipHostFree(rr->h_out)); rr->h_out = nullptr; rr->h_out_cap = 0; }
        JOLT_HIP_TRY(ctx, hipHostMalloc(&rr->h_out, out_words * sizeof(Fr), hipHostMallocDefault));
        rr->h_out_cap = out_words;
    }
    if (rr->order[rr->cur].suffix_len != (int64_t)suffix_len) {  // not prepared by the previous call (first phase of a proof, or phases out of sequence)
        if (rr->order[1 - rr->cur].suffix_len == (int64_t)suffix_len) rr->cur = 1 - rr->cur;
        else JOLT_TRY(build_order(rr->order[rr->cur], suffix_len));
    }
    jolt_read_raf::Order& o = rr->order[rr->cur];
    if (total_suffixes) JOLT_HIP_TRY(ctx, hipMemsetAsync(rr->d_suffix, 0, (size_t)total_suffixes * kRafChunk * sizeof(Fr), st));
    const uint32_t upper_suffix_bits = suffix_len > address_bits / 2 ? suffix_len - address_bits / 2 : 0;  // suffix_len.saturating_sub(address_bits / 2) (:765)
    const unsigned grid = (unsigned)std::min<size_t>((max_items + 3) / 4, (size_t)ctx->num_cus * 16);
    if (gathered) {
        hipLaunchKernelGGL(k_rr_gather_u, dim3((unsigned)((T + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, (const uint32_t*)o.sorted, T, (const Fr*)u->data(), rr->u_sorted);
        hipLaunchKernelGGL(k_rr_accumulate<true>, dim3(grid), dim3(kBlock), 0, st, (const uint64_t*)o.index_sorted, (const uint8_t*)o.raf_sorted, (const Fr*)rr->u_sorted,
                           (const uint32_t*)o.sorted, (const uint32_t*)o.hist, (const uint32_t*)o.offs, n_tables, suffix_len, upper_suffix_bits, (int)canonical, (const uint32_t*)rr->d_cfg,
                           (const uint32_t*)o.seg_start, slots, kRafSegRows, rr->part);
    } else {
        hipLaunchKernelGGL(k_rr_accumulate<false>, dim3(grid), dim3(kBlock), 0, st, (const uint64_t*)rr->index, (const uint8_t*)rr->raf, (const Fr*)u->data(),
                           (const uint32_t*)o.sorted, (const uint32_t*)o.hist, (const uint32_t*)o.offs, n_tables, suffix_len, upper_suffix_bits, (int)canonical, (const uint32_t*)rr->d_cfg,
                           (const uint32_t*)o.seg_start, slots, kRafSegRows, rr->part);
    }
    hipLaunchKernelGGL(k_rr_fold_items, dim3((unsigned)(((size_t)B * slots + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, (const Fr*)rr->part, (const uint32_t*)o.seg_start, n_tables,
                       slots, (const uint32_t*)rr->d_cfg, rr->bin_raf, rr->d_suffix);
    hipLaunchKernelGGL(k_rr_fold_raf, dim3(kRafSums), dim3(kRafChunk), 0, st, (const Fr*)rr->bin_raf, n_tables + 1, rr->d_raf);
    JOLT_HIP_TRY(ctx, hipGetLastError());
    Fr* h_raf = (Fr*)rr->h_out;
    Fr* h_suf = h_raf + (size_t)kRafSums * kRafChunk;
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(h_raf, rr->d_raf, kRafSums * kRafChunk * sizeof(Fr), hipMemcpyDeviceToHost, st));
    if (total_suffixes) JOLT_HIP_TRY(ctx, hipMemcpyAsync(h_suf, rr->d_suffix, (size_t)total_suffixes * kRafChunk * sizeof(Fr), hipMemcpyDeviceToHost, st));
    JOLT_HIP_TRY(ctx, hipEventRecord(rr->ev_out, st));
    // the next phase's order, queued BEHIND the read-back: it runs while the caller's host rounds do (address phases go from the top chunk down, 8 bits at a time)
    int32_t ahead_status = JOLT_OK;
    if (ahead && suffix_len >= kRafChunkBits) ahead_status = build_order(rr->order[1 - rr->cur], suffix_len - kRafChunkBits);
    JOLT_HIP_TRY(ctx, hipEventSynchronize(rr->ev_out));
    std::memcpy(raf_out, h_raf, kRafSums * kRafChunk * sizeof(Fr));
    if (total_suffixes) std::memcpy(suffix_out, h_suf, (size_t)total_suffixes * kRafChunk * sizeof(Fr));
    return ahead_status;
}

extern "C" int32_t jolt_read_raf_condense(jolt_ctx* ctx, jolt_read_raf* rr, jolt_table* u, const jolt_fr_t* v_table, uint32_t shift) {
    if (!ctx || !rr || !u || !v_table || shift + kRafChunkBits > 128) return JOLT_ERR_INVALID_ARG;
    if (u->len != rr->cycles) return JOLT_ERR_SIZE_MISMATCH;
    for (uint32_t k = 0; k < kRafChunk; ++k) JOLT_REQUIRE(ctx, fr_is_canonical(fr_from_abi(&v_table[k])), "eq table entry is not a canonical Fr");
    Fr* dv = nullptr;
    JOLT_TRY(jolt_internal_dev_alloc(ctx, kRafChunk * sizeof(Fr), (void**)&dv));
    hipError_t e = hipMemcpyAsync(dv, v_table, kRafChunk * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) {
        const unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>((rr->cycles + kBlock - 1) / kBlock, (size_t)ctx->num_cus * 8));
        hipLaunchKernelGGL(k_rr_condense, dim3(grid), dim3(kBlock), 0, ctx->stream, (const uint64_t*)rr->index, rr->cycles, (const Fr*)dv, shift, u->data());
        e = hipGetLastError();
    }
    jolt_internal_dev_free(ctx, dv);
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return JOLT_ERR_HIP; }
    return JOLT_OK;
}

extern "C" int32_t jolt_read_raf_cycle_tables(jolt_ctx* ctx, jolt_read_raf* rr, const jolt_fr_t* table_values, const jolt_fr_t* raf_interleaved, const jolt_fr_t* raf_identity,
                                              const jolt_fr_t* v_tables, uint32_t phases, uint32_t address_bits, uint32_t ra_count, jolt_table** combined_out,
                                              jolt_table** ra_out) {
    if (!ctx || !rr || !table_values || !raf_interleaved || !raf_identity || !v_tables || !combined_out || !ra_out) return JOLT_ERR_INVALID_ARG;
    if (ra_count == 0 || ra_count > 16 || phases == 0 || phases % ra_count || phases * kRafChunkBits != address_bits || address_bits > 128) return JOLT_ERR_INVALID_ARG;
    const size_t nv = (size_t)phases * kRafChunk;
    Fr *dv = nullptr, *dt = nullptr;
    jolt_table* combined = nullptr;
    std::vector<jolt_table*> ra(ra_count, nullptr);
    int32_t s = jolt_internal_dev_alloc(ctx, nv * sizeof(Fr), (void**)&dv);
    if (s == JOLT_OK) s = jolt_internal_dev_alloc(ctx, (size_t)rr->n_tables * sizeof(Fr), (void**)&dt);
    if (s == JOLT_OK) s = jolt_internal_table_new(ctx, rr->cycles, &combined);
    for (uint32_t i = 0; i < ra_count && s == JOLT_OK; ++i) s = jolt_internal_table_new(ctx, rr->cycles, &ra[i]);
    if (s == JOLT_OK) {
        hipError_t e = hipMemcpyAsync(dv, v_tables, nv * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dt, table_values, (size_t)rr->n_tables * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) {
            CycleOut out;
            for (uint32_t i = 0; i < 16; ++i) out.ra[i] = i < ra_count ? ra[i]->data() : nullptr;
            hipLaunchKernelGGL(k_rr_cycle_tables, dim3((unsigned)((rr->cycles + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, (const uint64_t*)rr->index,
                               (const uint8_t*)rr->table, (const uint8_t*)rr->raf, rr->cycles, (const Fr*)dt, fr_from_abi(raf_interleaved), fr_from_abi(raf_identity),
                               (const Fr*)dv, address_bits, ra_count, phases / ra_count, combined->data(), out);
            e = hipGetLastError();
        }
        if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); s = JOLT_ERR_HIP; }
    }
    if (dv) jolt_internal_dev_free(ctx, dv);
    if (dt) jolt_internal_dev_free(ctx, dt);
    if (s != JOLT_OK) {
        if (combined) jolt_table_free(ctx, combined);
        for (jolt_table* t : ra)
            if (t) jolt_table_free(ctx, t);
        return s;
    }
    *combined_out = combined;
    for (uint32_t i = 0; i < ra_count; ++i) ra_out[i] = ra[i];
    return JOLT_OK;
}

extern "C" int32_t jolt_read_raf_cycles(const jolt_read_raf* rr, size_t* cycles, uint32_t* n_tables) {
    if (!rr) return JOLT_ERR_INVALID_ARG;
    if (cycles) *cycles = rr->cycles;
    if (n_tables) *n_tables = rr->n_tables;
    return JOLT_OK;
}

// suffix_mle.hip.h built for the host: the CPU suite checks the device's suffix polynomials against the oracle and its big-integer model
extern "C" int32_t jolt_host_suffix_mle(uint32_t kind, uint64_t lo, uint64_t hi, uint32_t len, uint64_t* out) {
    if (!out || kind >= (uint32_t)kNumSuffixKinds || len > 128) return JOLT_ERR_INVALID_ARG;
    if (len < 128) {  // LookupBits::new masks to `len` bits
        if (len >= 64) hi &= len == 64 ? 0ull : ((1ull << (len - 64)) - 1);
        else { hi = 0; lo &= len == 0 ? 0ull : ((1ull << len) - 1); }
    }
    *out = suffix_mle(kind, lo, hi, len);
    return JOLT_OK;
}
