// jolt_amd/csrc/pcs.hip -- the committed trace polynomials over the proof's shared commitment grid, for a KZG-type scheme
// (HyperKZG: commit = one MSM against the SRS prefix, crates/jolt-hyperkzg/src/kzg.rs:15-27).
//
// Every witness polynomial is committed in ONE grid shape of total_vars = log_k_chunk + log_t variables
// (crates/jolt-kernels/src/commitment.rs:1-8,86-130); with the cycle-major placement the coefficient of (address k, cycle j)
// sits at index k * T + j and dense columns live at k = 0 (TracePlacement, crates/jolt-kernels/src/optimized/opening.rs:340-372;
// TraceOpeningPoly::entry :404-420).  On that grid
//   * a one-hot RA column has exactly one unit coefficient per hot cycle: its commitment is a plain SUM of T selected bases --
//     no scalar work at all (the KZG twin of the one-hot batch additions of crates/jolt-dory/src/streaming.rs:230-275);
//   * the joint polynomial of the stage-8 batch opening, sum_i gamma_i * f_i (HomomorphicBatch::prove_batch,
//     crates/jolt-openings/src/schemes.rs:487-524 -> RlcSource::to_dense, crates/jolt-poly/src/multilinear.rs:159-170), is
//     written in one pass from the hot indices and the dense columns: J[k*T + j] = sum_p s_p [hot_p(j) = k] + [k = 0] sum_d c_d f_d[j].
// Also here: the promotion of device-resident integer columns to field tables (Polynomial::bind_to_field's From<T>,
// crates/jolt-poly/src/dense.rs:129-142) for inputs that are already in HBM.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "grid_hint.hpp"
#include "ints.hpp"
#include "msm_kernels.hip.h"
#include "onehot.hpp"
#include "srs.hpp"
#include "term_map.hip.h"

using namespace jolt;
using namespace jolt::msmk;

int32_t jolt_internal_table_new(jolt_ctx* ctx, size_t len, jolt_table** out);
extern "C" int32_t jolt_grid_commit_onehot_range(jolt_ctx* ctx, const jolt_srs* srs, const jolt_onehot* source, size_t cycle_lo, size_t cycle_hi, jolt_g1_t* out);

namespace {

constexpr int kGridSumBlocks = 512;  // workgroups per column: 2048 wavefront partial sums, folded by the second kernel

// partial[(p * gridDim.x + block) * 4 + wave] = sum over this wavefront's cycles of bases[hot_p(j) * T + j]
// (lo, hi): the cycle range this launch sums -- the whole column, or one rank's block of a sharded commitment
template <bool LFORM>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(JOLT_BUCKET_WAVES, JOLT_BUCKET_WAVES))) void k_grid_onehot_sum(
    const uint8_t* __restrict__ idx, uint32_t wide, size_t grid_cycles, size_t lo, size_t cycles, const G1Affine* __restrict__ bases, G1Jac* __restrict__ partial,
    LformConsts lc, uint32_t shift) {
    // shift > 0 (blockIdx.z = residue class c of the cycle mod 2^shift): the sum over the cycles j = c mod 2^shift of the bases at (hot * T + j) >> shift -- the
    // commitment of the column's class-c part on the grid of a polynomial folded `shift` times (jolt_grid_commit_onehot_classes); shift = 0 is the column itself
    const size_t p = blockIdx.y, cls = blockIdx.z;
    const uint8_t* col = hot_col(idx, p * grid_cycles, wide);
    const size_t stride = ((size_t)gridDim.x * kBlock) << shift, folded_cycles = grid_cycles >> shift;
    // XYZZ accumulator: 8M + 2S per mixed addition (g1.hip.h); LFORM: the bases are window 0 of the SRS's L-form tables and the
    // accumulator stays in limb form (fq_limb.hip.h)
    G1Xyzz acc = g1x_identity();
    G1XyzzL acc_l = g1xl_identity();
    const FqL one = fql_from_words(lc.one_l);
    // software pipeline as in sum_bucket_points<true>: the next index byte and point are in flight during the mixed addition
    size_t j = lo + cls + (((size_t)blockIdx.x * kBlock + threadIdx.x) << shift);
    uint32_t a = j < cycles ? hot_load(col, j, wide) : kColdIdx;
    G1Affine pt;
    pt.x = Fq::zero();
    pt.y = Fq::zero();
    if (a != kColdIdx) pt = ld_aff(bases + (size_t)a * folded_cycles + (j >> shift));
    while (j < cycles) {
        const size_t jn = j + stride;
        const uint32_t an = jn < cycles ? hot_load(col, jn, wide) : kColdIdx;
        G1Affine pn;
        pn.x = Fq::zero();
        pn.y = Fq::zero();
        if (an != kColdIdx) pn = ld_aff(bases + (size_t)an * folded_cycles + (jn >> shift));
        if (LFORM) {
            if (!g1_aff_is_inf(pt)) acc_l = g1xl_add_mixed(acc_l, fql_from_words(pt.x), fql_from_words(pt.y), one);
        } else {
            acc = g1x_add_mixed(acc, pt);  // (0, 0) = infinity: a cold cycle adds nothing
        }
        pt = pn;
        j = jn;
    }
    G1Jac mine = g1x_to_jac(acc);
    if (LFORM) {
        mine = g1_identity();
        if (!g1xl_is_identity(acc_l)) {
            const FqL r256 = fql_from_words(lc.r256);
            mine.x = fql_to_std(fql_mul(acc_l.x, fql_sqr(acc_l.zz)), r256);
            mine.y = fql_to_std(fql_mul(acc_l.y, fql_sqr(acc_l.zzz)), r256);
            mine.z = fql_to_std(acc_l.zzz, r256);
        }
    }
    const G1Jac total = wave_sum_g1(mine, 64);
    if ((threadIdx.x & 63) == 0) partial[((cls * gridDim.y + p) * gridDim.x + blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6)] = total;
}
// out[..] = sum of the column's `count` partial sums (one wavefront per (class, column)); blockIdx.x = cls * n_cols + p lands at out[cls * out_cols + first + p]
// (out_cols = n_cols, first = 0: out[blockIdx.x]; otherwise the columns of one source inside a row of all the sources' columns: jolt_grid_hint)
__global__ __launch_bounds__(64) void k_grid_onehot_fold(const G1Jac* __restrict__ partial, uint32_t count, G1Jac* __restrict__ out, uint32_t n_cols, uint32_t out_cols,
                                                         uint32_t first) {
    const size_t p = blockIdx.x;
    G1Jac acc = g1_identity();
    for (uint32_t k = threadIdx.x; k < count; k += 64) acc = g1_add(acc, partial[p * count + k]);
    acc = wave_sum_g1(acc, 64);
    if (threadIdx.x == 0) out[(p / n_cols) * out_cols + first + (p % n_cols)] = acc;
}
// out[level] = sum_i scalars[i] * points[i] over the few hundred (class, column) terms of one level of a grid hint: one lane per term (MSB-first double-and-add over the
// canonical scalar), wavefront sums, one workgroup per level (blockIdx.x); the terms of level l are [offsets[l], offsets[l + 1])
__global__ __launch_bounds__(1024) void k_grid_hint_combine(const G1Jac* __restrict__ points, const Fr* __restrict__ scalars, const uint32_t* __restrict__ offsets,
                                                            G1Jac* __restrict__ out) {
    __shared__ G1Jac wave_total[16];
    const uint32_t lo = offsets[blockIdx.x], hi = offsets[blockIdx.x + 1];
    G1Jac acc = g1_identity();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const Fr k = from_mont(scalars[i]);
        acc = g1_add(acc, g1_mul_canonical(points[i], k.l));
    }
    acc = wave_sum_g1(acc, 64);
    if ((threadIdx.x & 63) == 0) wave_total[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        G1Jac total = wave_total[0];
        for (uint32_t w = 1; w < (blockDim.x >> 6); ++w) total = g1_add(total, wave_total[w]);
        out[blockIdx.x] = total;
    }
}

constexpr int kJointMaxSources = 4;
constexpr int kJointMaxDense = 8;
struct JointArgs {
    const uint8_t* idx[kJointMaxSources];  // [polys of the source][cycles]
    uint32_t wide[kJointMaxSources];
    uint32_t n_polys[kJointMaxSources];
    uint32_t first[kJointMaxSources];      // offset of the source's first polynomial in `scalars`
    int n_sources;
    const Fr* dense[kJointMaxDense];
    Fr dense_scalar[kJointMaxDense];
    uint32_t dense_one[kJointMaxDense];
    int n_dense;
};
// out[k * T + j] = sum_p scalars[p] * [hot_p(j) == k]  (+ the dense columns on row k = 0); blockIdx.y = k
__global__ __launch_bounds__(kBlock) void k_grid_joint(JointArgs a, const Fr* __restrict__ scalars, size_t cycles, Fr* __restrict__ out) {
    const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const uint32_t k = blockIdx.y;
    if (j >= cycles) return;
    Fr acc = Fr::zero();
    for (int s = 0; s < a.n_sources; ++s) {
        for (uint32_t p = 0; p < a.n_polys[s]; ++p)
            if (hot_load(a.idx[s], (size_t)p * cycles + j, a.wide[s]) == k) acc = add(acc, scalars[a.first[s] + p]);  // scalar (wave-uniform) load of the coefficient
    }
    if (k == 0)
        for (int d = 0; d < a.n_dense; ++d) {
            Fr v = ld_fr(a.dense[d] + j);
            acc = add(acc, a.dense_one[d] ? v : mul(v, a.dense_scalar[d]));
        }
    st_fr(out + (size_t)k * cycles + j, acc);
}

// The same with ONE thread per cycle and its K row accumulators in LDS: k_grid_joint gives every (row, cycle) pair a thread that walks all the columns and adds under
// a predicate -- with 36 columns over 16 rows nearly every wavefront executes every addition, 16 x 36 per cycle (3.4 ms at T = 2^22, bound by those additions, where
// the 2 GiB of output take 0.4 ms).  Here a thread reads its cycle's hot addresses once and adds each coefficient into acc[hot][thread]: 36 additions per cycle.
// acc is [K][kJointThreads] in LDS: lane t touches column t of whatever row, the bank pattern of a linear access.
constexpr int kJointThreads = 128;
__global__ __launch_bounds__(kJointThreads) void k_grid_joint_rows(JointArgs a, const Fr* __restrict__ scalars, size_t cycles, uint32_t K, Fr* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char joint_raw[];
    Fr* acc = reinterpret_cast<Fr*>(joint_raw);
    const size_t j = (size_t)blockIdx.x * kJointThreads + threadIdx.x;
    const bool live = j < cycles;
    for (uint32_t k = 0; k < K; ++k) acc[k * kJointThreads + threadIdx.x] = Fr::zero();
    if (live) {
        for (int s = 0; s < a.n_sources; ++s) {
            for (uint32_t p = 0; p < a.n_polys[s]; ++p) {
                const uint32_t h = hot_load(a.idx[s], (size_t)p * cycles + j, a.wide[s]);
                if (h < K) {  // cold cycles (the sentinel) select no row
                    Fr* slot = acc + h * kJointThreads + threadIdx.x;
                    *slot = add(*slot, scalars[a.first[s] + p]);
                }
            }
        }
        Fr d0 = acc[threadIdx.x];
        for (int d = 0; d < a.n_dense; ++d) {
            Fr v = ld_fr(a.dense[d] + j);
            d0 = add(d0, a.dense_one[d] ? v : mul(v, a.dense_scalar[d]));
        }
        acc[threadIdx.x] = d0;
        for (uint32_t k = 0; k < K; ++k) st_fr(out + (size_t)k * cycles + j, acc[k * kJointThreads + threadIdx.x]);  // a thread only ever touches its own column: no barrier
    }
}

// The same polynomial restricted to the coefficients ONE RANK owns under a sharded term assignment (term_map.hip.h), as its compact
// array: out[c] = J[term_global(map, c)] -- every rank builds its 1 / world of the joint polynomial from the raw columns of the trace
__global__ __launch_bounds__(kBlock) void k_grid_joint_owned(JointArgs a, const Fr* __restrict__ scalars, size_t cycles, size_t len, TermMap map, Fr* __restrict__ out) {
    const size_t c = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (c >= len) return;
    const size_t pos = term_global(map, c);
    const uint32_t k = (uint32_t)(pos / cycles);
    const size_t j = pos % cycles;
    Fr acc = Fr::zero();
    for (int s = 0; s < a.n_sources; ++s) {
        for (uint32_t p = 0; p < a.n_polys[s]; ++p)
            if (hot_load(a.idx[s], (size_t)p * cycles + j, a.wide[s]) == k) acc = add(acc, ld_fr(scalars + a.first[s] + p));
    }
    if (k == 0)
        for (int d = 0; d < a.n_dense; ++d) {
            Fr v = ld_fr(a.dense[d] + j);
            acc = add(acc, a.dense_one[d] ? v : mul(v, a.dense_scalar[d]));
        }
    st_fr(out + c, acc);
}

// Ring::from_u64 / from_i64 / from_i128 per entry (crates/jolt-field/src/bn254/mod.rs:265-328) from device-resident integers
template <int KIND>
__global__ __launch_bounds__(kBlock) void k_promote_ints(const void* __restrict__ data, size_t offset, size_t n, Fr two64, Fr* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        uint64_t lo, hi = 0;
        bool negative = false;
        if (KIND == JOLT_INT_I128) {
            const uint64_t* p = reinterpret_cast<const uint64_t*>(data) + 2 * (offset + i);
            lo = p[0];
            hi = p[1];
            negative = (hi >> 63) != 0;
            if (negative) {
                lo = ~lo + 1;
                hi = ~hi + (lo == 0 ? 1 : 0);
            }
        } else {
            lo = reinterpret_cast<const uint64_t*>(data)[offset + i];
            negative = KIND == JOLT_INT_I64 && (lo >> 63) != 0;
            if (negative) lo = ~lo + 1;
        }
        Fr m = fr_from_u64(lo);
        if (KIND == JOLT_INT_I128 && hi) m = add(m, mul(fr_from_u64(hi), two64));
        st_fr(out + i, negative ? neg(m) : m);
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// entry points
// ------------------------------------------------------------------------------------------------------------------
extern "C" int32_t jolt_table_from_ints(jolt_ctx* ctx, const jolt_ints* values, size_t offset, size_t len, jolt_table** out) {
    if (!ctx || !values || !out) return JOLT_ERR_INVALID_ARG;
    if (len > values->count || offset > values->count - len) return JOLT_ERR_SIZE_MISMATCH;
    jolt_table* t = nullptr;
    JOLT_TRY(jolt_internal_table_new(ctx, len, &t));
    if (len) {
        const Fr t32 = fr_from_u64((uint64_t)1 << 32);
        const Fr two64 = mul(t32, t32);
        const unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>((len + kBlock - 1) / kBlock, (size_t)ctx->num_cus * 8));
        switch (values->kind) {
            case JOLT_INT_U64: hipLaunchKernelGGL(k_promote_ints<JOLT_INT_U64>, dim3(grid), dim3(kBlock), 0, ctx->stream, (const void*)values->data, offset, len, two64, t->data()); break;
            case JOLT_INT_I64: hipLaunchKernelGGL(k_promote_ints<JOLT_INT_I64>, dim3(grid), dim3(kBlock), 0, ctx->stream, (const void*)values->data, offset, len, two64, t->data()); break;
            default: hipLaunchKernelGGL(k_promote_ints<JOLT_INT_I128>, dim3(grid), dim3(kBlock), 0, ctx->stream, (const void*)values->data, offset, len, two64, t->data()); break;
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { jolt_table_free(ctx, t); ctx->last_error = hipGetErrorString(e); return JOLT_ERR_HIP; }
    }
    *out = t;
    return JOLT_OK;
}

extern "C" int32_t jolt_grid_commit_onehot(jolt_ctx* ctx, const jolt_srs* srs, const jolt_onehot* source, jolt_g1_t* out) {
    if (!source) return JOLT_ERR_INVALID_ARG;
    return jolt_grid_commit_onehot_range(ctx, srs, source, 0, source->cycles, out);
}

// the partial commitments over cycles [cycle_lo, cycle_hi): one rank's share of a commitment sharded over the cycles (the ranks'
// partial points add up to jolt_grid_commit_onehot's)
static int32_t grid_commit_onehot_impl(jolt_ctx* ctx, const jolt_srs* srs, const jolt_onehot* source, size_t cycle_lo, size_t cycle_hi, uint32_t shift, jolt_g1_t* out);
extern "C" int32_t jolt_grid_commit_onehot_range(jolt_ctx* ctx, const jolt_srs* srs, const jolt_onehot* source, size_t cycle_lo, size_t cycle_hi, jolt_g1_t* out) {
    return grid_commit_onehot_impl(ctx, srs, source, cycle_lo, cycle_hi, 0, out);
}
// out[c * n_polys + p] = sum over the cycles j = c mod 2^shift of srs[(hot_p(j) * T + j) >> shift]: the commitments of the 2^shift residue-class parts of every
// column on the grid of a polynomial folded `shift` times low to high.  With them the first level commitments of an opening of the joint polynomial follow by
// linearity instead of by MSM: com(P_1) = sum_p s_p ((1 - x) out[0][p] + x out[1][p]) + com(the dense columns' fold) for the fold variable x -- 36 sums of bases
// at the commit leg's rate where the MSM sorts and sums 2^25 full-width scalars (jolt_host_hyperkzg_open_with_levels takes the result).
extern "C" int32_t jolt_grid_commit_onehot_classes(jolt_ctx* ctx, const jolt_srs* srs, const jolt_onehot* source, uint32_t shift, jolt_g1_t* out) {
    if (!source) return JOLT_ERR_INVALID_ARG;
    if (shift > 4 || (source->cycles & (source->cycles - 1)) != 0 || source->cycles < ((size_t)1 << shift)) return JOLT_ERR_UNSUPPORTED;  // a power-of-two grid
    return grid_commit_onehot_impl(ctx, srs, source, 0, source->cycles, shift, out);
}
static int32_t grid_commit_onehot_impl(jolt_ctx* ctx, const jolt_srs* srs, const jolt_onehot* source, size_t cycle_lo, size_t cycle_hi, uint32_t shift, jolt_g1_t* out) {
    if (!ctx || !srs || !source || !out) return JOLT_ERR_INVALID_ARG;
    const size_t T = source->cycles, classes = (size_t)1 << shift, N = source->n_polys;
    if (cycle_lo > cycle_hi || cycle_hi > T) return JOLT_ERR_SIZE_MISMATCH;
    if ((size_t)source->k * T > srs->n) return JOLT_ERR_SRS_TOO_SMALL;  // HyperKZGError::SrsTooSmall (kzg.rs:19-24)
    if (N > 65535) return JOLT_ERR_UNSUPPORTED;
    // lanes per column: every lane ends with an XYZZ -> Jacobian conversion and six shuffle rounds of full additions (~11 mixed additions' worth), so a lane should own
    // >= 128 cycles (32 cycles per lane at T = 2^22 made that a third of the kernel); enough workgroups over all columns to fill the chip all the same
    const size_t span = (cycle_hi - cycle_lo) >> shift;  // cycles per (column, class)
    const size_t by_work = (span + (size_t)kBlock * 128 - 1) / ((size_t)kBlock * 128), fill = ((size_t)ctx->num_cus * 8 + N - 1) / std::max<size_t>(N, 1);
    const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>({(span + kBlock - 1) / kBlock, std::max(by_work, fill), (size_t)kGridSumBlocks}));
    const uint32_t per_col = blocks * (kBlock / 64);
    G1Jac *partial = nullptr, *sums = nullptr;
    JOLT_TRY(jolt_internal_dev_alloc(ctx, classes * N * per_col * sizeof(G1Jac), (void**)&partial));
    int32_t st = jolt_internal_dev_alloc(ctx, classes * N * sizeof(G1Jac), (void**)&sums);
    if (st != JOLT_OK) { jolt_internal_dev_free(ctx, partial); return st; }
    LformConsts lc;
    {
        Fq thirty_two = Fq::zero();
        thirty_two.l[0] = 32;
        lc.one_l = to_mont(thirty_two);
        lc.r256 = Fq::one();
    }
    // window 0 of the fixed-base tables IS the SRS in L-form (msm_fixed.hip): the sums then run on the limb-form accumulator
    if (srs->pre && srs->pre_lform && srs->pre_stride >= (size_t)source->k * T)
        hipLaunchKernelGGL(k_grid_onehot_sum<true>, dim3(blocks, (unsigned)N, (unsigned)classes), dim3(kBlock), 0, ctx->stream, (const uint8_t*)source->idx, source->wide, T, cycle_lo,
                           cycle_hi, (const G1Affine*)srs->pre, partial, lc, shift);
    else
        hipLaunchKernelGGL(k_grid_onehot_sum<false>, dim3(blocks, (unsigned)N, (unsigned)classes), dim3(kBlock), 0, ctx->stream, (const uint8_t*)source->idx, source->wide, T, cycle_lo,
                           cycle_hi, (const G1Affine*)srs->pts, partial, lc, shift);
    hipLaunchKernelGGL(k_grid_onehot_fold, dim3((unsigned)(classes * N)), dim3(64), 0, ctx->stream, (const G1Jac*)partial, per_col, sums, (uint32_t)N, (uint32_t)N, 0u);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, sums, classes * N * sizeof(G1Jac), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    jolt_internal_dev_free(ctx, partial);
    jolt_internal_dev_free(ctx, sums);
    if (e != hipSuccess) { ctx->last_error = std::string("grid one-hot commit: ") + hipGetErrorString(e); return JOLT_ERR_HIP; }
    return JOLT_OK;
}

// ---- the opening hint of the grid's one-hot columns (grid_hint.hpp) ------------------------------------------------------------------------------------------
static int32_t hint_stream_of(jolt_ctx* ctx, hipStream_t* out) {
    if (!ctx->hint_stream) {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);  // numerically lowest = highest priority
        if (hipStreamCreateWithPriority(&ctx->hint_stream, hipStreamNonBlocking, least) != hipSuccess) {
            (void)hipGetLastError();
            JOLT_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->hint_stream, hipStreamNonBlocking));
        }
    }
    *out = ctx->hint_stream;
    return JOLT_OK;
}
extern "C" int32_t jolt_grid_hint_free(jolt_ctx* ctx, jolt_grid_hint* h) {
    if (!h) return JOLT_OK;
    if (!ctx || h->ctx != ctx) return JOLT_ERR_INVALID_ARG;
    // the buffers go back to the pool, whose blocks later main-stream work may reuse at once: the main stream first joins the stream that wrote them
    if (h->ready) {
        if (h->stream != ctx->stream) (void)hipStreamWaitEvent(ctx->stream, h->ready, 0);
        (void)hipEventDestroy(h->ready);
    }
    if (h->sums) jolt_internal_dev_free(ctx, h->sums);
    if (h->partial) jolt_internal_dev_free(ctx, h->partial);
    delete h;
    return JOLT_OK;
}
extern "C" int32_t jolt_grid_hint_begin(jolt_ctx* ctx, const jolt_srs* srs, const jolt_onehot* const* sources, size_t n_sources, uint32_t levels, int32_t background,
                                        jolt_grid_hint** out) {
    if (!ctx || !srs || !sources || !n_sources || !out || levels == 0 || levels > 4) return JOLT_ERR_INVALID_ARG;
    *out = nullptr;
    const size_t T = sources[0] ? sources[0]->cycles : 0;
    size_t n_cols = 0, max_cols = 0;
    for (size_t q = 0; q < n_sources; ++q) {
        if (!sources[q] || sources[q]->cycles != T || sources[q]->k != sources[0]->k || sources[q]->n_polys > 65535) return JOLT_ERR_INVALID_ARG;
        n_cols += sources[q]->n_polys;
        max_cols = std::max(max_cols, sources[q]->n_polys);
    }
    if (T == 0 || (T & (T - 1)) != 0 || T < ((size_t)1 << levels)) return JOLT_ERR_UNSUPPORTED;  // a power-of-two grid at least 2^levels wide
    if ((size_t)sources[0]->k * T > srs->n) return JOLT_ERR_SRS_TOO_SMALL;
    JOLT_TRY(jolt_internal_engine_quiesce(ctx));
    jolt_grid_hint* h = new (std::nothrow) jolt_grid_hint();
    if (!h) return JOLT_ERR_OOM;
    h->ctx = ctx;
    h->levels = levels;
    h->k = sources[0]->k;
    h->n_cols = n_cols;
    h->cycles = T;
    for (uint32_t s_ = 1; s_ <= levels; ++s_) h->level_offset[s_] = h->level_offset[s_ - 1] + ((size_t)n_cols << s_);
    // one wavefront per SIMD in the background: a 256-lane workgroup that reserves most of a CU's LDS keeps the other wave slots (and 2 / 3 of the registers) free
    // for whatever the main stream launches meanwhile -- latency-bound round kernels find a slot at once instead of waiting for one of these long workgroups to retire
    const size_t lds = background ? std::min<size_t>(ctx->max_lds_per_block, (size_t)96 * 1024) : 0;
    int32_t st = JOLT_OK;
    if (background) st = hint_stream_of(ctx, &h->stream);
    else h->stream = ctx->stream;
    const size_t span_min = T >> levels;
    const size_t blocks_max = std::max<size_t>(1, std::min<size_t>((T / 2 + kBlock - 1) / kBlock, (size_t)kGridSumBlocks));
    (void)span_min;
    if (st == JOLT_OK) st = jolt_internal_dev_alloc(ctx, h->level_offset[levels] * sizeof(G1Jac), (void**)&h->sums);
    if (st == JOLT_OK) st = jolt_internal_dev_alloc(ctx, ((size_t)max_cols << levels) * blocks_max * (kBlock / 64) * sizeof(G1Jac), (void**)&h->partial);
    hipError_t e = hipSuccess;
    if (st == JOLT_OK) {
        e = hipEventCreateWithFlags(&h->ready, hipEventDisableTiming);
        if (e == hipSuccess && h->stream != ctx->stream) {  // the sources were produced on the main stream; pool blocks may still be read by work queued there
            e = hipEventRecord(ctx->ev_fork, ctx->stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(h->stream, ctx->ev_fork, 0);
        }
        if (e == hipSuccess && lds && !ctx->grid_hint_attr_set) {
            e = hipFuncSetAttribute((const void*)k_grid_onehot_sum<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_grid_onehot_sum<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess) ctx->grid_hint_attr_set = true;
        }
    }
    LformConsts lc;
    {
        Fq thirty_two = Fq::zero();
        thirty_two.l[0] = 32;
        lc.one_l = to_mont(thirty_two);
        lc.r256 = Fq::one();
    }
    for (uint32_t s_ = 1; s_ <= levels && st == JOLT_OK && e == hipSuccess; ++s_) {
        const size_t classes = (size_t)1 << s_, span = T >> s_;
        uint32_t first = 0;
        for (size_t q = 0; q < n_sources; ++q) {
            const jolt_onehot* src = sources[q];
            const size_t N = src->n_polys;
            const size_t by_work = (span + (size_t)kBlock * 128 - 1) / ((size_t)kBlock * 128), fill = ((size_t)ctx->num_cus * 8 + N - 1) / std::max<size_t>(N, 1);
            const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>({(span + kBlock - 1) / kBlock, std::max(by_work, fill), blocks_max}));
            const uint32_t per_col = blocks * (kBlock / 64);
            if (srs->pre && srs->pre_lform && srs->pre_stride >= (size_t)src->k * T)
                hipLaunchKernelGGL(k_grid_onehot_sum<true>, dim3(blocks, (unsigned)N, (unsigned)classes), dim3(kBlock), lds, h->stream, (const uint8_t*)src->idx, src->wide, T, (size_t)0, T,
                                   (const G1Affine*)srs->pre, h->partial, lc, s_);
            else
                hipLaunchKernelGGL(k_grid_onehot_sum<false>, dim3(blocks, (unsigned)N, (unsigned)classes), dim3(kBlock), lds, h->stream, (const uint8_t*)src->idx, src->wide, T, (size_t)0, T,
                                   (const G1Affine*)srs->pts, h->partial, lc, s_);
            hipLaunchKernelGGL(k_grid_onehot_fold, dim3((unsigned)(classes * N)), dim3(64), 0, h->stream, (const G1Jac*)h->partial, per_col, h->sums + h->level_offset[s_ - 1],
                               (uint32_t)N, (uint32_t)n_cols, first);
            first += (uint32_t)N;
            e = hipGetLastError();
            if (e != hipSuccess) break;
        }
    }
    if (st == JOLT_OK && e == hipSuccess) e = hipEventRecord(h->ready, h->stream);
    if (st == JOLT_OK && e != hipSuccess) {
        (void)hipGetLastError();
        ctx->last_error = std::string("grid hint: ") + hipGetErrorString(e);
        st = JOLT_ERR_HIP;
    }
    if (st != JOLT_OK) {
        if (h->stream) (void)hipStreamSynchronize(h->stream);
        jolt_grid_hint_free(ctx, h);
        return st;
    }
    *out = h;
    return JOLT_OK;
}
extern "C" int32_t jolt_grid_hint_wait(jolt_ctx* ctx, jolt_grid_hint* h) {  // host-side: the sums have landed (tests, timing)
    if (!ctx || !h || h->ctx != ctx) return JOLT_ERR_INVALID_ARG;
    JOLT_HIP_TRY(ctx, hipEventSynchronize(h->ready));
    return JOLT_OK;
}
// test hook: the class sums of level `level` (1-based) as host points, [class][column] (= jolt_grid_commit_onehot_classes of each source, concatenated per class)
extern "C" int32_t jolt_grid_hint_download(jolt_ctx* ctx, jolt_grid_hint* h, uint32_t level, jolt_g1_t* out) {
    if (!ctx || !h || h->ctx != ctx || !out || level == 0 || level > h->levels) return JOLT_ERR_INVALID_ARG;
    JOLT_HIP_TRY(ctx, hipEventSynchronize(h->ready));
    JOLT_HIP_TRY(ctx, hipMemcpy(out, h->sums + h->level_offset[level - 1], (h->n_cols << level) * sizeof(G1Jac), hipMemcpyDeviceToHost));
    return JOLT_OK;
}
int32_t jolt_internal_grid_hint_combine(jolt_ctx* ctx, const jolt_grid_hint* h, uint32_t levels, const Fr* onehot_scalars, const Fr* xs, G1Jac** d_out, void** d_temp) {
    if (!ctx || !h || h->ctx != ctx || !onehot_scalars || !xs || !d_out || !d_temp || levels == 0 || levels > h->levels) return JOLT_ERR_INVALID_ARG;
    *d_out = nullptr;
    *d_temp = nullptr;
    const size_t total = h->level_offset[levels];
    std::vector<Fr> scalars(total);
    std::vector<uint32_t> offsets(levels + 1);
    for (uint32_t s_ = 1; s_ <= levels; ++s_) {
        offsets[s_ - 1] = (uint32_t)h->level_offset[s_ - 1];
        for (size_t c = 0; c < ((size_t)1 << s_); ++c) {
            Fr w = Fr::one();
            for (uint32_t b = 0; b < s_; ++b) w = mul(w, ((c >> b) & 1) ? xs[b] : sub(Fr::one(), xs[b]));
            for (size_t p = 0; p < h->n_cols; ++p) scalars[h->level_offset[s_ - 1] + c * h->n_cols + p] = mul(onehot_scalars[p], w);
        }
    }
    offsets[levels] = (uint32_t)total;
    const size_t bytes_scalars = total * sizeof(Fr), bytes_offsets = ((size_t)levels + 1) * sizeof(uint32_t);
    unsigned char* temp = nullptr;
    G1Jac* dout = nullptr;
    JOLT_TRY(jolt_internal_dev_alloc(ctx, bytes_scalars + bytes_offsets, (void**)&temp));
    int32_t st = jolt_internal_dev_alloc(ctx, (size_t)levels * sizeof(G1Jac), (void**)&dout);
    if (st != JOLT_OK) { jolt_internal_dev_free(ctx, temp); return st; }
    hipError_t e = hipSuccess;
    if (h->stream != ctx->stream) {  // fresh pool blocks may still be read by work queued on the main stream
        e = hipEventRecord(ctx->ev_fork, ctx->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(h->stream, ctx->ev_fork, 0);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(temp, scalars.data(), bytes_scalars, hipMemcpyHostToDevice, h->stream);  // pageable sources: staged before the call returns
    if (e == hipSuccess) e = hipMemcpyAsync(temp + bytes_scalars, offsets.data(), bytes_offsets, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(h->stream, h->ready, 0);
    if (e == hipSuccess) {
        const size_t widest = h->n_cols << levels;
        const unsigned threads = (unsigned)std::min<size_t>(1024, (widest + 63) / 64 * 64);
        hipLaunchKernelGGL(k_grid_hint_combine, dim3(levels), dim3(threads), 0, h->stream, (const G1Jac*)h->sums, (const Fr*)temp, (const uint32_t*)(temp + bytes_scalars), dout);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(h->stream);
        jolt_internal_dev_free(ctx, temp);
        jolt_internal_dev_free(ctx, dout);
        ctx->last_error = std::string("grid hint combine: ") + hipGetErrorString(e);
        return JOLT_ERR_HIP;
    }
    *d_out = dout;
    *d_temp = temp;
    return JOLT_OK;
}

static int32_t grid_joint_impl(jolt_ctx* ctx, const jolt_onehot* const* sources, size_t n_sources, const jolt_fr_t* onehot_scalars, jolt_table* const* dense,
                               size_t n_dense, const jolt_fr_t* dense_scalars, uint32_t log_k, const TermMap& map, jolt_table** out);
extern "C" int32_t jolt_grid_joint_polynomial(jolt_ctx* ctx, const jolt_onehot* const* sources, size_t n_sources, const jolt_fr_t* onehot_scalars,
                                              jolt_table* const* dense, size_t n_dense, const jolt_fr_t* dense_scalars, uint32_t log_k, jolt_table** out) {
    return grid_joint_impl(ctx, sources, n_sources, onehot_scalars, dense, n_dense, dense_scalars, log_k, TermMap{}, out);
}
// one rank's compact array of the same polynomial under the subtree assignment (the input of jolt_host_hyperkzg_open_subtree):
// 2^log_k * T / world coefficients, built from the columns of the WHOLE trace
extern "C" int32_t jolt_grid_joint_polynomial_subtree(jolt_ctx* ctx, const jolt_onehot* const* sources, size_t n_sources, const jolt_fr_t* onehot_scalars,
                                                      jolt_table* const* dense, size_t n_dense, const jolt_fr_t* dense_scalars, uint32_t log_k, int32_t rank,
                                                      int32_t world, jolt_table** out) {
    TermMap m;
    if (!make_subtree_map(rank, world, &m)) return JOLT_ERR_INVALID_ARG;
    return grid_joint_impl(ctx, sources, n_sources, onehot_scalars, dense, n_dense, dense_scalars, log_k, m, out);
}
static int32_t grid_joint_impl(jolt_ctx* ctx, const jolt_onehot* const* sources, size_t n_sources, const jolt_fr_t* onehot_scalars, jolt_table* const* dense,
                               size_t n_dense, const jolt_fr_t* dense_scalars, uint32_t log_k, const TermMap& map, jolt_table** out) {
    if (!ctx || !out || (n_sources && (!sources || !onehot_scalars)) || (n_dense && (!dense || !dense_scalars))) return JOLT_ERR_INVALID_ARG;
    if (n_sources > (size_t)kJointMaxSources || n_dense > (size_t)kJointMaxDense || log_k > 8 || n_sources + n_dense == 0) return JOLT_ERR_UNSUPPORTED;
    const uint32_t K = 1u << log_k;
    JointArgs a;
    std::memset(&a, 0, sizeof(a));
    size_t T = n_sources ? sources[0]->cycles : dense[0]->len, total = 0;
    for (size_t s = 0; s < n_sources; ++s) {
        if (!sources[s]) return JOLT_ERR_INVALID_ARG;
        if (sources[s]->cycles != T) return JOLT_ERR_SIZE_MISMATCH;
        if (sources[s]->k > K) return JOLT_ERR_SIZE_MISMATCH;  // a hot address outside the grid
        a.idx[s] = sources[s]->idx;
        a.wide[s] = sources[s]->wide;
        a.n_polys[s] = (uint32_t)sources[s]->n_polys;
        a.first[s] = (uint32_t)total;
        total += sources[s]->n_polys;
    }
    a.n_sources = (int)n_sources;
    for (size_t d = 0; d < n_dense; ++d) {
        if (!dense[d]) return JOLT_ERR_INVALID_ARG;
        if (dense[d]->len != T) return JOLT_ERR_SIZE_MISMATCH;
        a.dense[d] = dense[d]->data();
        a.dense_scalar[d] = fr_from_abi(&dense_scalars[d]);
        JOLT_REQUIRE(ctx, fr_is_canonical(a.dense_scalar[d]), "scalar is not a canonical Fr");
        a.dense_one[d] = a.dense_scalar[d] == Fr::one() ? 1u : 0u;
    }
    a.n_dense = (int)n_dense;
    for (size_t p = 0; p < total; ++p) JOLT_REQUIRE(ctx, fr_is_canonical(fr_from_abi(&onehot_scalars[p])), "scalar is not a canonical Fr");
    jolt_table *r = nullptr, *ds = nullptr;
    const size_t len = term_owned(map, (size_t)K * T);
    if (map.kind != kTermsAll && ((T & (T - 1)) != 0 || len * map.world != (size_t)K * T)) return JOLT_ERR_SIZE_MISMATCH;  // a power-of-two grid, evenly owned
    JOLT_TRY(jolt_internal_table_new(ctx, len, &r));
    int32_t st = JOLT_OK;
    if (total) st = jolt_table_upload(ctx, onehot_scalars, total, &ds);  // synchronises: the caller's array may be short-lived
    if (st != JOLT_OK) { jolt_table_free(ctx, r); return st; }
    static const bool by_rows = !(std::getenv("JOLT_JOINT_ROWS") && std::atoi(std::getenv("JOLT_JOINT_ROWS")) == 0);
    const size_t rows_lds = (size_t)K * kJointThreads * sizeof(Fr);
    if (map.kind == kTermsAll && by_rows && rows_lds <= 64 * 1024)
        hipLaunchKernelGGL(k_grid_joint_rows, dim3((unsigned)((T + kJointThreads - 1) / kJointThreads)), dim3(kJointThreads), rows_lds, ctx->stream, a,
                           ds ? (const Fr*)ds->data() : (const Fr*)nullptr, T, K, r->data());
    else if (map.kind == kTermsAll)
        hipLaunchKernelGGL(k_grid_joint, dim3((unsigned)((T + kBlock - 1) / kBlock), K), dim3(kBlock), 0, ctx->stream, a, ds ? (const Fr*)ds->data() : (const Fr*)nullptr, T, r->data());
    else
        hipLaunchKernelGGL(k_grid_joint_owned, dim3((unsigned)((len + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, a, ds ? (const Fr*)ds->data() : (const Fr*)nullptr, T, len, map,
                           r->data());
    hipError_t e = hipGetLastError();
    if (ds) jolt_table_free(ctx, ds);
    if (e != hipSuccess) { jolt_table_free(ctx, r); ctx->last_error = hipGetErrorString(e); return JOLT_ERR_HIP; }
    *out = r;
    return JOLT_OK;
}
