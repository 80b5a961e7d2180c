// jolt_amd/csrc/views.hip -- derived-table builders the dense members need and the joint
// polynomial of a homomorphic batch opening (a12).  All are single streaming passes (HBM bound, <= 1 multiply per entry).
//   address_fold / cycle_fold / tile / replicate_stream_lsb   crates/jolt-kernels/src/reference/views.rs:35-138
//   RlcSource::to_dense (sum_i s_i f_i)                        crates/jolt-poly/src/multilinear.rs:159-170,358-464,
//                                                              HomomorphicBatch::prove_batch crates/jolt-openings/src/schemes.rs:487-524
#include <algorithm>

#include "ctx.hpp"
#include "poly_kernels.hip.h"

using namespace jolt;

namespace {

// out[j] = sum_k w[k] * grid[(k << log_t) | j]
static __global__ __launch_bounds__(kBlock) void k_address_fold(const Fr* __restrict__ grid, const Fr* __restrict__ w, size_t addresses, size_t cycles,
                                                                Fr* __restrict__ out) {
    size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= cycles) return;
    Fr acc = Fr::zero();
    for (size_t k = 0; k < addresses; ++k) acc = add(acc, mul(ld_fr(grid + k * cycles + j), ld_fr(w + k)));
    st_fr(out + j, acc);
}

// partials[k * gridDim.x + b] = sum over this block's cycles of w[j] * grid[k * cycles + j]
static __global__ __launch_bounds__(kBlock) void k_cycle_fold(const Fr* __restrict__ grid, const Fr* __restrict__ w, size_t cycles, Fr* __restrict__ partials) {
    const size_t k = blockIdx.y;
    Fr acc[1] = {Fr::zero()};
    size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x; j < cycles; j += stride) acc[0] = add(acc[0], mul(ld_fr(grid + k * cycles + j), ld_fr(w + j)));
    block_reduce_store<1>(acc, partials + k * gridDim.x);
}
static __global__ __launch_bounds__(kBlock) void k_reduce_rows(const Fr* __restrict__ partials, int nb, Fr* __restrict__ out) {
    __shared__ Fr sm[kBlock];
    const size_t k = blockIdx.x;
    Fr s = Fr::zero();
    for (int b = threadIdx.x; b < nb; b += kBlock) s = add(s, ld_fr(partials + k * nb + b));
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int off = kBlock / 2; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) sm[threadIdx.x] = add(sm[threadIdx.x], sm[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) st_fr(out + k, sm[0]);
}

static __global__ __launch_bounds__(kBlock) void k_tile(const Fr* __restrict__ base, size_t len, size_t total, Fr* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < total) st_fr(out + i, ld_fr(base + (i % len)));
}
static __global__ __launch_bounds__(kBlock) void k_replicate_lsb(const Fr* __restrict__ base, size_t total, Fr* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < total) st_fr(out + i, ld_fr(base + (i >> 1)));
}

// pointers, scalars and unit flags by value (wave-uniform kernel arguments): no device staging, nothing to wait for on the host
struct RlcTables {
    const Fr* t[kMaxBatchTables];
    Fr s[kMaxBatchTables];
    uint32_t one[kMaxBatchTables];
    int k;
};
static __global__ __launch_bounds__(kBlock) void k_rlc_equal(RlcTables a, size_t n, Fr* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Fr acc = Fr::zero();
    for (int j = 0; j < a.k; ++j) {
        Fr v = ld_fr(a.t[j] + i);
        if (!a.one[j]) v = mul(v, a.s[j]);
        acc = add(acc, v);
    }
    st_fr(out + i, acc);
}

// sum_i a[i] * b[i] through the deferred-reduction accumulator (WideAccumulator::fmadd / reduce, crates/jolt-field/src/bn254/mont.rs:
// 334-602): per thread an unreduced 512-bit sum of products, ONE Montgomery reduction per kWideMaxProducts products
static __global__ __launch_bounds__(kBlock) void k_dot_wide(const Fr* __restrict__ a, const Fr* __restrict__ b, size_t n, Fr* __restrict__ partials) {
    Fr acc[1] = {Fr::zero()};
    WideAcc<FrParams> w = wide_zero<FrParams>();
    int pending = 0;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        wide_fmadd(w, ld_fr(a + i), ld_fr(b + i));
        if (++pending == kWideMaxProducts) {
            acc[0] = add(acc[0], wide_reduce(w));
            w = wide_zero<FrParams>();
            pending = 0;
        }
    }
    if (pending) acc[0] = add(acc[0], wide_reduce(w));
    block_reduce_store<1>(acc, partials);
}

unsigned blocks_for(size_t n) { return (unsigned)std::max<size_t>(1, (n + kBlock - 1) / kBlock); }

}  // namespace

// address_fold (views.rs:35-64): grid is address-major (K x T), weights = K entries (normally eq(point, .))
extern "C" int32_t jolt_address_fold(jolt_ctx* ctx, const jolt_table* grid, const jolt_table* weights, jolt_table** out) {
    if (!ctx || !grid || !weights || !out) return JOLT_ERR_INVALID_ARG;
    size_t addresses = weights->len;
    if (addresses == 0 || grid->len % addresses != 0) return JOLT_ERR_SIZE_MISMATCH;  // KernelError::TableSizeMismatch
    size_t cycles = grid->len / addresses;
    jolt_table* r = nullptr;
    JOLT_TRY(jolt_internal_table_new(ctx, cycles, &r));
    hipLaunchKernelGGL(k_address_fold, dim3(blocks_for(cycles)), dim3(kBlock), 0, ctx->stream, (const Fr*)grid->data(), (const Fr*)weights->data(), addresses, cycles,
                       r->data());
    JOLT_HIP_TRY(ctx, hipGetLastError());
    *out = r;
    return JOLT_OK;
}

// cycle_fold (views.rs:66-95): weights = T entries; out has K entries
extern "C" int32_t jolt_cycle_fold(jolt_ctx* ctx, const jolt_table* grid, const jolt_table* weights, jolt_table** out) {
    if (!ctx || !grid || !weights || !out) return JOLT_ERR_INVALID_ARG;
    size_t cycles = weights->len;
    if (cycles == 0 || grid->len % cycles != 0) return JOLT_ERR_SIZE_MISMATCH;
    size_t addresses = grid->len / cycles;
    if (addresses > 65535) return JOLT_ERR_UNSUPPORTED;
    unsigned nb = std::min<unsigned>(blocks_for(cycles), std::max<unsigned>(1, (unsigned)(2048 / addresses)));
    JOLT_TRY(jolt_internal_ensure_scratch(ctx, addresses * nb + 8, 8));
    jolt_table* r = nullptr;
    JOLT_TRY(jolt_internal_table_new(ctx, addresses, &r));
    hipLaunchKernelGGL(k_cycle_fold, dim3(nb, (unsigned)addresses), dim3(kBlock), 0, ctx->stream, (const Fr*)grid->data(), (const Fr*)weights->data(), cycles,
                       ctx->d_partials);
    hipLaunchKernelGGL(k_reduce_rows, dim3((unsigned)addresses), dim3(kBlock), 0, ctx->stream, (const Fr*)ctx->d_partials, (int)nb, r->data());
    JOLT_HIP_TRY(ctx, hipGetLastError());
    *out = r;
    return JOLT_OK;
}

// tile (views.rs:97-113) and replicate_stream_lsb (views.rs:115-138)
extern "C" int32_t jolt_tile(jolt_ctx* ctx, const jolt_table* base, size_t copies, jolt_table** out) {
    if (!ctx || !base || !out) return JOLT_ERR_INVALID_ARG;
    size_t total = base->len * copies;
    jolt_table* r = nullptr;
    JOLT_TRY(jolt_internal_table_new(ctx, total, &r));
    if (total) hipLaunchKernelGGL(k_tile, dim3(blocks_for(total)), dim3(kBlock), 0, ctx->stream, (const Fr*)base->data(), base->len, total, r->data());
    JOLT_HIP_TRY(ctx, hipGetLastError());
    *out = r;
    return JOLT_OK;
}
extern "C" int32_t jolt_replicate_stream_lsb(jolt_ctx* ctx, const jolt_table* base, jolt_table** out) {
    if (!ctx || !base || !out) return JOLT_ERR_INVALID_ARG;
    size_t total = base->len * 2;
    jolt_table* r = nullptr;
    JOLT_TRY(jolt_internal_table_new(ctx, total, &r));
    if (total) hipLaunchKernelGGL(k_replicate_lsb, dim3(blocks_for(total)), dim3(kBlock), 0, ctx->stream, (const Fr*)base->data(), total, r->data());
    JOLT_HIP_TRY(ctx, hipGetLastError());
    *out = r;
    return JOLT_OK;
}

// joint polynomial sum_i scalars[i] * f_i of a homomorphic batch opening (RlcSource::to_dense)
extern "C" int32_t jolt_rlc(jolt_ctx* ctx, jolt_table* const* tables, size_t k, const jolt_fr_t* scalars, jolt_table** out) {
    if (!ctx || !tables || !scalars || !out || k == 0) return JOLT_ERR_INVALID_ARG;
    if (k > (size_t)kMaxBatchTables) return JOLT_ERR_UNSUPPORTED;
    size_t n = tables[0]->len;
    RlcTables a;
    a.k = (int)k;
    for (size_t j = 0; j < (size_t)kMaxBatchTables; ++j) { a.t[j] = nullptr; a.s[j] = Fr::zero(); a.one[j] = 0; }
    for (size_t j = 0; j < k; ++j) {
        if (!tables[j]) return JOLT_ERR_INVALID_ARG;
        if (tables[j]->len != n) return JOLT_ERR_SIZE_MISMATCH;  // RlcSource::new assert (multilinear.rs:380-383)
        a.t[j] = tables[j]->data();
        a.s[j] = fr_from_abi(&scalars[j]);
        JOLT_REQUIRE(ctx, fr_is_canonical(a.s[j]), "scalar is not a canonical Fr");
        a.one[j] = a.s[j] == Fr::one() ? 1u : 0u;
    }
    jolt_table* r = nullptr;
    JOLT_TRY(jolt_internal_table_new(ctx, n, &r));
    hipLaunchKernelGGL(k_rlc_equal, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream, a, n, r->data());
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { jolt_table_free(ctx, r); ctx->last_error = hipGetErrorString(e); return JOLT_ERR_HIP; }
    *out = r;
    return JOLT_OK;
}

// sum_i a[i] * b[i]: plain field sums (deferred = 0) or the deferred-reduction accumulator (deferred = 1); same canonical value
extern "C" int32_t jolt_table_dot(jolt_ctx* ctx, const jolt_table* a, const jolt_table* b, int32_t deferred, jolt_fr_t* out) {
    if (!ctx || !a || !b || !out) return JOLT_ERR_INVALID_ARG;
    if (a->len != b->len) return JOLT_ERR_SIZE_MISMATCH;
    const int grid = (int)std::max<size_t>(1, std::min<size_t>((a->len + kBlock - 1) / kBlock, (size_t)ctx->num_cus * 8));
    JOLT_TRY(jolt_internal_ensure_scratch(ctx, (size_t)grid + 8, 8));
    if (deferred) hipLaunchKernelGGL(k_dot_wide, dim3(grid), dim3(kBlock), 0, ctx->stream, (const Fr*)a->data(), (const Fr*)b->data(), a->len, ctx->d_partials);
    else hipLaunchKernelGGL(k_sum_or_dot<true>, dim3(grid), dim3(kBlock), 0, ctx->stream, (const Fr*)a->data(), (const Fr*)b->data(), a->len, ctx->d_partials);
    JOLT_HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(kBlock), 0, ctx->stream, (const Fr*)ctx->d_partials, grid, 1, ctx->d_results);
    JOLT_HIP_TRY(ctx, hipGetLastError());
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(ctx->h_results, ctx->d_results, sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(out, ctx->h_results, sizeof(Fr));
    return JOLT_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// SplitLt: LT(., r) + constant served from ~sqrt(T) split tables and bound low-to-high
// (crates/jolt-kernels/src/optimized/support.rs:640-760).  Big-endian index j = j_hi || j_lo, r = r_hi || r_lo:
//   LT(j, r) = LT(j_hi, r_hi) + eq(j_hi, r_hi) * LT(j_lo, r_lo);  the additive constant rides in the hi table; low-to-high binds
// touch lt_lo only; once the lo variables are exhausted the lo scalar folds into the hi table and binding continues densely.
// ------------------------------------------------------------------------------------------------------------------
int32_t jolt_internal_bind(jolt_ctx* ctx, jolt_table* const* tables, size_t k, const Fr& r, int32_t order);
int32_t jolt_internal_table_new(jolt_ctx* ctx, size_t len, jolt_table** out);

struct jolt_split_lt {
    jolt_ctx* ctx = nullptr;
    jolt_table *lt_lo = nullptr, *lt_hi = nullptr, *eq_hi = nullptr;  // split state
    jolt_table* dense = nullptr;                                      // dense state
};

namespace {
static __global__ __launch_bounds__(kBlock) void k_add_constant(Fr* __restrict__ t, size_t n, Fr c) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) st_fr(t + i, add(ld_fr(t + i), c));
}
// out[j] = lt_hi[j / lo_len] + eq_hi[j / lo_len] * lt_lo[j % lo_len]   (lo_len a power of two)
static __global__ __launch_bounds__(kBlock) void k_split_lt_expand(const Fr* __restrict__ lt_lo, const Fr* __restrict__ lt_hi, const Fr* __restrict__ eq_hi,
                                                                   size_t lo_len, size_t total, Fr* __restrict__ out) {
    size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= total) return;
    size_t hi = j / lo_len, lo = j & (lo_len - 1);
    st_fr(out + j, add(ld_fr(lt_hi + hi), mul(ld_fr(eq_hi + hi), ld_fr(lt_lo + lo))));
}
}  // namespace

extern "C" int32_t jolt_split_lt_free(jolt_ctx* ctx, jolt_split_lt* s) {
    if (!s) return JOLT_OK;
    jolt_ctx* c = ctx ? ctx : s->ctx;
    for (jolt_table* t : {s->lt_lo, s->lt_hi, s->eq_hi, s->dense}) if (t) jolt_table_free(c, t);
    delete s;
    return JOLT_OK;
}

// SplitLt::new_plus_constant (support.rs:683-703); constant may be NULL (= 0)
extern "C" int32_t jolt_split_lt_create(jolt_ctx* ctx, const jolt_fr_t* r_cycle, size_t n, const jolt_fr_t* constant, jolt_split_lt** out) {
    if (!ctx || !out || (!r_cycle && n) || n > 40) return JOLT_ERR_INVALID_ARG;
    jolt_split_lt* s = new (std::nothrow) jolt_split_lt();
    if (!s) return JOLT_ERR_OOM;
    s->ctx = ctx;
    const size_t mid = n / 2, hi_len = n - mid;  // r_hi = r[..hi_len], r_lo = r[hi_len..]
    int32_t st = jolt_lt_evals(ctx, r_cycle, hi_len, &s->lt_hi);
    if (st == JOLT_OK && constant) {
        Fr c = fr_from_abi(constant);
        if (!fr_is_canonical(c)) st = JOLT_ERR_INVALID_ARG;
        else {
            size_t len = s->lt_hi->len;
            hipLaunchKernelGGL(k_add_constant, dim3((unsigned)((len + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, s->lt_hi->data(), len, c);
            if (hipGetLastError() != hipSuccess) st = JOLT_ERR_HIP;
        }
    }
    if (st == JOLT_OK && mid == 0) {  // no lo variables: dense from the start
        s->dense = s->lt_hi;
        s->lt_hi = nullptr;
    } else if (st == JOLT_OK) {
        st = jolt_lt_evals(ctx, r_cycle + hi_len, mid, &s->lt_lo);
        if (st == JOLT_OK) st = jolt_eq_evals(ctx, r_cycle, hi_len, nullptr, &s->eq_hi);
    }
    if (st != JOLT_OK) { jolt_split_lt_free(ctx, s); return st; }
    *out = s;
    return JOLT_OK;
}

extern "C" int32_t jolt_split_lt_len(const jolt_split_lt* s, size_t* len) {
    if (!s || !len) return JOLT_ERR_INVALID_ARG;
    *len = s->dense ? s->dense->len : s->lt_hi->len * s->lt_lo->len;
    return JOLT_OK;
}

// SplitLt::bind (support.rs:727-748), LowToHigh
extern "C" int32_t jolt_split_lt_bind(jolt_ctx* ctx, jolt_split_lt* s, const jolt_fr_t* r) {
    if (!ctx || !s || !r) return JOLT_ERR_INVALID_ARG;
    Fr c = fr_from_abi(r);
    JOLT_REQUIRE(ctx, fr_is_canonical(c), "bind challenge is not a canonical Fr");
    if (s->dense) {
        if (s->dense->len < 2) { ctx->last_error = "cannot bind a zero-variable polynomial"; return JOLT_ERR_INVALID_ARG; }
        return jolt_internal_bind(ctx, &s->dense, 1, c, JOLT_ORDER_LOW_TO_HIGH);
    }
    JOLT_TRY(jolt_internal_bind(ctx, &s->lt_lo, 1, c, JOLT_ORDER_LOW_TO_HIGH));
    if (s->lt_lo->len == 1) {  // lo variables exhausted: dense[hi] = lt_hi[hi] + eq_hi[hi] * lo_scalar
        jolt_table* d = nullptr;
        JOLT_TRY(jolt_internal_table_new(ctx, s->lt_hi->len, &d));
        hipLaunchKernelGGL(k_split_lt_expand, dim3((unsigned)((d->len + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, (const Fr*)s->lt_lo->data(),
                           (const Fr*)s->lt_hi->data(), (const Fr*)s->eq_hi->data(), (size_t)1, d->len, d->data());
        if (hipGetLastError() != hipSuccess) { jolt_table_free(ctx, d); return JOLT_ERR_HIP; }
        for (jolt_table* t : {s->lt_lo, s->lt_hi, s->eq_hi}) jolt_table_free(ctx, t);  // back to the pool: reused in stream order
        s->lt_lo = s->lt_hi = s->eq_hi = nullptr;
        s->dense = d;
    }
    return JOLT_OK;
}

// every current evaluation ((LT[2y], LT[2y+1]) = SplitLt::pair for all y, support.rs:705-725) as a dense table
extern "C" int32_t jolt_split_lt_to_dense(jolt_ctx* ctx, const jolt_split_lt* s, jolt_table** out) {
    if (!ctx || !s || !out) return JOLT_ERR_INVALID_ARG;
    if (s->dense) return jolt_table_clone(ctx, s->dense, out);
    const size_t lo_len = s->lt_lo->len, total = lo_len * s->lt_hi->len;
    jolt_table* d = nullptr;
    JOLT_TRY(jolt_internal_table_new(ctx, total, &d));
    hipLaunchKernelGGL(k_split_lt_expand, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, (const Fr*)s->lt_lo->data(),
                       (const Fr*)s->lt_hi->data(), (const Fr*)s->eq_hi->data(), lo_len, total, d->data());
    if (hipGetLastError() != hipSuccess) { jolt_table_free(ctx, d); return JOLT_ERR_HIP; }
    *out = d;
    return JOLT_OK;
}

// SplitLt::final_value (support.rs:750-758): defined once fully bound (always dense by then)
extern "C" int32_t jolt_split_lt_final_value(jolt_ctx* ctx, const jolt_split_lt* s, jolt_fr_t* out) {
    if (!ctx || !s || !out) return JOLT_ERR_INVALID_ARG;
    if (!s->dense || s->dense->len != 1) return JOLT_ERR_NOT_FULLY_BOUND;
    return jolt_table_download(ctx, s->dense, 0, 1, out);
}
