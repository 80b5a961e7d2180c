// jolt_amd/csrc/views.hip -- derived-table builders the dense members need and the joint
// polynomial of a homomorphic batch opening (a12).  All are single streaming passes (HBM bound, <= 1 multiply per entry).
//   address_fold / cycle_fold / tile / replicate_stream_lsb   crates/jolt-kernels/src/reference/views.rs:35-138
//   RlcSource::to_dense (sum_i s_i f_i)                        crates/jolt-poly/src/multilinear.rs:159-170,358-464,
//                                                              HomomorphicBatch::prove_batch crates/jolt-openings/src/schemes.rs:487-524
#include <algorithm>

#include "ctx.hpp"
#include "poly_kernels.cuh"

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

struct RlcTables {
    const Fr* t[kMaxBatchTables];
    int k;
};
static __global__ __launch_bounds__(kBlock) void k_rlc_equal(RlcTables a, const Fr* __restrict__ scalars, const uint32_t* __restrict__ is_one, size_t n,
                                                             Fr* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Fr acc = Fr::zero();
    for (int j = 0; j < a.k; ++j) {
        Fr v = ld_fr(a.t[j] + i);
        if (!is_one[j]) v = mul(v, ld_fr(scalars + j));
        acc = add(acc, v);
    }
    st_fr(out + i, acc);
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
    std::vector<uint32_t> ones(k);
    for (size_t j = 0; j < k; ++j) {
        if (!tables[j]) return JOLT_ERR_INVALID_ARG;
        if (tables[j]->len != n) return JOLT_ERR_SIZE_MISMATCH;  // RlcSource::new assert (multilinear.rs:380-383)
        a.t[j] = tables[j]->data();
        Fr s = fr_from_abi(&scalars[j]);
        JOLT_REQUIRE(ctx, fr_is_canonical(s), "scalar is not a canonical Fr");
        ones[j] = s == Fr::one() ? 1u : 0u;
    }
    jolt_table *r = nullptr, *ds = nullptr;
    uint32_t* d_one = nullptr;
    JOLT_TRY(jolt_internal_table_new(ctx, n, &r));
    int32_t st = jolt_table_upload(ctx, scalars, k, &ds);
    hipError_t e = st == JOLT_OK ? hipMalloc((void**)&d_one, k * sizeof(uint32_t)) : hipErrorUnknown;
    if (e == hipSuccess) e = hipMemcpyAsync(d_one, ones.data(), k * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_rlc_equal, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream, a, (const Fr*)ds->data(), (const uint32_t*)d_one, n, r->data());
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (d_one) (void)hipFree(d_one);
    if (ds) jolt_table_free(ctx, ds);
    if (e != hipSuccess) { jolt_table_free(ctx, r); ctx->last_error = hipGetErrorString(e); return st != JOLT_OK ? st : JOLT_ERR_HIP; }
    *out = r;
    return JOLT_OK;
}
