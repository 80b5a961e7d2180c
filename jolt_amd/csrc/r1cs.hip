// jolt_amd/csrc/r1cs.hip -- the T-scale sums of the Spartan outer (stage 1) kernels on the device.
//
// What touches every cycle in crates/jolt-kernels/src/{reference,optimized}/spartan_outer.rs:
//   uniskip_first_round_poly (reference :172-221): t1(node) = sum_t sum_s eq[(t << 1) | s] * Az(node,s,t) * Bz(node,s,t) for the 9
//     extended row nodes -- here with the per-(node, stream) row weights already folded into per-COLUMN weights by the caller
//     (ConstraintMatrices::weighted_columns / public_column_contributions, :246-256), so Az(node,s,t) = c0 + sum_v c_v * z_v(t);
//   the remainder member's Az / Bz (reference :236-300; the optimized tier's fused round-0 materialisation, optimized :33-36):
//     the two linear forms over the joint (cycle || stream) domain, written once and then summed by the split-eq PRODUCT member
//     (jolt_member_create_split_eq_product over tau_low with the Lagrange kernel value as scale) -- no per-relation round kernel;
//   the post-hoc opening evaluation (optimized :41-43): z_v(r_cycle) for all 35 inputs from ONE eq table (jolt_tables_evaluate).
// The constraint list (crates/jolt-r1cs/src/constraints/jolt.rs), spartan_outer_row_weights and the Lagrange interpolation are O(rows)
// host work and stay in Rust: the device sees column weights.  Multiplies per cycle: 4 per input and node for the uni-skip sums
// (the reference's optimized tier does this part in integer arithmetic on typed small-scalar rows: the known next lever here).
#include <algorithm>
#include <vector>

#include "ctx.hpp"
#include "poly_kernels.hip.h"

using namespace jolt;

int32_t jolt_internal_table_new(jolt_ctx* ctx, size_t len, jolt_table** out);
int32_t jolt_internal_ensure_scratch(jolt_ctx* ctx, size_t partials, size_t results);

namespace {

constexpr int kMaxR1csInputs = 64;
struct R1csInputs {
    const Fr* z[kMaxR1csInputs];
    int n;
};

// partials[node * gridDim.x + block] = this block's share of t1(node); weights: [node][stream][1 + n] (wave-uniform reads)
__global__ __launch_bounds__(kBlock) void k_r1cs_uniskip(R1csInputs in, const Fr* __restrict__ eq, size_t cycles, const Fr* __restrict__ wa, const Fr* __restrict__ wb,
                                                         Fr* __restrict__ partials) {
    const size_t node = blockIdx.y, stride_w = 1 + (size_t)in.n;
    const Fr* a0 = wa + (node * 2) * stride_w;
    const Fr* a1 = a0 + stride_w;
    const Fr* b0 = wb + (node * 2) * stride_w;
    const Fr* b1 = b0 + stride_w;
    Fr acc[1] = {Fr::zero()};
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < cycles; t += stride) {
        Fr az0 = ld_fr(a0), az1 = ld_fr(a1), bz0 = ld_fr(b0), bz1 = ld_fr(b1);
        for (int v = 0; v < in.n; ++v) {
            const Fr z = ld_fr(in.z[v] + t);
            az0 = add(az0, mul(ld_fr(a0 + 1 + v), z));
            az1 = add(az1, mul(ld_fr(a1 + 1 + v), z));
            bz0 = add(bz0, mul(ld_fr(b0 + 1 + v), z));
            bz1 = add(bz1, mul(ld_fr(b1 + 1 + v), z));
        }
        acc[0] = add(acc[0], add(mul(ld_fr(eq + 2 * t), mul(az0, bz0)), mul(ld_fr(eq + 2 * t + 1), mul(az1, bz1))));
    }
    block_reduce_store<1>(acc, partials + node * gridDim.x);
}

// az[(t << 1) | s] = wa[s][0] + sum_v wa[s][1 + v] * z_v(t); likewise bz
__global__ __launch_bounds__(kBlock) void k_r1cs_materialize(R1csInputs in, size_t cycles, const Fr* __restrict__ wa, const Fr* __restrict__ wb, Fr* __restrict__ az,
                                                             Fr* __restrict__ bz) {
    const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= cycles) return;
    const size_t stride_w = 1 + (size_t)in.n;
    Fr az0 = ld_fr(wa), az1 = ld_fr(wa + stride_w), bz0 = ld_fr(wb), bz1 = ld_fr(wb + stride_w);
    for (int v = 0; v < in.n; ++v) {
        const Fr z = ld_fr(in.z[v] + t);
        az0 = add(az0, mul(ld_fr(wa + 1 + v), z));
        az1 = add(az1, mul(ld_fr(wa + stride_w + 1 + v), z));
        bz0 = add(bz0, mul(ld_fr(wb + 1 + v), z));
        bz1 = add(bz1, mul(ld_fr(wb + stride_w + 1 + v), z));
    }
    st_fr(az + 2 * t, az0);
    st_fr(az + 2 * t + 1, az1);
    st_fr(bz + 2 * t, bz0);
    st_fr(bz + 2 * t + 1, bz1);
}

// partials[k * gridDim.x + block] = this block's share of sum_t eq[t] * table_k[t]
__global__ __launch_bounds__(kBlock) void k_tables_dot_eq(R1csInputs in, const Fr* __restrict__ eq, size_t len, Fr* __restrict__ partials) {
    const Fr* z = in.z[blockIdx.y];
    Fr acc[1] = {Fr::zero()};
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < len; t += stride) acc[0] = add(acc[0], mul(ld_fr(eq + t), ld_fr(z + t)));
    block_reduce_store<1>(acc, partials + (size_t)blockIdx.y * gridDim.x);
}

int32_t gather_inputs(jolt_ctx* ctx, jolt_table* const* inputs, size_t n_inputs, R1csInputs* out, size_t* cycles) {
    if (n_inputs == 0 || n_inputs > (size_t)kMaxR1csInputs) return JOLT_ERR_UNSUPPORTED;
    out->n = (int)n_inputs;
    for (size_t v = 0; v < (size_t)kMaxR1csInputs; ++v) out->z[v] = nullptr;
    for (size_t v = 0; v < n_inputs; ++v) {
        if (!inputs[v]) return JOLT_ERR_INVALID_ARG;
        if (inputs[v]->len != inputs[0]->len) return JOLT_ERR_SIZE_MISMATCH;
        out->z[v] = inputs[v]->data();
    }
    *cycles = inputs[0]->len;
    return JOLT_OK;
}

// weights from the caller's (possibly short-lived) host array into a pool block; canonical check on the way
int32_t upload_weights(jolt_ctx* ctx, const jolt_fr_t* w, size_t count, Fr** out) {
    for (size_t i = 0; i < count; ++i) JOLT_REQUIRE(ctx, fr_is_canonical(fr_from_abi(&w[i])), "weight is not a canonical Fr");
    JOLT_TRY(jolt_internal_dev_alloc(ctx, count * sizeof(Fr), (void**)out));
    hipError_t e = hipMemcpyAsync(*out, w, count * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { jolt_internal_dev_free(ctx, *out); *out = nullptr; ctx->last_error = hipGetErrorString(e); return JOLT_ERR_HIP; }
    return JOLT_OK;
}

// results[slot + k] = sum of row k of partials (k rows of nblocks entries), then to the host
int32_t reduce_rows_to_host(jolt_ctx* ctx, size_t rows, int nblocks, jolt_fr_t* out) {
    for (size_t k = 0; k < rows; ++k) {
        hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(kBlock), 0, ctx->stream, (const Fr*)(ctx->d_partials + k * (size_t)nblocks), nblocks, 1, ctx->d_results + k);
    }
    JOLT_HIP_TRY(ctx, hipGetLastError());
    JOLT_HIP_TRY(ctx, hipMemcpyAsync(ctx->h_results, ctx->d_results, rows * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    JOLT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(out, ctx->h_results, rows * sizeof(Fr));
    return JOLT_OK;
}

}  // namespace

extern "C" int32_t jolt_r1cs_uniskip_sums(jolt_ctx* ctx, jolt_table* const* inputs, size_t n_inputs, const jolt_table* eq, const jolt_fr_t* a_weights,
                                          const jolt_fr_t* b_weights, size_t n_nodes, jolt_fr_t* out) {
    if (!ctx || !inputs || !eq || !a_weights || !b_weights || !out || n_nodes == 0 || n_nodes > 64) return JOLT_ERR_INVALID_ARG;
    R1csInputs in;
    size_t cycles = 0;
    JOLT_TRY(gather_inputs(ctx, inputs, n_inputs, &in, &cycles));
    if (eq->len != 2 * cycles) return JOLT_ERR_SIZE_MISMATCH;
    const size_t wcount = n_nodes * 2 * (1 + n_inputs);
    Fr *wa = nullptr, *wb = nullptr;
    JOLT_TRY(upload_weights(ctx, a_weights, wcount, &wa));
    int32_t s = upload_weights(ctx, b_weights, wcount, &wb);
    if (s != JOLT_OK) { jolt_internal_dev_free(ctx, wa); return s; }
    const int grid = (int)std::max<size_t>(1, std::min<size_t>((cycles + kBlock - 1) / kBlock, (size_t)ctx->num_cus * 4));
    s = jolt_internal_ensure_scratch(ctx, n_nodes * (size_t)grid + 8, n_nodes + 8);
    if (s == JOLT_OK) {
        hipLaunchKernelGGL(k_r1cs_uniskip, dim3(grid, (unsigned)n_nodes), dim3(kBlock), 0, ctx->stream, in, (const Fr*)eq->data(), cycles, (const Fr*)wa, (const Fr*)wb,
                           ctx->d_partials);
        s = hipGetLastError() == hipSuccess ? JOLT_OK : JOLT_ERR_HIP;
    }
    if (s == JOLT_OK) s = reduce_rows_to_host(ctx, n_nodes, grid, out);
    jolt_internal_dev_free(ctx, wa);
    jolt_internal_dev_free(ctx, wb);
    return s;
}

extern "C" int32_t jolt_r1cs_materialize(jolt_ctx* ctx, jolt_table* const* inputs, size_t n_inputs, const jolt_fr_t* a_weights, const jolt_fr_t* b_weights,
                                         jolt_table** az_out, jolt_table** bz_out) {
    if (!ctx || !inputs || !a_weights || !b_weights || !az_out || !bz_out) return JOLT_ERR_INVALID_ARG;
    R1csInputs in;
    size_t cycles = 0;
    JOLT_TRY(gather_inputs(ctx, inputs, n_inputs, &in, &cycles));
    const size_t wcount = 2 * (1 + n_inputs);
    Fr *wa = nullptr, *wb = nullptr;
    jolt_table *az = nullptr, *bz = nullptr;
    int32_t s = upload_weights(ctx, a_weights, wcount, &wa);
    if (s == JOLT_OK) s = upload_weights(ctx, b_weights, wcount, &wb);
    if (s == JOLT_OK) s = jolt_internal_table_new(ctx, 2 * cycles, &az);
    if (s == JOLT_OK) s = jolt_internal_table_new(ctx, 2 * cycles, &bz);
    if (s == JOLT_OK) {
        hipLaunchKernelGGL(k_r1cs_materialize, dim3((unsigned)((cycles + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, in, cycles, (const Fr*)wa, (const Fr*)wb,
                           az->data(), bz->data());
        if (hipGetLastError() != hipSuccess) s = JOLT_ERR_HIP;
    }
    if (wa) jolt_internal_dev_free(ctx, wa);
    if (wb) jolt_internal_dev_free(ctx, wb);
    if (s != JOLT_OK) {
        if (az) jolt_table_free(ctx, az);
        if (bz) jolt_table_free(ctx, bz);
        return s;
    }
    *az_out = az;
    *bz_out = bz;
    return JOLT_OK;
}

// out[k] = Polynomial::evaluate(tables[k], point) for k tables of 2^n entries sharing ONE eq expansion
extern "C" int32_t jolt_tables_evaluate(jolt_ctx* ctx, jolt_table* const* tables, size_t k, const jolt_fr_t* point, size_t n, jolt_fr_t* out) {
    if (!ctx || !tables || !out || (!point && n)) return JOLT_ERR_INVALID_ARG;
    R1csInputs in;
    size_t len = 0;
    JOLT_TRY(gather_inputs(ctx, tables, k, &in, &len));
    if (len != ((size_t)1 << n)) return JOLT_ERR_SIZE_MISMATCH;  // dense.rs:341-345 assert
    jolt_table* eq = nullptr;
    JOLT_TRY(jolt_eq_evals(ctx, point, n, nullptr, &eq));
    const int grid = (int)std::max<size_t>(1, std::min<size_t>((len + kBlock - 1) / kBlock, (size_t)ctx->num_cus * 2));
    int32_t s = jolt_internal_ensure_scratch(ctx, k * (size_t)grid + 8, k + 8);
    if (s == JOLT_OK) {
        hipLaunchKernelGGL(k_tables_dot_eq, dim3(grid, (unsigned)k), dim3(kBlock), 0, ctx->stream, in, (const Fr*)eq->data(), len, ctx->d_partials);
        s = hipGetLastError() == hipSuccess ? JOLT_OK : JOLT_ERR_HIP;
    }
    if (s == JOLT_OK) s = reduce_rows_to_host(ctx, k, grid, out);
    jolt_table_free(ctx, eq);
    return s;
}
