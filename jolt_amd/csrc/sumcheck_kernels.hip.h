// jolt_amd/csrc/sumcheck_kernels.hip.h -- round-polynomial accumulation kernels.
//
// One kernel family serves every dense sumcheck member.  The summand is held in "sum of products of linear
// combinations" form
//        summand(x) = sum_g  prod_{f in group g} ( const_f + sum_k coeff_k * table_k(x) )
// which contains the reference tier's flat jolt_claims::Expr (every LC has one entry; crates/jolt-claims/src/claims.rs:
// 17-46, evaluated as in crates/jolt-kernels/src/reference/naive.rs:241-310) and the optimized tier's fused forms
// (linear-leaf fusion, crates/jolt-kernels/src/optimized/inc_claim_reduction.rs:68-89) as descriptor rewrites.
// Because every LC is linear in the round variable, a factor's values at t = 0..NE-1 are lo + t*(hi - lo): the
// multiplies per pair are (entries with coeff != 1)*2 for the LCs plus (factors-1)*NE for the products.
//
// Work is ALU-bound on v_mad_u64_u32 for all but the smallest summands; HBM traffic is 64 B per table per pair.
#pragma once
#include "desc.hpp"
#include "poly_kernels.hip.h"

namespace jolt {

// s(t) partial sums for t = 0..NE-1 (or, with SKIP1, for t in {0,2,3,..,NE}: slot k>=1 holds s(k+1) -- the
// optimized tier's skipped-evals form, crates/jolt-kernels/src/optimized/support.rs:450-459).
// ORDER 0: LowToHigh pairs (2y, 2y+1); ORDER 1: HighToLow pairs (y, y+half).
// lo + r*(hi-lo) with the challenge shape decided at run time (wave-uniform branch)
__device__ __forceinline__ Fr bind_pair_rt(const Fr& lo, const Fr& hi, const Fr& r, bool shifted) {
    Fr d = sub(hi, lo);
    Fr m;
    if (shifted) {
        uint32_t chi[4] = {r.l[4], r.l[5], r.l[6], r.l[7]};
        m = mul_shifted(d, chi);
    } else {
        m = mul(d, r);
    }
    return add(lo, m);
}
// The (lo, hi) pair of table `tp` at pair index y.  FUSED (LowToHigh only): the table has NOT been bound with the
// previous challenge yet -- read the four entries 4y..4y+3, bind them on the fly (the fused contract of
// ProveRounds::prove_round, crates/jolt-sumcheck/src/prover.rs:45-51) and, for the table's owner entry, store the two
// bound values so that the next round finds the bound table: one pass over memory per round instead of a bind-write
// pass followed by an evaluate-read pass.
template <int ORDER, bool FUSED>
__device__ __forceinline__ void load_pair(const Fr* __restrict__ tp, Fr* __restrict__ op, bool owner, size_t y, size_t half, const Fr& r, bool shifted,
                                          Fr& lo, Fr& hi) {
    if constexpr (FUSED) {
        Fr a0 = ld_fr(tp + 4 * y), a1 = ld_fr(tp + 4 * y + 1), a2 = ld_fr(tp + 4 * y + 2), a3 = ld_fr(tp + 4 * y + 3);
        lo = bind_pair_rt(a0, a1, r, shifted);
        hi = bind_pair_rt(a2, a3, r, shifted);
        if (owner) { st_fr(op + 2 * y, lo); st_fr(op + 2 * y + 1, hi); }
    } else if constexpr (ORDER == 0) {
        lo = ld_fr(tp + 2 * y);
        hi = ld_fr(tp + 2 * y + 1);
    } else {
        lo = ld_fr(tp + y);
        hi = ld_fr(tp + y + half);
    }
}

// e_out != nullptr: eq-weighted member -- the product values of pair y are multiplied by E_out[y >> in_bits] * E_in[y & mask]
// (GruenSplitEqPolynomial::par_fold_out_in's row weight, crates/jolt-poly/src/split_eq.rs:449-512) and `d` is the inner summand.
template <int NE, int ORDER, bool SKIP1, bool FUSED>
__device__ __forceinline__ void round_evals_body(const MemberDesc* __restrict__ d, const Fr* const* __restrict__ tabs, Fr* const* __restrict__ outs,
                                                 size_t half, const Fr& r, bool shifted, Fr* __restrict__ partials, const Fr* __restrict__ e_out = nullptr,
                                                 const Fr* __restrict__ e_in = nullptr, int in_bits = 0) {
    Fr acc[NE];
#pragma unroll
    for (int t = 0; t < NE; ++t) acc[t] = Fr::zero();
    // Work item = (group g, pair y), g slowest: a wave shares g (descriptor reads stay scalar) and walks consecutive pairs
    // (same coalescing as a per-pair sweep), while a summand with many groups exposes groups x pairs parallelism --
    // per-pair work of a big summand is a serial chain of >100 multiplies, which left mid-size rounds latency bound.
    const uint32_t n_groups = d->n_groups;
    const size_t items = half * n_groups;
    size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < items; i += stride) {
        const uint32_t g = (uint32_t)(i / half);
        const size_t y = i - (size_t)g * half;
        Fr prod[NE];
        const uint32_t f0 = d->grp_fac_off[g], f1 = d->grp_fac_off[g + 1];
        for (uint32_t f = f0; f < f1; ++f) {
            Fr lo, hi;
            if (d->fac_has_const[f]) { lo = d->fac_const[f]; hi = lo; }
            else { lo = Fr::zero(); hi = lo; }
            const uint32_t k0 = d->fac_lc_off[f], k1 = d->fac_lc_off[f + 1];
            for (uint32_t k = k0; k < k1; ++k) {
                const uint32_t ti = d->lc_tab[k];
                Fr a, b;
                load_pair<ORDER, FUSED>(tabs[ti], FUSED ? outs[ti] : nullptr, FUSED && d->lc_owner[k], y, half, r, shifted, a, b);
                if (!d->lc_one[k]) {
                    Fr c = d->lc_coeff[k];
                    a = mul(a, c);
                    b = mul(b, c);
                }
                lo = add(lo, a);
                hi = add(hi, b);
            }
            Fr step = sub(hi, lo);
            Fr v = lo;
            if (f == f0) {
                prod[0] = v;
                if constexpr (SKIP1) v = add(v, step);
#pragma unroll
                for (int t = 1; t < NE; ++t) { v = add(v, step); prod[t] = v; }
            } else {
                prod[0] = mul(prod[0], v);
                if constexpr (SKIP1) v = add(v, step);
#pragma unroll
                for (int t = 1; t < NE; ++t) { v = add(v, step); prod[t] = mul(prod[t], v); }
            }
        }
        if (f1 == f0) {  // empty product: the constant one
#pragma unroll
            for (int t = 0; t < NE; ++t) prod[t] = Fr::one();
        }
        if (e_out) {
            const Fr e = mul(ld_fr(e_out + (y >> in_bits)), ld_fr(e_in + (y & (((size_t)1 << in_bits) - 1))));
#pragma unroll
            for (int t = 0; t < NE; ++t) prod[t] = mul(prod[t], e);
        }
#pragma unroll
        for (int t = 0; t < NE; ++t) acc[t] = add(acc[t], prod[t]);
    }
    block_reduce_store<NE>(acc, partials);
}

// All members of one batch round that share (NE, ORDER, SKIP1) in ONE launch: blockIdx.y selects the member.
// (Launch count per batch round drops from 3 per member to ~5 per stage; SURVEY.md section 7 "member group".)
constexpr int kMaxGroupMembers = 16;
constexpr int kMaxGroupTables = 96;
struct RoundGroupArgs {
    const Fr* tabs[kMaxGroupTables];  // flat: member m uses tabs[tab_off[m] ...]
    Fr* outs[kMaxGroupTables];        // fused rounds: where the bound tables go
    const MemberDesc* desc[kMaxGroupMembers];
    size_t half[kMaxGroupMembers];
    uint32_t tab_off[kMaxGroupMembers];
    uint32_t part_off[kMaxGroupMembers];  // offset (in Fr) of the member's partial sums
    uint32_t ticket[kMaxGroupMembers];    // per-member ticket counter index
    uint32_t slot[kMaxGroupMembers];      // result slot of the member's first sum
    const Fr* e_out[kMaxGroupMembers];    // eq-weighted members: E tables in force this round (null otherwise)
    const Fr* e_in[kMaxGroupMembers];
    int32_t in_bits[kMaxGroupMembers];
};
template <int NE, int ORDER, bool SKIP1, bool FUSED>
static __global__ __launch_bounds__(kBlock) void k_round_evals_group(RoundGroupArgs a, Fr r, int shifted, Fr* __restrict__ partials, RoundDone rd) {
    const int m = blockIdx.y;
    round_evals_body<NE, ORDER, SKIP1, FUSED>(a.desc[m], a.tabs + a.tab_off[m], a.outs + a.tab_off[m], a.half[m], r, shifted != 0,
                                              partials + a.part_off[m], a.e_out[m], a.e_in[m], a.in_bits[m]);
    finish_member(partials + a.part_off[m], NE, a.ticket[m], a.slot[m], rd);
}

// Tail rounds (few pairs left): the per-pair work of a big summand is a serial chain of ~100+ multiplies, which made a
// round cost ~50-230 us however small the tables were.  Here every (pair, group, evaluation point) is its own work
// item and ALL expr members of the round share one launch (blockIdx.y = member, blockIdx.z = evaluation point), so
// the floor is one short chain (~ #factors multiplies) instead of the sum over classes.
struct TailArgs {
    RoundGroupArgs g;
    uint32_t ne[kMaxGroupMembers];
    uint32_t order[kMaxGroupMembers];
    uint32_t skip[kMaxGroupMembers];
    uint32_t fused[kMaxGroupMembers];  // bind the previous challenge on the fly (LowToHigh members only)
};
static __global__ __launch_bounds__(kBlock) void k_round_evals_tail(TailArgs a, Fr r, int shifted, Fr* __restrict__ partials, RoundDone rd) {
    const int m = blockIdx.y;
    const uint32_t t = blockIdx.z, ne = a.ne[m];
    if (t >= ne) return;
    const MemberDesc* __restrict__ d = a.g.desc[m];
    const Fr* const* __restrict__ tabs = a.g.tabs + a.g.tab_off[m];
    const size_t half = a.g.half[m];
    const uint32_t ng = d->n_groups;
    const bool l2h = a.order[m] == 0;
    const bool fused = a.fused[m] != 0;
    Fr* const* __restrict__ outs = a.g.outs + a.g.tab_off[m];
    const uint32_t point = (a.skip[m] && t >= 1) ? t + 1 : t;
    Fr acc[1] = {Fr::zero()};
    const size_t items = half * ng;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < items; i += stride) {
        const uint32_t g = (uint32_t)(i / half);  // g slowest: lanes of a wave mostly share the group
        const size_t y = i - (size_t)g * half;
        const size_t i_lo = l2h ? 2 * y : y, i_hi = l2h ? 2 * y + 1 : y + half;
        Fr prod = Fr::one();
        const uint32_t f0 = d->grp_fac_off[g], f1 = d->grp_fac_off[g + 1];
        for (uint32_t f = f0; f < f1; ++f) {
            Fr lo = d->fac_has_const[f] ? d->fac_const[f] : Fr::zero(), hi = lo;
            for (uint32_t k = d->fac_lc_off[f]; k < d->fac_lc_off[f + 1]; ++k) {
                const uint32_t ti = d->lc_tab[k];
                const Fr* __restrict__ tp = tabs[ti];
                Fr x, z;
                if (fused) {
                    load_pair<0, true>(tp, outs[ti], d->lc_owner[k] && t == 0, y, half, r, shifted != 0, x, z);
                } else {
                    x = ld_fr(tp + i_lo);
                    z = ld_fr(tp + i_hi);
                }
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
        if (a.g.e_out[m]) {  // eq-weighted member
            const int ib = a.g.in_bits[m];
            prod = mul(prod, mul(ld_fr(a.g.e_out[m] + (y >> ib)), ld_fr(a.g.e_in[m] + (y & (((size_t)1 << ib) - 1)))));
        }
        acc[0] = add(acc[0], prod);
    }
    Fr* mine = partials + a.g.part_off[m];
    block_reduce_store<1>(acc, mine + (size_t)t * gridDim.x);
    finish_member<true>(mine, (int)ne, a.g.ticket[m], a.g.slot[m], rd, (int)gridDim.x, gridDim.x * ne, t * gridDim.x + blockIdx.x);
}

// Split-eq product member (a6): q(0) = sum_rows E_out[x_out] E_in[x_in] a_lo b_lo,
// q(inf) = sum_rows E_out E_in (a_hi-a_lo)(b_hi-b_lo), row = (x_out << in_bits) | x_in over LowToHigh pairs
// (crates/jolt-kernels/src/optimized/support.rs:391-411 over crates/jolt-poly/src/split_eq.rs:449-512).
// eq is never materialised at size N: E_out, E_in are ~sqrt(N) tables that stay cache-resident.
template <bool FUSED>
static __global__ __launch_bounds__(kBlock) void k_split_eq_product(const Fr* __restrict__ a, const Fr* __restrict__ b, Fr* __restrict__ a_out,
                                                            Fr* __restrict__ b_out, Fr r, int shifted, const Fr* __restrict__ e_out,
                                                            const Fr* __restrict__ e_in, int in_bits, size_t rows, Fr* __restrict__ partials,
                                                            uint32_t ticket, uint32_t slot, RoundDone rd) {
    Fr acc[2] = {Fr::zero(), Fr::zero()};
    size_t stride = (size_t)gridDim.x * kBlock;
    size_t mask = ((size_t)1 << in_bits) - 1;
    for (size_t row = (size_t)blockIdx.x * kBlock + threadIdx.x; row < rows; row += stride) {
        Fr e = mul(ld_fr(e_out + (row >> in_bits)), ld_fr(e_in + (row & mask)));
        Fr a_lo, a_hi, b_lo, b_hi;
        load_pair<0, FUSED>(a, a_out, true, row, rows, r, shifted != 0, a_lo, a_hi);
        load_pair<0, FUSED>(b, b_out, true, row, rows, r, shifted != 0, b_lo, b_hi);
        acc[0] = add(acc[0], mul(e, mul(a_lo, b_lo)));
        acc[1] = add(acc[1], mul(e, mul(sub(a_hi, a_lo), sub(b_hi, b_lo))));
    }
    block_reduce_store<2>(acc, partials);
    finish_member(partials, 2, ticket, slot, rd);
}

// ---------------------------------------------------------------------------------------------------------------------
// Split-eq "uniform product" member:  eq(w, j) * sum_{v<V} c_v * prod_{i<F} T_{v,i}(j)   (degree F+1)
// -- the shape of instruction_ra_virtualization (V = 8, F = 4), ram_ra_virtualization (V = 1, F = d) and
// ram_hamming_booleanity (V = 1, F = 2) in SURVEY.md section 8 a13, the optimized tier's split-eq form
// (crates/jolt-kernels/src/optimized/ram_hamming_booleanity.rs:111-135, crates/jolt-poly/src/split_eq.rs:449-512).
// eq never enters the per-pair product: the kernel returns the eq-stripped q(t) = sum_rows E_out E_in sum_v c_v prod_i T(t)
// at t in {0, 2, .., F} and the host restores s(t) = l(t) q(t) from the running claim.  F is a compile-time constant, so
// the F pair loads of an item are issued back to back (memory-level parallelism the generic interpreter cannot get) and
// the product of F linear factors is built as a tree on evaluation points (finite-difference extension costs only adds):
// F = 4 takes 10 multiplies per item instead of 15.
// ---------------------------------------------------------------------------------------------------------------------
struct UniformArgs {
    const Fr* tabs[kMaxBatchTables];  // V*F tables, v-major
    Fr coeff[kMaxGroups];             // c_v
    uint32_t coeff_one[kMaxGroups];
    int V;
};
template <int F>
__device__ __forceinline__ void uniform_item(const Fr (&lo)[F], const Fr (&hi)[F], Fr (&q)[F]);
template <>
__device__ __forceinline__ void uniform_item<2>(const Fr (&lo)[2], const Fr (&hi)[2], Fr (&q)[2]) {
    Fr a2 = add(hi[0], sub(hi[0], lo[0])), b2 = add(hi[1], sub(hi[1], lo[1]));
    q[0] = mul(lo[0], lo[1]);
    q[1] = mul(a2, b2);  // t = 2
}
template <>
__device__ __forceinline__ void uniform_item<3>(const Fr (&lo)[3], const Fr (&hi)[3], Fr (&q)[3]) {
    Fr s0 = sub(hi[0], lo[0]), s1 = sub(hi[1], lo[1]), s2 = sub(hi[2], lo[2]);
    Fr a2 = add(hi[0], s0), b2 = add(hi[1], s1), c2 = add(hi[2], s2);
    Fr a3 = add(a2, s0), b3 = add(b2, s1), c3 = add(c2, s2);
    q[0] = mul(mul(lo[0], lo[1]), lo[2]);
    q[1] = mul(mul(a2, b2), c2);
    q[2] = mul(mul(a3, b3), c3);
}
// q = A*B on {0,2,3,4} from the quadratic halves A = f0*f1, B = f2*f3 given on {0,1,2} (extended to {3,4} by second differences)
__device__ __forceinline__ void uniform_quadratic_halves(const Fr& A0, const Fr& A1, const Fr& A2, const Fr& B0, const Fr& B1, const Fr& B2, Fr (&q)[4]) {
    Fr dA = sub(A2, A1), ddA = sub(dA, sub(A1, A0));
    Fr dB = sub(B2, B1), ddB = sub(dB, sub(B1, B0));
    Fr dA3 = add(dA, ddA), A3 = add(A2, dA3), A4 = add(A3, add(dA3, ddA));
    Fr dB3 = add(dB, ddB), B3 = add(B2, dB3), B4 = add(B3, add(dB3, ddB));
    q[0] = mul(A0, B0);
    q[1] = mul(A2, B2);
    q[2] = mul(A3, B3);
    q[3] = mul(A4, B4);
}
template <>
__device__ __forceinline__ void uniform_item<4>(const Fr (&lo)[4], const Fr (&hi)[4], Fr (&q)[4]) {
    // A = f0*f1, B = f2*f3 as quadratics on {0,1,2}; extend both to {3,4} by second differences; q = A*B on {0,2,3,4}
    Fr v2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v2[i] = add(hi[i], sub(hi[i], lo[i]));
    Fr A0 = mul(lo[0], lo[1]), A1 = mul(hi[0], hi[1]), A2 = mul(v2[0], v2[1]);
    Fr B0 = mul(lo[2], lo[3]), B1 = mul(hi[2], hi[3]), B2 = mul(v2[2], v2[3]);
    Fr dA = sub(A2, A1), ddA = sub(dA, sub(A1, A0));
    Fr dB = sub(B2, B1), ddB = sub(dB, sub(B1, B0));
    Fr dA3 = add(dA, ddA), A3 = add(A2, dA3), A4 = add(A3, add(dA3, ddA));
    Fr dB3 = add(dB, ddB), B3 = add(B2, dB3), B4 = add(B3, add(dB3, ddB));
    q[0] = mul(A0, B0);
    q[1] = mul(A2, B2);
    q[2] = mul(A3, B3);
    q[3] = mul(A4, B4);
}

template <int F>
static __global__ __launch_bounds__(kBlock) void k_split_eq_uniform(UniformArgs a, const Fr* __restrict__ e_out, const Fr* __restrict__ e_in, int in_bits,
                                                                    size_t rows, Fr* __restrict__ partials, uint32_t ticket, uint32_t slot, RoundDone rd) {
    Fr acc[F];
#pragma unroll
    for (int t = 0; t < F; ++t) acc[t] = Fr::zero();
    const size_t items = rows * (size_t)a.V;
    const size_t mask = ((size_t)1 << in_bits) - 1;
    size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < items; i += stride) {
        const uint32_t v = (uint32_t)(i / rows);  // v slowest: a wave shares v
        const size_t row = i - (size_t)v * rows;
        Fr lo[F], hi[F];
#pragma unroll
        for (int k = 0; k < F; ++k) {  // all F pair loads in flight before the first multiply
            const Fr* __restrict__ tp = a.tabs[v * F + k];
            lo[k] = ld_fr(tp + 2 * row);
            hi[k] = ld_fr(tp + 2 * row + 1);
        }
        Fr w = mul(ld_fr(e_out + (row >> in_bits)), ld_fr(e_in + (row & mask)));
        if (!a.coeff_one[v]) w = mul(w, a.coeff[v]);
        lo[0] = mul(lo[0], w);
        hi[0] = mul(hi[0], w);
        Fr q[F];
        uniform_item<F>(lo, hi, q);
#pragma unroll
        for (int t = 0; t < F; ++t) acc[t] = add(acc[t], q[t]);
    }
    block_reduce_store<F>(acc, partials);
    finish_member(partials, F, ticket, slot, rd);
}

// Row-major form for the big rounds: one work item per pair, the V products inside.  The eq weight w = E_out*E_in is formed
// once per pair and multiplies the SUM over v of the product evaluations (F multiplies per pair instead of 2V + V), and a
// coefficient c_v costs two multiplies per product -- none when the caller pre-scaled it into the first factor's table.
// Per pair: 1 + V*(tree + 2*[c_v != 1]) + F multiplies (V = 8, F = 4: 101, pre-scaled 85, against 112+ for the (v, pair) items of
// k_split_eq_uniform, which stays the better shape once a round has too few pairs to fill the chip).
// `load(v, k, row, lo, hi)` yields the pair of factor k of product v.
template <int F, class Load>
__device__ __forceinline__ void uniform_rows_body(const Load& load, int V, const Fr* __restrict__ coeff, const uint32_t* __restrict__ coeff_one,
                                                  const Fr* __restrict__ e_out, const Fr* __restrict__ e_in, int in_bits, size_t rows, Fr (&acc)[F]) {
    const size_t mask = ((size_t)1 << in_bits) - 1;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t row = (size_t)blockIdx.x * kBlock + threadIdx.x; row < rows; row += stride) {
        // (Summing the V products in deferred-reduction accumulators -- field.hip.h WideAcc -- was measured SLOWER here: 1335 vs
        // 1062 us for round 0 at T = 2^20; the 17-limb carry ripples and 132 VGPRs cost more than the saved REDC rows.)
        Fr s[F];
#pragma unroll
        for (int t = 0; t < F; ++t) s[t] = Fr::zero();
        for (int v = 0; v < V; ++v) {
            Fr lo[F], hi[F];
#pragma unroll
            for (int k = 0; k < F; ++k) load(v, k, row, lo[k], hi[k]);
            if (!coeff_one[v]) {
                const Fr c = coeff[v];
                lo[0] = mul(lo[0], c);
                hi[0] = mul(hi[0], c);
            }
            Fr q[F];
            uniform_item<F>(lo, hi, q);
#pragma unroll
            for (int t = 0; t < F; ++t) s[t] = add(s[t], q[t]);
        }
        const Fr w = mul(ld_fr(e_out + (row >> in_bits)), ld_fr(e_in + (row & mask)));
#pragma unroll
        for (int t = 0; t < F; ++t) acc[t] = add(acc[t], mul(s[t], w));
    }
}
template <int F>
static __global__ __launch_bounds__(kBlock) void k_split_eq_uniform_rows(UniformArgs a, const Fr* __restrict__ e_out, const Fr* __restrict__ e_in, int in_bits,
                                                                         size_t rows, Fr* __restrict__ partials, uint32_t ticket, uint32_t slot, RoundDone rd) {
    Fr acc[F];
#pragma unroll
    for (int t = 0; t < F; ++t) acc[t] = Fr::zero();
    auto load = [&](int v, int k, size_t row, Fr& lo, Fr& hi) {
        const Fr* __restrict__ tp = a.tabs[v * F + k];
        lo = ld_fr(tp + 2 * row);
        hi = ld_fr(tp + 2 * row + 1);
    };
    uniform_rows_body<F>(load, a.V, a.coeff, a.coeff_one, e_out, e_in, in_bits, rows, acc);
    block_reduce_store<F>(acc, partials);
    finish_member(partials, F, ticket, slot, rd);
}

// summand summed over the whole hypercube (member input claim): same descriptor, no pairing
static __global__ __launch_bounds__(kBlock) void k_member_claim(const MemberDesc* __restrict__ d, TablePtrs tabs, size_t len, Fr* __restrict__ partials) {
    Fr acc[1] = {Fr::zero()};
    size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t x = (size_t)blockIdx.x * kBlock + threadIdx.x; x < len; x += stride) {
        for (uint32_t g = 0; g < d->n_groups; ++g) {
            Fr prod = Fr::one();
            for (uint32_t f = d->grp_fac_off[g]; f < d->grp_fac_off[g + 1]; ++f) {
                Fr v = d->fac_has_const[f] ? d->fac_const[f] : Fr::zero();
                for (uint32_t k = d->fac_lc_off[f]; k < d->fac_lc_off[f + 1]; ++k) {
                    Fr a = ld_fr(tabs.p[d->lc_tab[k]] + x);
                    if (!d->lc_one[k]) a = mul(a, d->lc_coeff[k]);
                    v = add(v, a);
                }
                prod = mul(prod, v);
            }
            acc[0] = add(acc[0], prod);
        }
    }
    block_reduce_store<1>(acc, partials);
}

}  // namespace jolt
