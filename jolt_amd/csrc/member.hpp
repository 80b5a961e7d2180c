// jolt_amd/csrc/member.hpp -- the object behind `jolt_member*`: one sumcheck batch member with device-resident tables
// (the device twin of Box<dyn SumcheckKernel>, crates/jolt-kernels/src/kernel.rs:72-126).
#pragma once
#include "ctx.hpp"
#include "desc.hpp"

struct jolt_member {
    enum Kind { kExpr = 0, kSplitEqProduct = 1, kSplitEqUniform = 2, kSplitEqBooleanity = 3 };
    jolt_ctx* ctx = nullptr;
    int kind = kExpr;
    size_t rounds = 0, bound = 0;
    size_t len = 0;  // current table length
    uint32_t degree = 0;
    int32_t order = JOLT_ORDER_LOW_TO_HIGH;
    bool skip_one = false;
    size_t muls_per_pair = 0;     // field multiplies of the round kernel per pair (decides whether fusing the bind pays)
    bool all_tables_used = true;  // every table is mentioned by the summand (needed to fuse binds into the round kernel)
    bool borrowed = false;  // tables are views of caller-owned tables (never written); scratch is owned
    std::vector<jolt_table*> tables;  // owned
    jolt::MemberDesc desc;            // host copy
    jolt::MemberDesc* d_desc = nullptr;
    // kExpr with the eq weight factored out (the optimized tier's eq(r,j) * q(j) members, e.g. optimized/instruction_input.rs:1-22):
    // `desc` is the INNER summand q of degree `degree` (s(1) skipped), every pair's product values are multiplied by E_out*E_in of
    // its row, the host assembles s(t) = l(t) q(t) (gruen_poly_from_q); the message degree is degree + 1.
    bool eq_weighted = false;
    bool has_split_eq() const { return kind != kExpr || eq_weighted; }
    // split-eq state (GruenSplitEqPolynomial, crates/jolt-poly/src/split_eq.rs:159-166)
    std::vector<Fr> w;
    Fr current_scalar, initial_scalar;
    size_t out_len = 0, in_len = 0;        // lengths of out_point / in_point
    size_t e_out_bits = 0, e_in_bits = 0;  // prefix lengths of the CURRENT E_out / E_in tables
    std::vector<jolt_table*> e_out_cache, e_in_cache;  // evals_cached: index j = eq over the first j coordinates
    // split-eq uniform product: eq * sum_v coeff[v] * prod_{i<F} tables[v*F+i]
    uint32_t uni_V = 0, uni_F = 0;
    std::vector<Fr> uni_coeff;
    // lazily bound one-hot selector columns (LazyFoldedRa, crates/jolt-kernels/src/optimized/lazy_ra.rs:55-182): until the fourth
    // bind table p is index-encoded, value(p, j) = sum_{off<width} branch[p][off*K + index(p, j*width + off)]; `tables` then
    // only carry the bookkeeping length.  The fourth bind materialises them dense at cycles/16 and the member continues as a
    // plain split-eq uniform member.
    const struct jolt_onehot* onehot = nullptr;
    uint32_t lazy_width = 0;  // 0: dense state; 1, 2, 4, 8: index-encoded with that many branches
    Fr* d_branch[2] = {nullptr, nullptr};  // ping-pong branch tables [poly][width*K] (capacity 16*K per polynomial)
    int branch_cur = 0;
    // F = 4, unbound state only: pair tables P[v][h][a*17+b] = T_{4v+2h}[a] * T_{4v+2h+1}[b] (index 16 = cold = 0), so that the first
    // round's quadratic halves (f0*f1 at 0, 1, 2) are gathers and additions instead of three multiplies each
    Fr* d_pair = nullptr;
    Fr* d_base = nullptr;     // the unbound scale tables [poly][K] (kept for jolt_member_reset), product coefficients folded in
    // c_v is pre-scaled into the scale table of product v's first factor (the reference's gamma pre-scaling,
    // optimized/booleanity.rs:32-38): kernels see coefficient one, the reported final values are multiplied by unscale[table]
    // kSplitEqBooleanity: eq(w,j) * sum_i H_i(j) * (H_i(j) - rho[i]) over the member's columns (two round sums, like the product member)
    std::vector<Fr> bool_rho;
    bool uni_prescaled = false;
    std::vector<Fr> final_unscale;  // per table, empty = none
    // jolt_member_create_lc_small: some tables are u64 witness columns until the first bind (Polynomial<T> compact scalars + bind_to_field, dense.rs:129-142)
    jolt::SmallDesc* h_small = nullptr;  // host copy (source of the upload), owned
    jolt::SmallDesc* d_small = nullptr;
    std::vector<void*> promoted;         // members too small for the integer round kernel promote their columns once, here
    bool ints_live() const {
        for (const jolt_table* t : tables) if (t->ints) return true;
        return false;
    }
};

size_t jolt_internal_member_n_evals(const jolt_member* m);
int32_t jolt_internal_bind(jolt_ctx* ctx, jolt_table* const* tables, size_t k, const Fr& r, int32_t order);
