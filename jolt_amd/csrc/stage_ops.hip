// jolt_amd/csrc/stage_ops.hip -- the stage operators as ProveRounds objects behind the C ABI (HOST code: hipcc emits no device code for this file).
//
// The reference's backend is a struct of slots (crates/jolt-kernels/src/backend.rs:126-171); a slot is a PrepareKernel whose `prepare` returns a
// Box<dyn SumcheckKernel> (kernel.rs:72-126) = ProveRounds (crates/jolt-sumcheck/src/prover.rs:52-72) + output_claims.  A `jolt_stage_op` is that object for one
// operator: its constructor is the slot's `prepare` (every T-scale pass that does not depend on a round challenge runs there), prove_round / finish_rounds are the
// fused contract (the previous round's challenge arrives with the next round's request, prover.rs:45-51) and return the round message as COEFFICIENTS
// (UnivariatePoly), output_claims are the kernel's output claims.  The per-operator round loops that rounds 2-5 kept in Python (jolt_amd/stages.py) live here now;
// what stays above the ABI is the order in which a prover calls the stages.
//
//   operator (constructor)                         reference slot / kernel
//   jolt_stage_spartan_remainder_create            spartan_outer / spartan_product remainders   optimized/spartan_outer.rs:236-300,780-850, spartan_product.rs:321-437
//   jolt_stage_ram_read_write_create               ram_read_write                               optimized/ram_read_write.rs:58-330
//   jolt_stage_registers_read_write_create         registers_read_write                         optimized/registers_read_write/mod.rs:79-402
//   jolt_stage_booleanity_address_create           booleanity_address                           optimized/booleanity.rs:152-427
//   jolt_stage_booleanity_cycle_create             booleanity_cycle                             optimized/booleanity.rs:436-690
//   jolt_stage_hamming_weight_create               hamming_weight_claim_reduction               optimized/hamming_weight_claim_reduction.rs:83-300
//   jolt_stage_instruction_read_raf_create         instruction_read_raf (address + cycle)       optimized/instruction_read_raf.rs:736-1456
//   jolt_stage_bytecode_read_raf_address_create    bytecode_read_raf_address                    optimized/bytecode_read_raf.rs:152-437
//   jolt_stage_bytecode_read_raf_cycle_create      bytecode_read_raf_cycle                      optimized/bytecode_read_raf.rs:440-690
//   jolt_stage_ram_raf_evaluation_create           ram_raf_evaluation                           optimized/ram_raf_evaluation.rs:17-62
//   jolt_stage_ram_output_check_create             ram_output_check                             optimized/ram_output_check.rs:50-215
//
// Drivers (the reference's callers, restated above the contract for the tests and the bench; a Rust host calls prove_round / finish_rounds from ITS prove_batch):
//   jolt_host_prove_batch_ops     prove_batch (prover.rs:193-362) over operators, the library's test transcript
//   jolt_host_stage_op_prove_alone  one operator driven alone, every message absorbed coefficient by coefficient (the loop of the kernels' own unit tests)
#include <map>
#include <string>

#include "host_mirror.hpp"
#include "ints.hpp"
#include "member.hpp"
#include "onehot.hpp"

using namespace jolt;
using namespace jolt_host;

// ------------------------------------------------------------------------------------------------------------------
// the object
// ------------------------------------------------------------------------------------------------------------------
struct jolt_stage_op : ProveRounds {
    jolt_ctx* ctx = nullptr;
    size_t rounds = 0, degree = 0;  // degree: the largest degree a round message can have
    std::vector<Fr> binds;          // the challenges received so far, in round order
    std::map<std::string, std::vector<Fr>> kept;  // intermediate values the parity tests compare (scan sums, pushforward masses, ...): host copies, O(K)
    Fr carry = Fr::zero();          // a window's last challenge, waiting for the next window's first round (jolt_stage_op_window)
    bool has_carry = false;
    const char* name = "";

    size_t num_rounds() const override { return rounds; }
    virtual int32_t output_claims(std::vector<Fr>* out) = 0;
    virtual int32_t input_claim(Fr* /*out*/) { return JOLT_ERR_UNSUPPORTED; }

    int32_t fail(int32_t st, const std::string& what) const {
        if (ctx && st != JOLT_OK && ctx->last_error.empty()) ctx->last_error = std::string(name) + ": " + what;
        return st;
    }
};

namespace {

// ---- owning handles -------------------------------------------------------------------------------------------------------------
struct TableH {
    jolt_ctx* ctx = nullptr;
    jolt_table* t = nullptr;
    TableH() = default;
    TableH(jolt_ctx* c, jolt_table* p) : ctx(c), t(p) {}
    TableH(const TableH&) = delete;
    TableH& operator=(const TableH&) = delete;
    TableH(TableH&& o) noexcept : ctx(o.ctx), t(o.t) { o.t = nullptr; }
    TableH& operator=(TableH&& o) noexcept {
        if (this != &o) { reset(); ctx = o.ctx; t = o.t; o.t = nullptr; }
        return *this;
    }
    ~TableH() { reset(); }
    void reset() {
        if (t) jolt_table_free(ctx, t);
        t = nullptr;
    }
    jolt_table* release() {
        jolt_table* p = t;
        t = nullptr;
        return p;
    }
};
struct MemberH {
    jolt_member* m = nullptr;
    MemberH() = default;
    MemberH(const MemberH&) = delete;
    MemberH& operator=(const MemberH&) = delete;
    ~MemberH() { reset(); }
    void reset() {
        if (m) jolt_member_destroy(m);
        m = nullptr;
    }
};

std::vector<jolt_fr_t> to_abi(const std::vector<Fr>& v) {
    std::vector<jolt_fr_t> out(v.size() ? v.size() : 1);
    for (size_t i = 0; i < v.size(); ++i) fr_to_abi(&out[i], v[i]);
    return out;
}
std::vector<Fr> from_abi(const jolt_fr_t* p, size_t n) {
    std::vector<Fr> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = fr_from_abi(&p[i]);
    return out;
}
bool all_canonical(const jolt_fr_t* p, size_t n) {
    for (size_t i = 0; i < n; ++i)
        if (!fr_is_canonical(fr_from_abi(&p[i]))) return false;
    return true;
}
std::vector<Fr> reversed(const Fr* first, size_t n) {
    std::vector<Fr> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = first[n - 1 - i];
    return out;
}
size_t log2_exact(size_t v) {
    size_t l = 0;
    while (((size_t)1 << l) < v) ++l;
    return l;
}

int32_t eq_table(jolt_ctx* ctx, const std::vector<Fr>& point, TableH* out) {
    std::vector<jolt_fr_t> p = to_abi(point);
    jolt_table* t = nullptr;
    JOLT_TRY(jolt_eq_evals(ctx, p.data(), point.size(), nullptr, &t));
    *out = TableH(ctx, t);
    return JOLT_OK;
}
int32_t host_eq(const std::vector<Fr>& point, std::vector<jolt_fr_t>* out) {
    std::vector<jolt_fr_t> p = to_abi(point);
    out->assign((size_t)1 << point.size(), jolt_fr_t{});
    return jolt_host_eq_evals(p.data(), point.size(), nullptr, out->data());
}
int32_t u64_table(jolt_ctx* ctx, const std::vector<uint64_t>& v, TableH* out) {
    jolt_table* t = nullptr;
    JOLT_TRY(jolt_table_from_u64(ctx, v.data(), v.size(), &t));
    *out = TableH(ctx, t);
    return JOLT_OK;
}
int32_t fr_table(jolt_ctx* ctx, const jolt_fr_t* v, size_t n, TableH* out) {
    jolt_table* t = nullptr;
    JOLT_TRY(jolt_table_upload(ctx, v, n, &t));
    *out = TableH(ctx, t);
    return JOLT_OK;
}
int32_t rlc_tables(jolt_ctx* ctx, const std::vector<jolt_table*>& tables, const std::vector<Fr>& scalars, TableH* out) {
    std::vector<jolt_fr_t> s = to_abi(scalars);
    jolt_table* t = nullptr;
    JOLT_TRY(jolt_rlc(ctx, tables.data(), tables.size(), s.data(), &t));
    *out = TableH(ctx, t);
    return JOLT_OK;
}

// One round of ONE device member with the message's field inversion computed while the device runs the round (what DeviceGroupedRounds does for a batch).
int32_t member_round(jolt_ctx* ctx, jolt_member* m, const Fr* bind, const Fr& claim, UnivariatePoly* out) {
    DeviceMember dm(m);
    jolt_member* ms[1] = {m};
    jolt_fr_t b;
    const jolt_fr_t* binds[1] = {nullptr};
    if (bind) {
        fr_to_abi(&b, *bind);
        binds[0] = &b;
    }
    Fr l1 = Fr::zero(), inv_l1 = Fr::zero();
    const bool has = dm.next_l1(bind != nullptr, bind ? *bind : Fr::zero(), &l1) && !l1.is_zero();
    const std::function<void()> overlap = [&]() {
        if (has) inv_l1 = inv(l1);
    };
    jolt_fr_t evals[JOLT_MAX_DEGREE + 1];
    const size_t ne = dm.n_evals();
    JOLT_TRY(jolt_internal_round_group_prove(ctx, ms, 1, binds, evals, ne, &overlap));
    Fr ev[JOLT_MAX_DEGREE + 1];
    for (size_t k = 0; k < ne; ++k) ev[k] = fr_from_abi(&evals[k]);
    return dm.assemble(ev, claim, out, has ? &inv_l1 : nullptr);
}
int32_t member_finish(jolt_member* m, const Fr& bind) {
    jolt_fr_t b;
    fr_to_abi(&b, bind);
    return jolt_member_finish(m, &b);
}
int32_t member_finals(jolt_member* m, size_t k, std::vector<Fr>* out) {
    std::vector<jolt_fr_t> v(k ? k : 1);
    JOLT_TRY(jolt_member_final_values(m, v.data(), k));
    *out = from_abi(v.data(), k);
    return JOLT_OK;
}
// a dense member over a product of tables in jolt_claims::Expr form: sum_k coeffs[k] prod_{f in term k} tables[f] (takes ownership of the tables)
int32_t expr_member(jolt_ctx* ctx, std::vector<TableH>& tables, const std::vector<std::pair<Fr, std::vector<uint32_t>>>& terms, uint32_t degree, MemberH* out) {
    std::vector<uint32_t> offs(1, 0), facs;
    std::vector<Fr> coeffs;
    for (const auto& t : terms) {
        facs.insert(facs.end(), t.second.begin(), t.second.end());
        offs.push_back((uint32_t)facs.size());
        coeffs.push_back(t.first);
    }
    if (facs.empty()) facs.push_back(0);
    std::vector<jolt_fr_t> c = to_abi(coeffs);
    jolt_member_desc d;
    d.n_tables = (uint32_t)tables.size();
    d.n_terms = (uint32_t)terms.size();
    d.degree = degree;
    d.order = JOLT_ORDER_LOW_TO_HIGH;
    d.term_offsets = offs.data();
    d.factors = facs.data();
    d.coeffs = c.data();
    std::vector<jolt_table*> hs;
    for (TableH& t : tables) hs.push_back(t.t);
    JOLT_TRY(jolt_member_create_expr(ctx, hs.data(), &d, &out->m));
    for (TableH& t : tables) t.release();  // ownership moved into the member
    return JOLT_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Spartan outer / product: the remainder rounds after the uni-skip round
// ------------------------------------------------------------------------------------------------------------------
struct SpartanRemainderOp final : jolt_stage_op {
    std::vector<const jolt_ints*> cols;
    size_t cycle_vars = 0;
    MemberH member;

    int32_t prove_round(const Fr* bind, size_t, const Fr& claim, UnivariatePoly* out) override {
        if (bind) binds.push_back(*bind);
        return member_round(ctx, member.m, bind, claim, out);
    }
    int32_t finish_rounds(const Fr& bind) override {
        binds.push_back(bind);
        return member_finish(member.m, bind);
    }
    int32_t input_claim(Fr* out) override {
        jolt_fr_t c;
        JOLT_TRY(jolt_member_input_claim(member.m, &c));
        *out = fr_from_abi(&c);
        return JOLT_OK;
    }
    // compute_claimed_inputs (optimized/spartan_outer.rs:780-850): every input column at the cycle coordinates of the bind point, most significant first
    int32_t output_claims(std::vector<Fr>* out) override {
        if (binds.size() != rounds) return JOLT_ERR_NOT_FULLY_BOUND;
        std::vector<jolt_fr_t> point = to_abi(reversed(binds.data() + (rounds - cycle_vars), cycle_vars));
        std::vector<jolt_fr_t> v(cols.size() ? cols.size() : 1);
        JOLT_TRY(jolt_ints_evaluate(ctx, cols.data(), cols.size(), point.data(), cycle_vars, v.data()));
        *out = from_abi(v.data(), cols.size());
        return JOLT_OK;
    }
};

// ------------------------------------------------------------------------------------------------------------------
// RAM / registers read-write checking over the sparse matrix
// ------------------------------------------------------------------------------------------------------------------
struct RwOp final : jolt_stage_op {
    jolt_rw_matrix* m = nullptr;
    bool registers = false;
    size_t log_t = 0, log_k = 0;
    const jolt_onehot* regs = nullptr;

    ~RwOp() override {
        if (m) jolt_rw_matrix_destroy(m);
    }
    // ram_read_write.rs:160-217 / registers_read_write/mod.rs:217-252: cycle rounds complete the cubic with gruen_poly_deg_3 from the two sums and the split-eq
    // state, address rounds interpolate (s(0), claim - s(0), s(2)) (RAM) or the four sampled points (registers)
    int32_t prove_round(const Fr* bind, size_t round, const Fr& claim, UnivariatePoly* out) override {
        jolt_fr_t b, evals[4], aux[3];
        if (bind) {
            binds.push_back(*bind);
            fr_to_abi(&b, *bind);
        }
        if (registers) JOLT_TRY(jolt_registers_rw_prove_round(m, bind ? &b : nullptr, evals, aux));
        else JOLT_TRY(jolt_rw_matrix_prove_round(m, bind ? &b : nullptr, evals, aux));
        const Fr e0 = fr_from_abi(&evals[0]), e1 = fr_from_abi(&evals[1]);
        if (round < log_t) return gruen_poly_deg_3(fr_from_abi(&aux[0]), fr_from_abi(&aux[1]), e0, e1, claim, out);
        if (registers) {
            const Fr e[4] = {e0, e1, fr_from_abi(&evals[2]), fr_from_abi(&evals[3])};
            *out = UnivariatePoly::from_evals(e, 4);
        } else {
            const Fr e[3] = {e0, sub(claim, e0), e1};
            *out = UnivariatePoly::from_evals(e, 3);
        }
        return JOLT_OK;
    }
    int32_t finish_rounds(const Fr& bind) override {
        binds.push_back(bind);
        jolt_fr_t b;
        fr_to_abi(&b, bind);
        return jolt_rw_matrix_finish(m, &b);
    }
    // RAM: {ra, val, inc, bound cycle-eq factor}.  Registers: {registers_val, rd_wa, gamma rs1_ra + gamma^2 rs2_ra, rd_inc, bound cycle-eq factor} and
    // RegistersReadWriteOutputClaims::{rs1_ra, rs2_ra}: the index columns evaluated at (r_address, r_cycle) = the reversed halves of the challenges
    int32_t output_claims(std::vector<Fr>* out) override {
        if (binds.size() != rounds) return JOLT_ERR_NOT_FULLY_BOUND;
        jolt_fr_t v[5];
        if (!registers) {
            JOLT_TRY(jolt_rw_matrix_final_values(m, v));
            *out = from_abi(v, 4);
            return JOLT_OK;
        }
        JOLT_TRY(jolt_registers_rw_final_values(m, v));
        *out = from_abi(v, 5);
        std::vector<jolt_fr_t> r_cycle = to_abi(reversed(binds.data(), log_t)), eq_adr;
        JOLT_TRY(host_eq(reversed(binds.data() + log_t, log_k), &eq_adr));
        TableH scale;
        JOLT_TRY(fr_table(ctx, eq_adr.data(), eq_adr.size(), &scale));
        for (size_t p = 0; p < 2; ++p) {
            jolt_table* col = nullptr;
            JOLT_TRY(jolt_onehot_materialize(ctx, regs, p, scale.t, &col));
            TableH hold(ctx, col);
            jolt_fr_t c;
            JOLT_TRY(jolt_table_evaluate(ctx, col, r_cycle.data(), log_t, &c));
            out->push_back(fr_from_abi(&c));
        }
        return JOLT_OK;
    }
};

// ------------------------------------------------------------------------------------------------------------------
// Booleanity, address phase: T-scale pushforward in the constructor, log K rounds over K-entry tables on the host (where the reference keeps them)
// ------------------------------------------------------------------------------------------------------------------
struct BooleanityAddressOp final : jolt_stage_op {
    size_t n_polys = 0, K = 0, len = 0;
    std::vector<jolt_fr_t> linear, squared, weights, eq;

    int32_t prove_round(const Fr* bind, size_t, const Fr&, UnivariatePoly* out) override {
        if (bind) JOLT_TRY(apply(*bind));
        jolt_fr_t e[4];
        JOLT_TRY(jolt_host_booleanity_address_round(linear.data(), squared.data(), n_polys, K, len, weights.data(), eq.data(), e));
        const Fr ev[4] = {fr_from_abi(&e[0]), fr_from_abi(&e[1]), fr_from_abi(&e[2]), fr_from_abi(&e[3])};
        *out = UnivariatePoly::from_evals(ev, 4);
        return JOLT_OK;
    }
    int32_t finish_rounds(const Fr& bind) override { return apply(bind); }
    int32_t apply(const Fr& bind) {
        if (len < 2) return JOLT_ERR_INVALID_ARG;
        binds.push_back(bind);
        jolt_fr_t b;
        fr_to_abi(&b, bind);
        JOLT_TRY(jolt_host_booleanity_address_bind(linear.data(), squared.data(), n_polys, K, len, eq.data(), &b));
        len /= 2;
        return JOLT_OK;
    }
    int32_t input_claim(Fr* out) override {  // zero by construction (booleanity.rs:344-403)
        *out = Fr::zero();
        return JOLT_OK;
    }
    // the phase's intermediate claim eq(r_address) * sum_i gamma^(2i) (G2_i - G_i) at the bound point
    int32_t output_claims(std::vector<Fr>* out) override {
        if (len != 1) return JOLT_ERR_NOT_FULLY_BOUND;
        Fr acc = Fr::zero();
        for (size_t i = 0; i < n_polys; ++i) acc = add(acc, mul(fr_from_abi(&weights[i]), sub(fr_from_abi(&squared[i * K]), fr_from_abi(&linear[i * K]))));
        out->assign(1, mul(fr_from_abi(&eq[0]), acc));
        return JOLT_OK;
    }
};

// ------------------------------------------------------------------------------------------------------------------
// Booleanity, cycle phase (stage 6b): eq(reference_cycle, j) * eq(r_address, reference_address) * sum_i (H_i(j)^2 - gamma^i H_i(j)) over the lazily bound RA columns,
// H_i(j) = gamma^i eq(r_address)[hot_i(j)] (optimized/booleanity.rs:436-633).  The member is jolt_member_create_lazy_booleanity; the round message is
// gruen_poly_deg_3 of its two sums (member_round); the output claims are the bound columns unscaled by gamma^-i (booleanity.rs:652-657).
// ------------------------------------------------------------------------------------------------------------------
struct BooleanityCycleOp final : jolt_stage_op {
    size_t n_polys = 0;
    bool dense = false;  // fewer than four cycle variables: the same summand over materialised columns (the lazy form is index-encoded for four binds)
    std::vector<Fr> rho_inv;
    MemberH member;

    int32_t prove_round(const Fr* bind, size_t, const Fr& claim, UnivariatePoly* out) override {
        if (bind) binds.push_back(*bind);
        return member_round(ctx, member.m, bind, claim, out);
    }
    int32_t finish_rounds(const Fr& bind) override {
        binds.push_back(bind);
        return member_finish(member.m, bind);
    }
    // (the input claim is the address phase's intermediate claim, BooleanityAddressPhaseOutputClaims::intermediate: the caller holds it; a wrong one fails the first round check)
    int32_t output_claims(std::vector<Fr>* out) override {
        if (binds.size() != rounds) return JOLT_ERR_NOT_FULLY_BOUND;
        std::vector<Fr> fin;
        JOLT_TRY(member_finals(member.m, n_polys + 1, &fin));  // the bound H_i and the eq scalar (the fully bound EqAddressCycle, validate_derived_tables :666-680): last (lazy) or first (dense)
        out->resize(n_polys);
        for (size_t i = 0; i < n_polys; ++i) (*out)[i] = mul(fin[dense ? 1 + i : i], rho_inv[i]);
        kept["eq_scalar"] = {fin[dense ? 0 : n_polys]};
        return JOLT_OK;
    }
};

// ------------------------------------------------------------------------------------------------------------------
// Hamming-weight claim reduction
// ------------------------------------------------------------------------------------------------------------------
struct HammingWeightOp final : jolt_stage_op {
    size_t n_polys = 0, K = 0, len = 0;
    std::vector<jolt_fr_t> g, w;

    // s(0) and s(2) per round, s(1) recovered from the running claim (round_poly_from_skipped_evals, hamming_weight_claim_reduction.rs:268-298)
    int32_t prove_round(const Fr* bind, size_t, const Fr& claim, UnivariatePoly* out) override {
        if (bind) JOLT_TRY(apply(*bind));
        jolt_fr_t e[3];
        JOLT_TRY(jolt_host_pair_tables_round(g.data(), w.data(), n_polys, K, len, e));
        const Fr s0 = fr_from_abi(&e[0]);
        const Fr ev[3] = {s0, sub(claim, s0), fr_from_abi(&e[1])};
        *out = UnivariatePoly::from_evals(ev, 3);
        return JOLT_OK;
    }
    int32_t finish_rounds(const Fr& bind) override { return apply(bind); }
    int32_t apply(const Fr& bind) {
        if (len < 2) return JOLT_ERR_INVALID_ARG;
        binds.push_back(bind);
        jolt_fr_t b;
        fr_to_abi(&b, bind);
        JOLT_TRY(jolt_host_pair_tables_bind(g.data(), w.data(), n_polys, K, len, &b));
        len /= 2;
        return JOLT_OK;
    }
    int32_t input_claim(Fr* out) override {  // sum_i sum_k G_i(k) W_i(k) before the first round
        if (len != K) return JOLT_ERR_INVALID_ARG;
        jolt_fr_t e[3];
        JOLT_TRY(jolt_host_pair_tables_round(g.data(), w.data(), n_polys, K, len, e));
        *out = fr_from_abi(&e[2]);
        return JOLT_OK;
    }
    int32_t output_claims(std::vector<Fr>* out) override {  // the bound G_i
        if (len != 1) return JOLT_ERR_NOT_FULLY_BOUND;
        out->clear();
        for (size_t i = 0; i < n_polys; ++i) out->push_back(fr_from_abi(&g[i * K]));
        return JOLT_OK;
    }
};

// pushforward masses of all columns of `cols` against eq(point, .): (n_polys x K) host values
int32_t pushforward_masses(jolt_ctx* ctx, const jolt_onehot* cols, const std::vector<Fr>& point, std::vector<jolt_fr_t>* out) {
    TableH eq;
    JOLT_TRY(eq_table(ctx, point, &eq));
    jolt_table* g = nullptr;
    JOLT_TRY(jolt_onehot_pushforward(ctx, cols, eq.t, &g));
    TableH hold(ctx, g);
    out->assign(cols->n_polys * (size_t)cols->k, jolt_fr_t{});
    return jolt_table_download(ctx, g, 0, out->size(), out->data());
}

// ------------------------------------------------------------------------------------------------------------------
// Instruction read + RAF: 128 address rounds (16 phases: T-scale scans on the device, 8 rounds over 256-entry polynomials on the host) and log T cycle rounds
// ------------------------------------------------------------------------------------------------------------------
constexpr size_t kAddressBits = 128, kPhaseRounds = 8, kPhases = kAddressBits / kPhaseRounds, kChunk = 256, kLookupTables = 42;

struct InstructionReadRafOp final : jolt_stage_op {
    jolt_read_raf* rr = nullptr;             // borrowed: the resident lookup rows
    const jolt_onehot* claim_columns = nullptr;  // borrowed: the packed output-claim facts (tables 0..15 / 16..31 / 32..41 in three K = 16 columns, RAF rows in a fourth)
    jolt_read_raf_address* state = nullptr;
    std::vector<Fr> reduction;
    std::vector<uint8_t> present;
    uint32_t ra_count = 0;
    size_t n_vars = 0, n_f = 0;
    std::vector<uint32_t> suffix_offsets;
    std::vector<uint8_t> suffix_kinds;
    TableH u;
    uint32_t phase = 0;  // the open phase
    std::vector<jolt_fr_t> raf, suf;  // the open phase's scan sums (init_phase keeps pointers into them only during the call)
    MemberH member;

    ~InstructionReadRafOp() override {
        if (state) jolt_host_read_raf_address_destroy(state);
    }
    // init_phase (instruction_read_raf.rs:824-971): condensation of the per-cycle mass with the previous phase's eq table, the fused RAF scan and the per-table
    // suffix accumulators on the device, the 256-entry prefix / suffix polynomials on the host
    int32_t open_phase() {
        const uint32_t suffix_len = (uint32_t)(kAddressBits - kPhaseRounds * (phase + 1));
        if (phase) {
            jolt_fr_t v[kChunk];
            JOLT_TRY(jolt_host_read_raf_address_v_table(state, phase - 1, v));
            JOLT_TRY(jolt_read_raf_condense(ctx, rr, u.t, v, suffix_len + (uint32_t)kPhaseRounds));
        }
        const size_t total = suffix_offsets[kLookupTables];
        raf.assign(6 * kChunk, jolt_fr_t{});
        suf.assign(std::max<size_t>(total * kChunk, 1), jolt_fr_t{});
        JOLT_TRY(jolt_read_raf_phase_scan(ctx, rr, u.t, suffix_len, (uint32_t)kAddressBits, 0, suffix_offsets.data(), suffix_kinds.empty() ? nullptr : suffix_kinds.data(),
                                          raf.data(), suf.data()));
        JOLT_TRY(jolt_host_read_raf_address_init_phase(state, phase, raf.data(), suf.data()));
        std::vector<Fr>& kr = kept["scan_raf"];
        std::vector<Fr>& ks = kept["scan_suffix"];
        for (const jolt_fr_t& x : raf) kr.push_back(fr_from_abi(&x));
        for (size_t i = 0; i < total * kChunk; ++i) ks.push_back(fr_from_abi(&suf[i]));
        return JOLT_OK;
    }
    // init_cycle_rounds (:1140-1232): table values at r_address, the combined-value and ra_i columns, the member eq(r_reduction, j) * combined(j) * prod_i ra_i(j)
    int32_t open_cycle_rounds() {
        jolt_fr_t tv[kLookupTables], ri, rid;
        JOLT_TRY(jolt_host_read_raf_address_finish(state, tv, &ri, &rid));
        std::vector<jolt_fr_t> vt(kPhases * kChunk);
        for (uint32_t p = 0; p < kPhases; ++p) JOLT_TRY(jolt_host_read_raf_address_v_table(state, p, &vt[p * kChunk]));
        kept["v_tables"] = from_abi(vt.data(), vt.size());
        kept["table_values"] = from_abi(tv, kLookupTables);
        kept["raf_values"] = {fr_from_abi(&ri), fr_from_abi(&rid)};
        u.reset();
        jolt_table* combined = nullptr;
        std::vector<jolt_table*> ra(ra_count, nullptr);
        JOLT_TRY(jolt_read_raf_cycle_tables(ctx, rr, tv, &ri, &rid, vt.data(), (uint32_t)kPhases, (uint32_t)kAddressBits, ra_count, &combined, ra.data()));
        std::vector<TableH> tabs;
        tabs.emplace_back(ctx, combined);
        for (jolt_table* t : ra) tabs.emplace_back(ctx, t);
        // one group of n_f factors, each a single table with coefficient one; the eq weight is factored out (jolt_member_create_split_eq_lc)
        std::vector<uint32_t> goff = {0, (uint32_t)n_f}, foff(n_f + 1), ltab(n_f);
        std::vector<jolt_fr_t> consts(n_f, jolt_fr_t{}), lcoef(n_f);
        for (size_t i = 0; i < n_f; ++i) {
            foff[i + 1] = (uint32_t)(i + 1);
            ltab[i] = (uint32_t)i;
            fr_to_abi(&lcoef[i], Fr::one());
        }
        jolt_member_lc_desc d;
        d.n_tables = (uint32_t)n_f;
        d.n_groups = 1;
        d.n_factors = (uint32_t)n_f;
        d.n_lc = (uint32_t)n_f;
        d.degree = (uint32_t)n_f;
        d.order = JOLT_ORDER_LOW_TO_HIGH;
        d.flags = 0;
        d.group_factor_offsets = goff.data();
        d.factor_lc_offsets = foff.data();
        d.factor_consts = consts.data();
        d.lc_tables = ltab.data();
        d.lc_coeffs = lcoef.data();
        std::vector<jolt_table*> hs;
        for (TableH& t : tabs) hs.push_back(t.t);
        std::vector<jolt_fr_t> w = to_abi(reduction);
        JOLT_TRY(jolt_member_create_split_eq_lc(ctx, hs.data(), &d, w.data(), n_vars, nullptr, nullptr, &member.m));
        for (TableH& t : tabs) t.release();
        return JOLT_OK;
    }
    int32_t prove_round(const Fr* bind, size_t round, const Fr& claim, UnivariatePoly* out) override {
        jolt_fr_t b, c, e[3];
        fr_to_abi(&c, claim);
        if (bind) {
            binds.push_back(*bind);
            fr_to_abi(&b, *bind);
        }
        if (round < kAddressBits) {
            if (round == 0) {
                if (bind) return JOLT_ERR_INVALID_ARG;
                JOLT_TRY(jolt_host_read_raf_address_message(state, &c, e));
            } else if (round % kPhaseRounds) {
                if (!bind) return JOLT_ERR_INVALID_ARG;
                JOLT_TRY(jolt_host_read_raf_address_bind_message(state, &b, &c, e));
            } else {  // the phase's 8th bind closes it; the next phase's scans need its eq table
                if (!bind) return JOLT_ERR_INVALID_ARG;
                int32_t done = 0;
                JOLT_TRY(jolt_host_read_raf_address_bind(state, &b, &done));
                if (!done) return JOLT_ERR_INVALID_ARG;
                phase += 1;
                JOLT_TRY(open_phase());
                JOLT_TRY(jolt_host_read_raf_address_message(state, &c, e));
            }
            const Fr ev[3] = {fr_from_abi(&e[0]), fr_from_abi(&e[1]), fr_from_abi(&e[2])};
            *out = UnivariatePoly::from_evals(ev, 3);
            return JOLT_OK;
        }
        if (round == kAddressBits) {
            if (!bind) return JOLT_ERR_INVALID_ARG;
            int32_t done = 0;
            JOLT_TRY(jolt_host_read_raf_address_bind(state, &b, &done));
            if (!done) return JOLT_ERR_INVALID_ARG;
            JOLT_TRY(open_cycle_rounds());
            kept["cycle_claim"] = {claim};
            return member_round(ctx, member.m, nullptr, claim, out);
        }
        return member_round(ctx, member.m, bind, claim, out);
    }
    int32_t finish_rounds(const Fr& bind) override {
        binds.push_back(bind);
        if (!member.m) return JOLT_ERR_INVALID_ARG;
        return member_finish(member.m, bind);
    }
    // the relation's input claim (the prover holds it from the earlier stages): s_0(0) + s_0(1) summed from the tables of the first phase
    int32_t input_claim(Fr* out) override {
        if (!binds.empty()) return JOLT_ERR_INVALID_ARG;
        jolt_fr_t e[3];
        JOLT_TRY(jolt_host_read_raf_address_message(state, nullptr, e));
        *out = add(fr_from_abi(&e[0]), fr_from_abi(&e[1]));
        return JOLT_OK;
    }
    // output_claims (:1376-1456): [lookup_table_flags of the present tables][instruction_raf_flag][the bound ra_i] -- the flags are masses of eq(r_cycle, .) per
    // lookup table and over the RAF rows: one pushforward of the packed claim columns
    int32_t output_claims(std::vector<Fr>* out) override {
        if (binds.size() != rounds) return JOLT_ERR_NOT_FULLY_BOUND;
        std::vector<Fr> fin;
        JOLT_TRY(member_finals(member.m, n_f + 1, &fin));
        std::vector<jolt_fr_t> flags;
        JOLT_TRY(pushforward_masses(ctx, claim_columns, reversed(binds.data() + kAddressBits, n_vars), &flags));
        out->clear();
        for (size_t t = 0; t < kLookupTables; ++t)
            if (present[t]) out->push_back(fr_from_abi(&flags[(t / 16) * 16 + (t % 16)]));
        out->push_back(fr_from_abi(&flags[3 * 16]));
        for (size_t i = 1; i < n_f; ++i) out->push_back(fin[i]);
        return JOLT_OK;
    }
};

// ------------------------------------------------------------------------------------------------------------------
// Bytecode read + RAF: address phase (6a) and cycle phase (6b); the cycle operator is prepared from the address operator's residue
// ------------------------------------------------------------------------------------------------------------------
struct BytecodeAddressOp final : jolt_stage_op {
    size_t n_vars = 0, log_k = 0;
    uint64_t entry_index = 0;
    std::vector<Fr> gp;            // gamma^0 .. gamma^7
    std::vector<TableH> eqs;       // eq(r_cycle_s, .), s < 5: parked for the cycle phase (SumcheckKernel::park_residue)
    MemberH member;
    std::vector<Fr> fin;           // bound F_0..4, V_0..4, Int, entry_trace, entry_expected

    int32_t prove_round(const Fr* bind, size_t, const Fr& claim, UnivariatePoly* out) override {
        if (bind) binds.push_back(*bind);
        return member_round(ctx, member.m, bind, claim, out);
    }
    int32_t finish_rounds(const Fr& bind) override {
        binds.push_back(bind);
        JOLT_TRY(member_finish(member.m, bind));
        return member_finals(member.m, 13, &fin);
    }
    int32_t input_claim(Fr* out) override {
        jolt_fr_t c;
        JOLT_TRY(jolt_member_input_claim(member.m, &c));
        *out = fr_from_abi(&c);
        return JOLT_OK;
    }
    // [the 13 bound tables][the intermediate claim the cycle phase starts from]
    int32_t output_claims(std::vector<Fr>* out) override {
        if (fin.size() != 13) return JOLT_ERR_NOT_FULLY_BOUND;
        *out = fin;
        Fr intermediate = mul(gp[7], mul(fin[11], fin[12]));
        for (size_t s = 0; s < 5; ++s) {
            Fr val = fin[5 + s];
            if (s == 0) val = add(val, mul(gp[5], fin[10]));
            if (s == 2) val = add(val, mul(gp[4], fin[10]));
            intermediate = add(intermediate, mul(gp[s], mul(fin[s], val)));
        }
        out->push_back(intermediate);
        return JOLT_OK;
    }
};

struct ProductOp final : jolt_stage_op {  // sum_j prod_i tables[i](j), every bound table an output claim from `first_claim` on
    MemberH member;
    size_t n_f = 0, first_claim = 0;

    int32_t prove_round(const Fr* bind, size_t, const Fr& claim, UnivariatePoly* out) override {
        if (bind) binds.push_back(*bind);
        return member_round(ctx, member.m, bind, claim, out);
    }
    int32_t finish_rounds(const Fr& bind) override {
        binds.push_back(bind);
        return member_finish(member.m, bind);
    }
    int32_t input_claim(Fr* out) override {
        jolt_fr_t c;
        JOLT_TRY(jolt_member_input_claim(member.m, &c));
        *out = fr_from_abi(&c);
        return JOLT_OK;
    }
    int32_t output_claims(std::vector<Fr>* out) override {
        std::vector<Fr> fin;
        JOLT_TRY(member_finals(member.m, n_f, &fin));
        out->assign(fin.begin() + first_claim, fin.end());
        return JOLT_OK;
    }
};

struct RamOutputCheckOp final : jolt_stage_op {
    MemberH member;
    TableH val_final;
    size_t log_k = 0;

    int32_t prove_round(const Fr* bind, size_t, const Fr& claim, UnivariatePoly* out) override {
        if (bind) binds.push_back(*bind);
        return member_round(ctx, member.m, bind, claim, out);
    }
    int32_t finish_rounds(const Fr& bind) override {
        binds.push_back(bind);
        return member_finish(member.m, bind);
    }
    int32_t input_claim(Fr* out) override {
        jolt_fr_t c;
        JOLT_TRY(jolt_member_input_claim(member.m, &c));
        *out = fr_from_abi(&c);
        return JOLT_OK;
    }
    int32_t output_claims(std::vector<Fr>* out) override {  // val_final at the bound address point
        if (binds.size() != rounds) return JOLT_ERR_NOT_FULLY_BOUND;
        std::vector<jolt_fr_t> point = to_abi(reversed(binds.data(), log_k));
        jolt_fr_t c;
        JOLT_TRY(jolt_table_evaluate(ctx, val_final.t, point.data(), log_k, &c));
        out->assign(1, fr_from_abi(&c));
        return JOLT_OK;
    }
};

// rounds [first, first + n) of a parent operator as an operator of its own: lets a caller put the phases of one kernel (instruction read-RAF: address rounds,
// cycle rounds) under different drivers or transcripts.  The window's last challenge is carried to the parent's next round by the next window.
struct WindowOp final : jolt_stage_op {
    jolt_stage_op* parent = nullptr;
    size_t first = 0;

    int32_t prove_round(const Fr* bind, size_t round, const Fr& claim, UnivariatePoly* out) override {
        if (round >= rounds) return JOLT_ERR_INVALID_ARG;
        if (!bind && round == 0 && first > 0) {
            if (!parent->has_carry) return JOLT_ERR_INVALID_ARG;
            parent->has_carry = false;
            const Fr c = parent->carry;
            return parent->prove_round(&c, first, claim, out);
        }
        return parent->prove_round(bind, first + round, claim, out);
    }
    int32_t finish_rounds(const Fr& bind) override {
        if (first + rounds == parent->rounds) return parent->finish_rounds(bind);
        parent->carry = bind;
        parent->has_carry = true;
        return JOLT_OK;
    }
    int32_t input_claim(Fr* out) override { return first == 0 ? parent->input_claim(out) : JOLT_ERR_UNSUPPORTED; }
    int32_t output_claims(std::vector<Fr>* out) override { return parent->output_claims(out); }
};

// A dense member over HOST tables (NaiveSumcheckProver, crates/jolt-kernels/src/reference/naive.rs:53-377, LowToHigh): sum_k coeffs[k] prod_{f in term k} table[f].
// No device, no context: it exists so that the drivers above the contract (prove_batch over operators, the alone driver, round windows) and the contract's own error
// behaviour are exercised by the CPU suite against the oracle's prove_batch; the product's operators are the device-backed ones above.
struct HostExprOp final : jolt_stage_op {
    std::vector<std::vector<Fr>> tables;
    std::vector<uint32_t> offs, facs;
    std::vector<Fr> coeffs;
    size_t len = 0;

    Fr summand(const std::vector<Fr>& at) const {
        Fr acc = Fr::zero();
        for (size_t k = 0; k + 1 < offs.size(); ++k) {
            Fr term = coeffs[k];
            for (uint32_t f = offs[k]; f < offs[k + 1]; ++f) term = mul(term, at[facs[f]]);
            acc = add(acc, term);
        }
        return acc;
    }
    int32_t apply(const Fr& r) {
        if (len < 2) return JOLT_ERR_INVALID_ARG;
        binds.push_back(r);
        for (std::vector<Fr>& t : tables) {
            for (size_t y = 0; y < len / 2; ++y) t[y] = add(t[2 * y], mul(r, sub(t[2 * y + 1], t[2 * y])));  // dense.rs:222-263
            t.resize(len / 2);
        }
        len /= 2;
        return JOLT_OK;
    }
    int32_t prove_round(const Fr* bind, size_t, const Fr& claim, UnivariatePoly* out) override {
        if (bind) JOLT_TRY(apply(*bind));
        if (len < 2) return JOLT_ERR_INVALID_ARG;
        std::vector<Fr> evals(degree + 1, Fr::zero()), at(tables.size());
        for (size_t y = 0; y < len / 2; ++y) {
            for (size_t t = 0; t <= degree; ++t) {
                const Fr x = fr_from_u64(t);
                for (size_t i = 0; i < tables.size(); ++i) at[i] = add(tables[i][2 * y], mul(x, sub(tables[i][2 * y + 1], tables[i][2 * y])));
                evals[t] = add(evals[t], summand(at));
            }
        }
        if (add(evals[0], evals[1]) != claim) return JOLT_ERR_ROUND_CHECK;  // naive.rs:298-306
        *out = UnivariatePoly::from_evals(evals.data(), evals.size());
        return JOLT_OK;
    }
    int32_t finish_rounds(const Fr& bind) override { return apply(bind); }
    int32_t input_claim(Fr* out) override {
        std::vector<Fr> at(tables.size());
        Fr acc = Fr::zero();
        for (size_t j = 0; j < len; ++j) {
            for (size_t i = 0; i < tables.size(); ++i) at[i] = tables[i][j];
            acc = add(acc, summand(at));
        }
        *out = acc;
        return JOLT_OK;
    }
    int32_t output_claims(std::vector<Fr>* out) override {
        if (len != 1) return JOLT_ERR_NOT_FULLY_BOUND;
        out->clear();
        for (const std::vector<Fr>& t : tables) out->push_back(t[0]);
        return JOLT_OK;
    }
};

template <class Op>
Op* new_op(jolt_ctx* ctx, const char* name) {
    Op* op = new (std::nothrow) Op();
    if (op) {
        op->ctx = ctx;
        op->name = name;
    }
    return op;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// the contract
// ------------------------------------------------------------------------------------------------------------------
extern "C" int32_t jolt_stage_op_num_rounds(const jolt_stage_op* op, size_t* rounds) {
    if (!op || !rounds) return JOLT_ERR_INVALID_ARG;
    *rounds = op->rounds;
    return JOLT_OK;
}
extern "C" int32_t jolt_stage_op_degree(const jolt_stage_op* op, size_t* degree) {
    if (!op || !degree) return JOLT_ERR_INVALID_ARG;
    *degree = op->degree;
    return JOLT_OK;
}
extern "C" int32_t jolt_stage_op_input_claim(jolt_stage_op* op, jolt_fr_t* claim) {
    if (!op || !claim) return JOLT_ERR_INVALID_ARG;
    Fr c;
    JOLT_TRY(op->input_claim(&c));
    fr_to_abi(claim, c);
    return JOLT_OK;
}
extern "C" int32_t jolt_stage_op_prove_round(jolt_stage_op* op, const jolt_fr_t* bind, size_t round, const jolt_fr_t* previous_claim, jolt_fr_t* coeffs_out, size_t cap,
                                             size_t* n_coeffs) {
    if (!op || !previous_claim || !coeffs_out || !n_coeffs || round >= op->rounds) return JOLT_ERR_INVALID_ARG;
    Fr b = Fr::zero();
    if (bind) {
        b = fr_from_abi(bind);
        if (!fr_is_canonical(b)) return JOLT_ERR_INVALID_ARG;
    }
    const Fr claim = fr_from_abi(previous_claim);
    if (!fr_is_canonical(claim)) return JOLT_ERR_INVALID_ARG;
    UnivariatePoly poly;
    JOLT_TRY(op->prove_round(bind ? &b : nullptr, round, claim, &poly));
    if (poly.coefficients.size() > cap) return JOLT_ERR_SIZE_MISMATCH;
    for (size_t k = 0; k < poly.coefficients.size(); ++k) fr_to_abi(&coeffs_out[k], poly.coefficients[k]);
    *n_coeffs = poly.coefficients.size();
    return JOLT_OK;
}
extern "C" int32_t jolt_stage_op_finish_rounds(jolt_stage_op* op, const jolt_fr_t* bind) {
    if (!op || !bind) return JOLT_ERR_INVALID_ARG;
    const Fr b = fr_from_abi(bind);
    if (!fr_is_canonical(b)) return JOLT_ERR_INVALID_ARG;
    return op->finish_rounds(b);
}
extern "C" int32_t jolt_stage_op_output_claims(jolt_stage_op* op, jolt_fr_t* out, size_t cap, size_t* n) {
    if (!op || !n || (!out && cap)) return JOLT_ERR_INVALID_ARG;
    std::vector<Fr> v;
    JOLT_TRY(op->output_claims(&v));
    *n = v.size();
    if (v.size() > cap) return JOLT_ERR_SIZE_MISMATCH;
    for (size_t i = 0; i < v.size(); ++i) fr_to_abi(&out[i], v[i]);
    return JOLT_OK;
}
// the intermediate values an operator keeps for parity tests, by name (e.g. "masses", "scan_raf", "scan_suffix", "v_tables", "table_values", "raf_values",
// "cycle_claim", "uniskip"): *n = the count; out may be NULL to ask for it
extern "C" int32_t jolt_stage_op_kept(const jolt_stage_op* op, const char* key, jolt_fr_t* out, size_t cap, size_t* n) {
    if (!op || !key || !n) return JOLT_ERR_INVALID_ARG;
    const jolt_stage_op* o = op;
    if (const WindowOp* w = dynamic_cast<const WindowOp*>(op)) o = w->parent;
    const auto it = o->kept.find(key);
    if (it == o->kept.end()) return JOLT_ERR_INVALID_ARG;
    *n = it->second.size();
    if (!out) return JOLT_OK;
    if (it->second.size() > cap) return JOLT_ERR_SIZE_MISMATCH;
    for (size_t i = 0; i < it->second.size(); ++i) fr_to_abi(&out[i], it->second[i]);
    return JOLT_OK;
}
extern "C" int32_t jolt_stage_op_window(jolt_stage_op* parent, size_t first, size_t n, jolt_stage_op** out) {
    if (!parent || !out || first + n > parent->rounds || dynamic_cast<WindowOp*>(parent)) return JOLT_ERR_INVALID_ARG;
    WindowOp* w = new_op<WindowOp>(parent->ctx, parent->name);
    if (!w) return JOLT_ERR_OOM;
    w->parent = parent;
    w->first = first;
    w->rounds = n;
    w->degree = parent->degree;
    *out = w;
    return JOLT_OK;
}
extern "C" int32_t jolt_stage_op_destroy(jolt_stage_op* op) {
    delete op;
    return JOLT_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// constructors (= PrepareKernel::prepare of the slot)
// ------------------------------------------------------------------------------------------------------------------
// The uni-skip first round of Spartan outer / product (UniskipKernel, crates/jolt-kernels/src/uniskip.rs:28-54): the extended-node sums t1 off the integer columns
// against eq(tau_low, .) built here (tau: the n_tau coordinates the sums run over, stream variable last for the outer relation).
extern "C" int32_t jolt_stage_spartan_uniskip_sums(jolt_ctx* ctx, const jolt_ints* const* cols, size_t n_cols, uint32_t n_streams, const jolt_fr_t* tau, size_t n_tau,
                                                   const int64_t* a_weights, const int64_t* b_weights, size_t n_nodes, jolt_fr_t* sums_out) {
    if (!ctx || !cols || !tau || !a_weights || !b_weights || !sums_out || !all_canonical(tau, n_tau)) return JOLT_ERR_INVALID_ARG;
    TableH eq;
    JOLT_TRY(eq_table(ctx, from_abi(tau, n_tau), &eq));
    return jolt_r1cs_uniskip_sums_small(ctx, cols, n_cols, eq.t, n_streams, a_weights, b_weights, n_nodes, sums_out);
}

extern "C" int32_t jolt_stage_spartan_remainder_create(jolt_ctx* ctx, const jolt_ints* const* cols, size_t n_cols, uint32_t n_streams, const jolt_fr_t* a_weights,
                                                       const jolt_fr_t* b_weights, const jolt_fr_t* tau, size_t n_tau, const jolt_fr_t* scale, jolt_stage_op** out) {
    if (!ctx || !cols || !n_cols || !a_weights || !b_weights || !tau || !out || (n_streams != 1 && n_streams != 2) || n_tau < n_streams - 1) return JOLT_ERR_INVALID_ARG;
    SpartanRemainderOp* op = new_op<SpartanRemainderOp>(ctx, n_streams == 2 ? "spartan_outer" : "spartan_product");
    if (!op) return JOLT_ERR_OOM;
    std::unique_ptr<SpartanRemainderOp> hold(op);
    op->cols.assign(cols, cols + n_cols);
    op->rounds = n_tau;
    op->degree = 3;
    op->cycle_vars = n_tau - (n_streams - 1);
    jolt_table *az = nullptr, *bz = nullptr;
    JOLT_TRY(jolt_r1cs_materialize_small(ctx, cols, n_cols, n_streams, a_weights, b_weights, &az, &bz));
    TableH ha(ctx, az), hb(ctx, bz);
    JOLT_TRY(jolt_member_create_split_eq_product(ctx, az, bz, tau, n_tau, scale, &op->member.m));
    ha.release();
    hb.release();
    *out = hold.release();
    return JOLT_OK;
}

extern "C" int32_t jolt_stage_ram_read_write_create(jolt_ctx* ctx, const jolt_ints* addresses, const jolt_ints* pre_values, const jolt_ints* post_values, const jolt_ints* inc,
                                                    const jolt_ints* val_init, const jolt_fr_t* tau_low, const jolt_fr_t* gamma, jolt_stage_op** out) {
    if (!ctx || !addresses || !pre_values || !post_values || !inc || !val_init || !tau_low || !gamma || !out) return JOLT_ERR_INVALID_ARG;
    RwOp* op = new_op<RwOp>(ctx, "ram_read_write");
    if (!op) return JOLT_ERR_OOM;
    std::unique_ptr<RwOp> hold(op);
    op->log_t = log2_exact(addresses->count);
    op->log_k = log2_exact(val_init->count);
    if (((size_t)1 << op->log_t) != addresses->count || ((size_t)1 << op->log_k) != val_init->count) return JOLT_ERR_INVALID_ARG;
    op->rounds = op->log_t + op->log_k;
    op->degree = 3;
    jolt_table *ti = nullptr, *tv = nullptr;
    JOLT_TRY(jolt_table_from_ints(ctx, inc, 0, inc->count, &ti));
    TableH hi(ctx, ti);
    JOLT_TRY(jolt_table_from_ints(ctx, val_init, 0, val_init->count, &tv));
    TableH hv(ctx, tv);
    JOLT_TRY(jolt_rw_matrix_create_resident(ctx, addresses, pre_values, post_values, ti, tv, tau_low, gamma, &op->m));
    *out = hold.release();
    return JOLT_OK;
}

extern "C" int32_t jolt_stage_registers_read_write_create(jolt_ctx* ctx, const jolt_onehot* regs, const jolt_ints* rs1_val, const jolt_ints* rs2_val, const jolt_ints* rd_pre,
                                                          const jolt_ints* rd_post, const jolt_ints* inc, const jolt_fr_t* r_cycle, const jolt_fr_t* gamma, jolt_stage_op** out) {
    if (!ctx || !regs || !rs1_val || !rs2_val || !rd_pre || !rd_post || !inc || !r_cycle || !gamma || !out || regs->n_polys < 3) return JOLT_ERR_INVALID_ARG;
    RwOp* op = new_op<RwOp>(ctx, "registers_read_write");
    if (!op) return JOLT_ERR_OOM;
    std::unique_ptr<RwOp> hold(op);
    op->registers = true;
    op->regs = regs;
    op->log_t = log2_exact(regs->cycles);
    op->log_k = log2_exact(regs->k);
    if (((size_t)1 << op->log_t) != regs->cycles || ((size_t)1 << op->log_k) != regs->k) return JOLT_ERR_INVALID_ARG;
    op->rounds = op->log_t + op->log_k;
    op->degree = 3;
    jolt_table* ti = nullptr;
    JOLT_TRY(jolt_table_from_ints(ctx, inc, 0, inc->count, &ti));
    TableH hi(ctx, ti);
    JOLT_TRY(jolt_registers_rw_create(ctx, regs, rs1_val, rs2_val, rd_pre, rd_post, ti, r_cycle, gamma, &op->m));
    *out = hold.release();
    return JOLT_OK;
}

extern "C" int32_t jolt_stage_booleanity_address_create(jolt_ctx* ctx, const jolt_onehot* cols, const jolt_fr_t* reference_cycle, size_t n_cycle,
                                                        const jolt_fr_t* reference_address, const jolt_fr_t* gamma, jolt_stage_op** out) {
    if (!ctx || !cols || !reference_cycle || !reference_address || !gamma || !out || ((size_t)1 << n_cycle) != cols->cycles) return JOLT_ERR_INVALID_ARG;
    const size_t log_k = log2_exact(cols->k);
    if (((size_t)1 << log_k) != cols->k || !all_canonical(reference_cycle, n_cycle) || !all_canonical(reference_address, log_k)) return JOLT_ERR_INVALID_ARG;
    BooleanityAddressOp* op = new_op<BooleanityAddressOp>(ctx, "booleanity_address");
    if (!op) return JOLT_ERR_OOM;
    std::unique_ptr<BooleanityAddressOp> hold(op);
    op->n_polys = cols->n_polys;
    op->K = op->len = cols->k;
    op->rounds = log_k;
    op->degree = 3;
    // cycle_pushforward (booleanity.rs:152-237): the only T-scale work of the phase
    JOLT_TRY(pushforward_masses(ctx, cols, from_abi(reference_cycle, n_cycle), &op->linear));
    op->squared = op->linear;
    op->kept["masses"] = from_abi(op->linear.data(), op->linear.size());
    const Fr g = fr_from_abi(gamma), g2 = mul(g, g);
    Fr cur = Fr::one();
    op->weights.resize(op->n_polys);
    for (size_t i = 0; i < op->n_polys; ++i) {
        fr_to_abi(&op->weights[i], cur);
        cur = mul(cur, g2);
    }
    JOLT_TRY(host_eq(from_abi(reference_address, log_k), &op->eq));
    *out = hold.release();
    return JOLT_OK;
}

extern "C" int32_t jolt_stage_booleanity_cycle_create(jolt_ctx* ctx, const jolt_onehot* cols, const jolt_fr_t* r_address, const jolt_fr_t* reference_address,
                                                      const jolt_fr_t* reference_cycle, size_t n_cycle, const jolt_fr_t* gamma, jolt_stage_op** out) {
    if (!ctx || !cols || !r_address || !reference_address || (!reference_cycle && n_cycle) || !gamma || !out || ((size_t)1 << n_cycle) != cols->cycles) return JOLT_ERR_INVALID_ARG;
    const size_t log_k = log2_exact(cols->k);
    if (((size_t)1 << log_k) != cols->k || !all_canonical(r_address, log_k) || !all_canonical(reference_address, log_k) || !all_canonical(reference_cycle, n_cycle) ||
        !all_canonical(gamma, 1))
        return JOLT_ERR_INVALID_ARG;
    const Fr g = fr_from_abi(gamma);
    if (g.is_zero()) return JOLT_ERR_INVALID_ARG;  // "booleanity batching gamma must be invertible" (booleanity.rs:497-501)
    BooleanityCycleOp* op = new_op<BooleanityCycleOp>(ctx, "booleanity_cycle");
    if (!op) return JOLT_ERR_OOM;
    std::unique_ptr<BooleanityCycleOp> hold(op);
    op->n_polys = cols->n_polys;
    op->rounds = n_cycle;
    op->degree = 3;
    // the fixed address factor of the EqAddressCycle public rides in the split-eq scaling (booleanity.rs:489-495); the K-sized tables gamma^i * eq(r_address, .)
    const std::vector<Fr> ra = from_abi(r_address, log_k), ref = from_abi(reference_address, log_k);
    Fr scalar = Fr::one();
    for (size_t b = 0; b < log_k; ++b) {  // eq_mle: prod_b (a_b c_b + (1 - a_b)(1 - c_b))
        const Fr ac = mul(ra[b], ref[b]);
        scalar = mul(scalar, add(sub(sub(Fr::one(), ra[b]), ref[b]), add(ac, ac)));
    }
    std::vector<jolt_fr_t> eq_address;
    JOLT_TRY(host_eq(ra, &eq_address));
    const size_t K = cols->k;
    std::vector<jolt_fr_t> tables(op->n_polys * K), rho(op->n_polys);
    op->rho_inv.resize(op->n_polys);
    const Fr g_inv = inv(g);
    Fr cur = Fr::one(), cur_inv = Fr::one();
    for (size_t i = 0; i < op->n_polys; ++i) {
        fr_to_abi(&rho[i], cur);
        op->rho_inv[i] = cur_inv;
        for (size_t k = 0; k < K; ++k) fr_to_abi(&tables[i * K + k], mul(cur, fr_from_abi(&eq_address[k])));
        cur = mul(cur, g);
        cur_inv = mul(cur_inv, g_inv);
    }
    jolt_fr_t sc;
    fr_to_abi(&sc, scalar);
    if (n_cycle < 4) {
        op->dense = true;
        std::vector<TableH> tabs;
        jolt_table* eqt = nullptr;
        JOLT_TRY(jolt_eq_evals(ctx, reference_cycle, n_cycle, &sc, &eqt));
        tabs.emplace_back(ctx, eqt);
        std::vector<std::pair<Fr, std::vector<uint32_t>>> terms;
        for (size_t i = 0; i < op->n_polys; ++i) {
            TableH scale;
            JOLT_TRY(fr_table(ctx, &tables[i * K], K, &scale));
            jolt_table* col = nullptr;
            JOLT_TRY(jolt_onehot_materialize(ctx, cols, i, scale.t, &col));
            tabs.emplace_back(ctx, col);
            const uint32_t t = (uint32_t)(1 + i);
            terms.push_back({Fr::one(), {0u, t, t}});
            terms.push_back({neg(fr_from_abi(&rho[i])), {0u, t}});
        }
        JOLT_TRY(expr_member(ctx, tabs, terms, 3, &op->member));
    } else {
        JOLT_TRY(jolt_member_create_lazy_booleanity(ctx, cols, tables.data(), rho.data(), reference_cycle, n_cycle, &sc, &op->member.m));
    }
    *out = hold.release();
    return JOLT_OK;
}

extern "C" int32_t jolt_stage_hamming_weight_create(jolt_ctx* ctx, const jolt_onehot* cols, const jolt_fr_t* r_cycle, size_t n_cycle, const jolt_fr_t* r_address,
                                                    const jolt_fr_t* virtualization_points, const jolt_fr_t* gamma, jolt_stage_op** out) {
    if (!ctx || !cols || !r_cycle || !r_address || !virtualization_points || !gamma || !out || ((size_t)1 << n_cycle) != cols->cycles) return JOLT_ERR_INVALID_ARG;
    const size_t log_k = log2_exact(cols->k);
    if (((size_t)1 << log_k) != cols->k || !all_canonical(r_cycle, n_cycle)) return JOLT_ERR_INVALID_ARG;
    HammingWeightOp* op = new_op<HammingWeightOp>(ctx, "hamming_weight");
    if (!op) return JOLT_ERR_OOM;
    std::unique_ptr<HammingWeightOp> hold(op);
    op->n_polys = cols->n_polys;
    op->K = op->len = cols->k;
    op->rounds = log_k;
    op->degree = 2;
    // FamilySelectors::pushforwards (hamming_weight_claim_reduction.rs:83-117): all RA columns against ONE eq table
    JOLT_TRY(pushforward_masses(ctx, cols, from_abi(r_cycle, n_cycle), &op->g));
    op->kept["masses"] = from_abi(op->g.data(), op->g.size());
    op->w.assign(op->n_polys * op->K, jolt_fr_t{});
    JOLT_TRY(jolt_host_hamming_weights(gamma, r_address, virtualization_points, op->n_polys, log_k, op->w.data()));
    *out = hold.release();
    return JOLT_OK;
}

extern "C" int32_t jolt_stage_instruction_read_raf_create(jolt_ctx* ctx, jolt_read_raf* rows, const jolt_onehot* claim_columns, const jolt_fr_t* r_reduction, size_t n_vars,
                                                          const jolt_fr_t* gamma, const uint8_t* table_present /* 42 */, uint32_t ra_count, jolt_stage_op** out) {
    if (!ctx || !rows || !claim_columns || !r_reduction || !gamma || !table_present || !out || ra_count == 0 || ra_count + 2 > JOLT_MAX_DEGREE) return JOLT_ERR_INVALID_ARG;
    size_t cycles = 0;
    uint32_t n_tables = 0;
    JOLT_TRY(jolt_read_raf_cycles(rows, &cycles, &n_tables));
    if (((size_t)1 << n_vars) != cycles || n_tables != kLookupTables || claim_columns->cycles != cycles || claim_columns->n_polys < 4 || claim_columns->k != 16 ||
        !all_canonical(r_reduction, n_vars))
        return JOLT_ERR_INVALID_ARG;
    InstructionReadRafOp* op = new_op<InstructionReadRafOp>(ctx, "instruction_read_raf");
    if (!op) return JOLT_ERR_OOM;
    std::unique_ptr<InstructionReadRafOp> hold(op);
    op->rr = rows;
    op->claim_columns = claim_columns;
    op->reduction = from_abi(r_reduction, n_vars);
    op->present.assign(table_present, table_present + kLookupTables);
    op->ra_count = ra_count;
    op->n_vars = n_vars;
    op->n_f = 1 + ra_count;
    op->rounds = kAddressBits + n_vars;
    op->degree = op->n_f + 1;
    op->suffix_offsets.assign(kLookupTables + 1, 0);
    JOLT_TRY(jolt_lookup_suffix_layout(op->suffix_offsets.data(), nullptr));
    op->suffix_kinds.assign(op->suffix_offsets[kLookupTables], 0);
    JOLT_TRY(jolt_lookup_suffix_layout(op->suffix_offsets.data(), op->suffix_kinds.data()));
    JOLT_TRY(jolt_host_read_raf_address_create(gamma, table_present, 0, &op->state));
    JOLT_TRY(eq_table(ctx, op->reduction, &op->u));  // the per-cycle mass eq(r_reduction, j) every phase condenses
    JOLT_TRY(op->open_phase());
    *out = hold.release();
    return JOLT_OK;
}

extern "C" int32_t jolt_stage_bytecode_read_raf_address_create(jolt_ctx* ctx, const jolt_key_index* pc_index, const jolt_fr_t* stage_points /* 5 x n_vars */, size_t n_vars,
                                                               const jolt_fr_t* stage_values /* 5 x K */, const jolt_fr_t* gamma, uint64_t first_pc, uint64_t entry_index,
                                                               jolt_stage_op** out) {
    if (!ctx || !pc_index || (!stage_points && n_vars) || !stage_values || !gamma || !out) return JOLT_ERR_INVALID_ARG;
    size_t cycles = 0;
    uint64_t K = 0;
    uint32_t items = 0;
    JOLT_TRY(jolt_key_index_size(pc_index, &cycles, &K, &items));
    const size_t log_k = log2_exact((size_t)K);
    if (((size_t)1 << n_vars) != cycles || ((uint64_t)1 << log_k) != K || first_pc >= K || entry_index >= K) return JOLT_ERR_INVALID_ARG;
    BytecodeAddressOp* op = new_op<BytecodeAddressOp>(ctx, "bytecode_read_raf_address");
    if (!op) return JOLT_ERR_OOM;
    std::unique_ptr<BytecodeAddressOp> hold(op);
    op->n_vars = n_vars;
    op->log_k = log_k;
    op->entry_index = entry_index;
    op->rounds = log_k;
    op->degree = 2;
    op->gp.assign(1, Fr::one());
    for (int i = 0; i < 7; ++i) op->gp.push_back(mul(op->gp.back(), fr_from_abi(gamma)));
    // stage_pushforwards (bytecode_read_raf.rs:152-237): F_s(k) = sum_{j: pc(j) = k} eq(r_cycle_s, j), the five stages in one walk over the PC index
    std::vector<jolt_table*> eq_handles;
    for (size_t s = 0; s < 5; ++s) {
        TableH eq;
        JOLT_TRY(eq_table(ctx, from_abi(stage_points + s * n_vars, n_vars), &eq));
        eq_handles.push_back(eq.t);
        op->eqs.push_back(std::move(eq));
    }
    jolt_table* F[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    JOLT_TRY(jolt_key_index_pushforward(ctx, pc_index, eq_handles.data(), 5, F));
    std::vector<TableH> tables;
    for (size_t s = 0; s < 5; ++s) tables.emplace_back(ctx, F[s]);
    for (size_t s = 0; s < 5; ++s) {
        TableH v;
        JOLT_TRY(fr_table(ctx, stage_values + s * (size_t)K, (size_t)K, &v));
        tables.push_back(std::move(v));
    }
    std::vector<uint64_t> ident((size_t)K), trace((size_t)K, 0), expected((size_t)K, 0);
    for (size_t k = 0; k < (size_t)K; ++k) ident[k] = k;
    trace[first_pc] = 1;  // the PC of the trace's first cycle
    expected[entry_index] = 1;
    for (const std::vector<uint64_t>* v : {&ident, &trace, &expected}) {
        TableH t;
        JOLT_TRY(u64_table(ctx, *v, &t));
        tables.push_back(std::move(t));
    }
    // AddressKernel's summand (:303-311): sum_s g^s F_s V_s + g^5 F_0 Int + g^6 F_2 Int + g^7 entry_trace entry_expected
    std::vector<std::pair<Fr, std::vector<uint32_t>>> terms;
    for (uint32_t s = 0; s < 5; ++s) terms.push_back({op->gp[s], {s, 5 + s}});
    terms.push_back({op->gp[5], {0, 10}});
    terms.push_back({op->gp[6], {2, 10}});
    terms.push_back({op->gp[7], {11, 12}});
    JOLT_TRY(expr_member(ctx, tables, terms, 2, &op->member));
    *out = hold.release();
    return JOLT_OK;
}

// CycleKernel (:440-690): C(j) * prod_i ra_i(j) with ra_i(j) = eq(chunk_i)[chunk_i(pc_j)] and the combined coefficient column C from the address phase's residue.
// `address` must have finished its rounds; its parked eq tables are consumed here.
extern "C" int32_t jolt_stage_bytecode_read_raf_cycle_create(jolt_ctx* ctx, jolt_stage_op* address, const jolt_onehot* pc_chunks, uint32_t chunk_bits, jolt_stage_op** out) {
    BytecodeAddressOp* adr = dynamic_cast<BytecodeAddressOp*>(address);
    if (!ctx || !adr || !pc_chunks || !out || chunk_bits == 0 || chunk_bits > 8 || adr->fin.size() != 13 || adr->eqs.size() != 5) return JOLT_ERR_INVALID_ARG;
    const size_t n_vars = adr->n_vars, log_k = adr->log_k, n_chunks = (log_k + chunk_bits - 1) / chunk_bits;
    if (pc_chunks->n_polys < n_chunks || pc_chunks->k != (1u << chunk_bits) || pc_chunks->cycles != ((size_t)1 << n_vars) || n_chunks + 1 > JOLT_MAX_DEGREE) return JOLT_ERR_INVALID_ARG;
    ProductOp* op = new_op<ProductOp>(ctx, "bytecode_read_raf_cycle");
    if (!op) return JOLT_ERR_OOM;
    std::unique_ptr<ProductOp> hold(op);
    op->rounds = n_vars;
    op->n_f = 1 + n_chunks;
    op->degree = op->n_f;
    op->first_claim = 1;
    const std::vector<Fr>& fin = adr->fin;
    const std::vector<Fr>& gp = adr->gp;
    const std::vector<Fr> r_address = reversed(adr->binds.data(), log_k);  // LowToHigh rounds: the last challenge is the most significant address bit
    // committed_address_chunks (crates/jolt-claims/src/protocols/jolt/geometry/dimensions.rs:350-366): zero-padded at the FRONT to a multiple of chunk_bits
    std::vector<Fr> padded((chunk_bits - log_k % chunk_bits) % chunk_bits, Fr::zero());
    padded.insert(padded.end(), r_address.begin(), r_address.end());
    std::vector<TableH> tables(1);
    for (size_t i = 0; i < n_chunks; ++i) {
        std::vector<jolt_fr_t> e;
        JOLT_TRY(host_eq(std::vector<Fr>(padded.begin() + i * chunk_bits, padded.begin() + (i + 1) * chunk_bits), &e));
        TableH scale;
        JOLT_TRY(fr_table(ctx, e.data(), e.size(), &scale));
        jolt_table* col = nullptr;
        JOLT_TRY(jolt_onehot_materialize(ctx, pc_chunks, i, scale.t, &col));
        tables.emplace_back(ctx, col);
    }
    const Fr int_r = fin[10];  // IdentityPolynomial(r_address): the bound Int table
    std::vector<Fr> weights;
    for (size_t s = 0; s < 5; ++s) weights.push_back(mul(gp[s], fin[5 + s]));
    weights[0] = add(weights[0], mul(gp[5], int_r));
    weights[2] = add(weights[2], mul(gp[6], int_r));
    std::vector<jolt_fr_t> eq_adr;
    JOLT_TRY(host_eq(r_address, &eq_adr));
    weights.push_back(mul(gp[7], fr_from_abi(&eq_adr[adr->entry_index])));
    TableH spike;  // eq(0, j) = [j = 0]
    JOLT_TRY(eq_table(ctx, std::vector<Fr>(n_vars, Fr::zero()), &spike));
    std::vector<jolt_table*> srcs;
    for (TableH& t : adr->eqs) srcs.push_back(t.t);
    srcs.push_back(spike.t);
    JOLT_TRY(rlc_tables(ctx, srcs, weights, &tables[0]));
    adr->eqs.clear();
    spike.reset();
    std::vector<uint32_t> all;
    for (uint32_t i = 0; i < op->n_f; ++i) all.push_back(i);
    JOLT_TRY(expr_member(ctx, tables, {{Fr::one(), all}}, (uint32_t)op->n_f, &op->member));
    *out = hold.release();
    return JOLT_OK;
}

// optimized/ram_raf_evaluation.rs:17-62: ra_folded = fold_cycles(eq(tau_low)) over the RAM address column, unmap(k) = 8 k + lowest_address, log K rounds
extern "C" int32_t jolt_stage_ram_raf_evaluation_create(jolt_ctx* ctx, const jolt_key_index* ram_index, const jolt_fr_t* tau_low, size_t n_vars, uint64_t lowest_address,
                                                        jolt_stage_op** out) {
    if (!ctx || !ram_index || (!tau_low && n_vars) || !out) return JOLT_ERR_INVALID_ARG;
    size_t cycles = 0;
    uint64_t K = 0;
    uint32_t items = 0;
    JOLT_TRY(jolt_key_index_size(ram_index, &cycles, &K, &items));
    const size_t log_k = log2_exact((size_t)K);
    if (((size_t)1 << n_vars) != cycles || ((uint64_t)1 << log_k) != K || !all_canonical(tau_low, n_vars)) return JOLT_ERR_INVALID_ARG;
    ProductOp* op = new_op<ProductOp>(ctx, "ram_raf_evaluation");
    if (!op) return JOLT_ERR_OOM;
    std::unique_ptr<ProductOp> hold(op);
    op->rounds = log_k;
    op->degree = 2;
    op->n_f = 2;
    op->first_claim = 0;
    TableH eq;
    JOLT_TRY(eq_table(ctx, from_abi(tau_low, n_vars), &eq));
    jolt_table* w[1] = {eq.t};
    jolt_table* folded[1] = {nullptr};
    JOLT_TRY(jolt_key_index_pushforward(ctx, ram_index, w, 1, folded));
    std::vector<TableH> tables;
    tables.emplace_back(ctx, folded[0]);
    eq.reset();
    std::vector<uint64_t> unmap((size_t)K);
    for (size_t k = 0; k < (size_t)K; ++k) unmap[k] = 8 * (uint64_t)k + lowest_address;
    TableH t;
    JOLT_TRY(u64_table(ctx, unmap, &t));
    tables.push_back(std::move(t));
    JOLT_TRY(expr_member(ctx, tables, {{Fr::one(), {0, 1}}}, 2, &op->member));
    *out = hold.release();
    return JOLT_OK;
}

// optimized/ram_output_check.rs:50-215: eq(r_address, k) * io_mask(k) * (val_final(k) - val_io(k)) with the eq factor split (Gruen); val_final = the word every
// address holds after its last access, built from the resident access columns
extern "C" int32_t jolt_stage_ram_output_check_create(jolt_ctx* ctx, const jolt_key_index* ram_index, const jolt_ints* post_values, const uint64_t* val_init,
                                                      const uint64_t* val_io, uint64_t io_lo, uint64_t io_len, const jolt_fr_t* r_address, jolt_stage_op** out) {
    if (!ctx || !ram_index || !post_values || !val_init || !val_io || !r_address || !out) return JOLT_ERR_INVALID_ARG;
    size_t cycles = 0;
    uint64_t K = 0;
    uint32_t items = 0;
    JOLT_TRY(jolt_key_index_size(ram_index, &cycles, &K, &items));
    const size_t log_k = log2_exact((size_t)K);
    if (((uint64_t)1 << log_k) != K || io_lo > K || io_len > K - io_lo || post_values->count != cycles) return JOLT_ERR_INVALID_ARG;
    RamOutputCheckOp* op = new_op<RamOutputCheckOp>(ctx, "ram_output_check");
    if (!op) return JOLT_ERR_OOM;
    std::unique_ptr<RamOutputCheckOp> hold(op);
    op->rounds = op->log_k = log_k;
    op->degree = 3;
    TableH init, io, mask, diff;
    JOLT_TRY(u64_table(ctx, std::vector<uint64_t>(val_init, val_init + K), &init));
    jolt_table* vf = nullptr;
    JOLT_TRY(jolt_key_index_last_value(ctx, ram_index, post_values, init.t, &vf));
    op->val_final = TableH(ctx, vf);
    init.reset();
    JOLT_TRY(u64_table(ctx, std::vector<uint64_t>(val_io, val_io + K), &io));
    JOLT_TRY(rlc_tables(ctx, {op->val_final.t, io.t}, {Fr::one(), sub(Fr::zero(), Fr::one())}, &diff));
    io.reset();
    std::vector<uint64_t> m((size_t)K, 0);
    for (uint64_t k = io_lo; k < io_lo + io_len; ++k) m[k] = 1;
    JOLT_TRY(u64_table(ctx, m, &mask));
    JOLT_TRY(jolt_member_create_split_eq_product(ctx, mask.t, diff.t, r_address, log_k, nullptr, &op->member.m));
    mask.release();
    diff.release();
    *out = hold.release();
    return JOLT_OK;
}

// Test hook (CPU suite): the reference tier's dense member over host tables as a stage operator -- see HostExprOp.  `tables`: n_tables arrays of `len` (a power of two)
// canonical field elements, copied; the descriptor as for jolt_member_create_expr (LowToHigh only).
extern "C" int32_t jolt_stage_host_expr_create(const jolt_fr_t* const* tables, size_t len, const jolt_member_desc* desc, jolt_stage_op** out) {
    if (!tables || !desc || !out || len == 0 || (len & (len - 1)) != 0 || desc->n_tables == 0 || desc->n_tables > JOLT_MAX_MEMBER_TABLES || desc->n_terms == 0 ||
        desc->n_terms > JOLT_MAX_MEMBER_TERMS || desc->degree == 0 || desc->degree > JOLT_MAX_DEGREE || desc->order != JOLT_ORDER_LOW_TO_HIGH || !desc->term_offsets ||
        !desc->factors || !desc->coeffs)
        return JOLT_ERR_INVALID_ARG;
    HostExprOp* op = new_op<HostExprOp>(nullptr, "host_expr");
    if (!op) return JOLT_ERR_OOM;
    std::unique_ptr<HostExprOp> hold(op);
    op->len = len;
    op->rounds = log2_exact(len);
    op->degree = desc->degree;
    op->offs.assign(desc->term_offsets, desc->term_offsets + desc->n_terms + 1);
    if (op->offs.back() > JOLT_MAX_MEMBER_FACTORS) return JOLT_ERR_INVALID_ARG;
    op->facs.assign(desc->factors, desc->factors + op->offs.back());
    for (uint32_t f : op->facs)
        if (f >= desc->n_tables) return JOLT_ERR_INVALID_ARG;
    for (uint32_t k = 0; k < desc->n_terms; ++k) {
        if (op->offs[k + 1] < op->offs[k] || op->offs[k + 1] - op->offs[k] > desc->degree) return JOLT_ERR_INVALID_ARG;
        op->coeffs.push_back(fr_from_abi(&desc->coeffs[k]));
        if (!fr_is_canonical(op->coeffs.back())) return JOLT_ERR_INVALID_ARG;
    }
    for (uint32_t i = 0; i < desc->n_tables; ++i) {
        if (!tables[i] || !all_canonical(tables[i], len)) return JOLT_ERR_INVALID_ARG;
        op->tables.push_back(from_abi(tables[i], len));
    }
    *out = hold.release();
    return JOLT_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// drivers
// ------------------------------------------------------------------------------------------------------------------
// prove_batch (prover.rs:193-362) over stage operators with the library's test transcript: what a stage driver does with the kernels a backend's slots returned.
extern "C" int32_t jolt_host_prove_batch_ops(jolt_ctx* ctx, jolt_stage_op* const* ops, size_t n_ops, const jolt_fr_t* input_claims, const jolt_fr_t* coefficients,
                                             const size_t* offsets, size_t max_num_vars, size_t max_degree, uint64_t transcript_label, int32_t challenge_mode,
                                             jolt_fr_t* out_polys, jolt_fr_t* out_challenges, jolt_fr_t* out_member_claims, jolt_fr_t* out_final_claim) {
    (void)ctx;  // (may be NULL: host-only operators carry none; device-backed ones carry their own)
    if ((!ops && n_ops) || !input_claims || !coefficients || !offsets || !out_polys || !out_challenges || !out_member_claims || !out_final_claim) return JOLT_ERR_INVALID_ARG;
    std::vector<ProveRounds*> ms;
    std::vector<BatchMember> described;
    for (size_t i = 0; i < n_ops; ++i) {
        if (!ops[i]) return JOLT_ERR_INVALID_ARG;
        if (offsets[i] > max_num_vars || ops[i]->rounds > max_num_vars - offsets[i]) return JOLT_ERR_INVALID_ARG;  // WindowOutOfRange (before the prelude scales claims by 2^(max - rounds))
        ms.push_back(ops[i]);
        described.push_back(BatchMember{fr_from_abi(&input_claims[i]), fr_from_abi(&coefficients[i]), ops[i]->rounds, offsets[i]});
    }
    BatchPrelude prelude = BatchPrelude::make(std::move(described), max_num_vars, max_degree);
    LabelledTranscript tr(transcript_label);
    SequentialRounds seq;
    ProvedBatch proved;
    SumcheckError err;
    JOLT_TRY(prove_batch(prelude, ms, seq, tr, challenge_mode != 0, &proved, &err));
    const size_t stride = max_degree + 1;
    const Fr zero = Fr::zero();
    for (size_t r = 0; r < max_num_vars; ++r)
        for (size_t k = 0; k < stride; ++k) fr_to_abi(&out_polys[r * stride + k], k < proved.round_polys[r].coefficients.size() ? proved.round_polys[r].coefficients[k] : zero);
    for (size_t r = 0; r < max_num_vars; ++r) fr_to_abi(&out_challenges[r], proved.challenges[r]);
    for (size_t i = 0; i < n_ops; ++i) fr_to_abi(&out_member_claims[i], proved.member_claims[i]);
    fr_to_abi(out_final_claim, proved.final_claim);
    return JOLT_OK;
}

// ONE operator driven alone, the way the reference's kernel tests drive a ProveRounds (and the way rounds 2-5 drove these operators from Python): per round the
// message, every coefficient absorbed, Transcript::challenge, the running claim = the message at the challenge; finish_rounds with the last challenge.
//   claim: in = the operator's input claim, out = the claim after the last round.  coeffs_out: rounds x stride, a round's unused tail zeroed; n_coeffs_out[r] = the
//   number of coefficients round r's message had.
extern "C" int32_t jolt_host_stage_op_prove_alone(jolt_stage_op* op, jolt_host_transcript* transcript, jolt_fr_t* claim, jolt_fr_t* coeffs_out, size_t stride,
                                                  uint32_t* n_coeffs_out, jolt_fr_t* challenges_out) {
    if (!op || !transcript || !claim) return JOLT_ERR_INVALID_ARG;
    Fr running = fr_from_abi(claim), bind = Fr::zero();
    if (!fr_is_canonical(running)) return JOLT_ERR_INVALID_ARG;
    bool has_bind = false;
    for (size_t round = 0; round < op->rounds; ++round) {
        UnivariatePoly poly;
        JOLT_TRY(op->prove_round(has_bind ? &bind : nullptr, round, running, &poly));
        if (coeffs_out && poly.coefficients.size() > stride) return JOLT_ERR_SIZE_MISMATCH;
        std::vector<jolt_fr_t> c = to_abi(poly.coefficients);
        JOLT_TRY(jolt_host_transcript_append_fr(transcript, c.data(), poly.coefficients.size()));
        jolt_fr_t r;
        JOLT_TRY(jolt_host_transcript_challenge(transcript, 0, &r));
        bind = fr_from_abi(&r);
        has_bind = true;
        running = poly.evaluate(bind);
        if (coeffs_out)
            for (size_t k = 0; k < stride; ++k) coeffs_out[round * stride + k] = k < poly.coefficients.size() ? c[k] : jolt_fr_t{};
        if (n_coeffs_out) n_coeffs_out[round] = (uint32_t)poly.coefficients.size();
        if (challenges_out) challenges_out[round] = r;
    }
    if (has_bind) JOLT_TRY(op->finish_rounds(bind));
    fr_to_abi(claim, running);
    return JOLT_OK;
}
