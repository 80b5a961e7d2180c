// jolt_amd/csrc/batch.hip -- resumable batched-sumcheck round loop for the hypercube-sharded (multi-GPU) prover.
//
// Same algorithm as jolt_sumcheck::prove_batch (crates/jolt-sumcheck/src/prover.rs:193-362) -- see host_mirror.hip for
// the single-process mirror -- restructured so that (a) the per-round local sums can come from ANY local backend (the
// device members through jolt_round_group_prove, or a caller-supplied callback), (b) an all-gather hook adds the
// partial sums of the other ranks before the round message is assembled, and (c) the loop can be paused after the
// rounds that are local to a shard and resumed on the gathered G-entry tables (DESIGN.md
// section 6).  Every rank runs the same transcript on the same summed data, so all ranks draw identical challenges
// without further communication.
#include "host_mirror.hpp"
#include "member.hpp"

using namespace jolt;
using namespace jolt_host;

struct jolt_batch {
    jolt_ctx* ctx = nullptr;
    size_t n = 0, max_num_vars = 0, max_degree = 0, round = 0;
    bool full_width = false;
    LabelledTranscript transcript{0};
    // caller-owned Fiat-Shamir (jolt_host_batch_set_transcript): absorbs the round's compressed polynomial, returns the challenge
    jolt_round_transcript_fn round_transcript = nullptr;
    void* round_transcript_user = nullptr;
    std::vector<BatchMember> described;
    std::vector<int32_t> kind;        // 0 expr, 1 expr with skipped s(1), 2 split-eq product, 3 split-eq uniform product
    std::vector<uint32_t> degree;
    std::vector<std::vector<Fr>> w;   // split-eq: global point (rounds coordinates)
    std::vector<Fr> current_scalar;   // split-eq: GruenSplitEqPolynomial::current_scalar
    std::vector<size_t> bound;        // rounds bound so far per member
    std::vector<Fr> member_claims, pending;
    std::vector<bool> has_pending;
    Fr running_claim;
    std::vector<Fr> challenges;
    std::vector<UnivariatePoly> round_polys;
    size_t n_evals(size_t i) const { return kind[i] == 2 ? 2 : (kind[i] == 3 ? degree[i] - 1 : (kind[i] == 1 ? degree[i] : degree[i] + 1)); }
};

extern "C" int32_t jolt_host_batch_begin(jolt_ctx* ctx, size_t n_members, const jolt_fr_t* input_claims, const jolt_fr_t* coefficients,
                                         const size_t* rounds, const size_t* offsets, const int32_t* kinds, const uint32_t* degrees,
                                         const jolt_fr_t* const* split_eq_points, const jolt_fr_t* split_eq_scales, size_t max_num_vars,
                                         size_t max_degree, uint64_t transcript_label, int32_t challenge_mode, jolt_batch** out) {
    if (!input_claims || !coefficients || !rounds || !offsets || !kinds || !degrees || !out) return JOLT_ERR_INVALID_ARG;
    jolt_batch* b = new (std::nothrow) jolt_batch();
    if (!b) return JOLT_ERR_OOM;
    b->ctx = ctx;
    b->n = n_members;
    b->max_num_vars = max_num_vars;
    b->max_degree = max_degree;
    b->full_width = challenge_mode != 0;
    b->transcript = LabelledTranscript(transcript_label);
    b->running_claim = Fr::zero();
    for (size_t i = 0; i < n_members; ++i) {
        if (offsets[i] + rounds[i] > max_num_vars || degrees[i] < 1 || degrees[i] > max_degree) { delete b; return JOLT_ERR_INVALID_ARG; }
        b->described.push_back(BatchMember{fr_from_abi(&input_claims[i]), fr_from_abi(&coefficients[i]), rounds[i], offsets[i]});
        b->kind.push_back(kinds[i]);
        b->degree.push_back(degrees[i]);
        std::vector<Fr> w;
        Fr scalar = Fr::one();
        if (kinds[i] >= 2) {
            if (!split_eq_points || !split_eq_points[i]) { delete b; return JOLT_ERR_INVALID_ARG; }
            for (size_t k = 0; k < rounds[i]; ++k) w.push_back(fr_from_abi(&split_eq_points[i][k]));
            if (split_eq_scales) scalar = fr_from_abi(&split_eq_scales[i]);
        }
        b->w.push_back(std::move(w));
        b->current_scalar.push_back(scalar);
        b->bound.push_back(0);
        Fr padded = fr_mul_pow_2(b->described[i].input_claim, max_num_vars - rounds[i]);  // prover.rs:244-248
        b->member_claims.push_back(padded);
        b->running_claim = add(b->running_claim, mul(b->described[i].coefficient, padded));  // batch.rs:57-64
        b->pending.push_back(Fr::zero());
        b->has_pending.push_back(false);
    }
    *out = b;
    return JOLT_OK;
}

// local sums of the active members for one round: evals_out is the concatenation in `active` order
typedef int32_t (*jolt_local_round_fn)(void* user, const size_t* active, size_t n_active, const jolt_fr_t* const* binds, jolt_fr_t* evals_out,
                                       size_t evals_count);
typedef int32_t (*jolt_gather_fn)(void* user, const jolt_fr_t* local, size_t count, jolt_fr_t* gathered);
typedef int32_t (*jolt_round_transcript_fn)(void* user, const jolt_fr_t* compressed_coeffs, size_t n_coeffs, jolt_fr_t* challenge_out);

static void note_bind(jolt_batch* b, size_t i, const Fr& c) {
    if (b->kind[i] >= 2) {  // split_eq.rs:334-337
        size_t current_index = b->described[i].rounds - b->bound[i];
        Fr p = b->w[i][current_index - 1];
        Fr prod = mul(p, c);
        b->current_scalar[i] = mul(b->current_scalar[i], add(add(sub(sub(Fr::one(), p), c), prod), prod));
    }
    b->bound[i] += 1;
}

static int32_t assemble(const jolt_batch* b, size_t i, const Fr* ev, const Fr& claim, UnivariatePoly* out, const Fr* inv_l1) {
    if (b->kind[i] == 2) {
        size_t current_index = b->described[i].rounds - b->bound[i];
        return gruen_poly_deg_3(b->current_scalar[i], b->w[i][current_index - 1], ev[0], ev[1], claim, out, inv_l1);
    }
    if (b->kind[i] == 3) {
        size_t current_index = b->described[i].rounds - b->bound[i];
        return gruen_poly_from_q(b->current_scalar[i], b->w[i][current_index - 1], ev, b->degree[i] - 1, claim, out, inv_l1);
    }
    std::vector<Fr> full;
    if (b->kind[i] == 1) {
        full.push_back(ev[0]);
        full.push_back(sub(claim, ev[0]));
        for (uint32_t t = 1; t < b->degree[i]; ++t) full.push_back(ev[t]);
    } else {
        full.assign(ev, ev + b->degree[i] + 1);
        if (add(full[0], full[1]) != claim) return JOLT_ERR_ROUND_CHECK;
    }
    *out = UnivariatePoly::from_evals(full.data(), full.size());
    return JOLT_OK;
}

// Run the next `n_rounds` rounds.  Local sums come from `members` (device members, jolt_round_group_prove) when
extern "C" int32_t jolt_host_batch_set_transcript(jolt_batch* b, jolt_round_transcript_fn fn, void* user) {
    if (!b) return JOLT_ERR_INVALID_ARG;
    b->round_transcript = fn;
    b->round_transcript_user = user;
    return JOLT_OK;
}

// local_fn is NULL, else from the callback.  With world > 1 the local sums of all ranks are gathered through `gather`
// and added (RCCL has no mod-r reduction: all-gather of a few KiB + local modular sum).
extern "C" int32_t jolt_host_batch_run(jolt_batch* b, jolt_member* const* members, size_t n_rounds, int32_t world, jolt_gather_fn gather,
                                       jolt_local_round_fn local_fn, void* user) {
    if (!b || (!members && !local_fn) || world < 1 || (world > 1 && !gather)) return JOLT_ERR_INVALID_ARG;
    static const Fr two_inv = inv(fr_from_u64(2));
    for (size_t step = 0; step < n_rounds; ++step) {
        if (b->round >= b->max_num_vars) return JOLT_ERR_INVALID_ARG;
        const size_t round = b->round;
        std::vector<Fr> batched(b->max_degree + 1, Fr::zero());
        std::vector<size_t> active;
        for (size_t i = 0; i < b->n; ++i) {
            const BatchMember& d = b->described[i];
            if (round >= d.offset && round < d.offset + d.rounds) { active.push_back(i); continue; }
            b->member_claims[i] = mul(b->member_claims[i], two_inv);  // prover.rs:273-282
            batched[0] = add(batched[0], mul(d.coefficient, b->member_claims[i]));
        }
        size_t total = 0;
        std::vector<jolt_fr_t> bind_store(active.size());
        std::vector<const jolt_fr_t*> binds(active.size(), nullptr);
        for (size_t a = 0; a < active.size(); ++a) {
            size_t i = active[a];
            total += b->n_evals(i);
            if (b->has_pending[i]) {
                fr_to_abi(&bind_store[a], b->pending[i]);
                binds[a] = &bind_store[a];
                note_bind(b, i, b->pending[i]);
                b->has_pending[i] = false;
            }
        }
        std::vector<jolt_fr_t> local(total ? total : 1);
        // the inversion of every split-eq message depends only on the challenge already noted: done while the device works
        std::vector<Fr> l1(active.size()), inv_l1(active.size());
        std::vector<char> has_l1(active.size(), 0);
        for (size_t a = 0; a < active.size(); ++a) {
            size_t i = active[a];
            if (b->kind[i] < 2 || b->bound[i] >= b->described[i].rounds) continue;
            l1[a] = mul(b->current_scalar[i], b->w[i][b->described[i].rounds - b->bound[i] - 1]);
            has_l1[a] = l1[a].is_zero() ? 0 : 1;
        }
        const std::function<void()> overlap = [&]() {
            for (size_t a = 0; a < active.size(); ++a)
                if (has_l1[a]) inv_l1[a] = inv(l1[a]);
        };
        if (local_fn) {
            // the sums come from the callback, not from the context's last device round: a gather hook must not mistake a stale
            // device mirror of the same size for them (jolt_comm_gather_round_sums sends ctx->d_round when it mirrors `local`)
            if (b->ctx) b->ctx->d_round_count = 0;
            JOLT_TRY(local_fn(user, active.data(), active.size(), binds.data(), local.data(), total));
            overlap();
        } else {
            std::vector<jolt_member*> ms;
            for (size_t i : active) ms.push_back(members[i]);
            JOLT_TRY(jolt_internal_round_group_prove(b->ctx, ms.data(), ms.size(), binds.data(), local.data(), total, &overlap));
        }
        std::vector<Fr> sums(total);
        for (size_t k = 0; k < total; ++k) sums[k] = fr_from_abi(&local[k]);
        if (gather) {  // also with world == 1 when the caller asks for it (exercises the exchange on a one-GPU box)
            std::vector<jolt_fr_t> gathered((size_t)world * total);
            JOLT_TRY(gather(user, local.data(), total, gathered.data()));
            for (size_t k = 0; k < total; ++k) {
                Fr s = Fr::zero();
                for (int r = 0; r < world; ++r) s = add(s, fr_from_abi(&gathered[(size_t)r * total + k]));
                sums[k] = s;
            }
        }
        std::vector<UnivariatePoly> msgs(active.size());
        size_t off = 0;
        for (size_t a = 0; a < active.size(); ++a) {
            size_t i = active[a];
            JOLT_TRY(assemble(b, i, sums.data() + off, b->member_claims[i], &msgs[a], has_l1[a] ? &inv_l1[a] : nullptr));
            off += b->n_evals(i);
            if (msgs[a].degree() > b->max_degree) return JOLT_ERR_UNSUPPORTED;
            for (size_t k = 0; k < msgs[a].coefficients.size(); ++k)
                batched[k] = add(batched[k], mul(b->described[i].coefficient, msgs[a].coefficients[k]));
        }
        while (batched.size() > 2 && batched.back().is_zero()) batched.pop_back();
        UnivariatePoly poly;
        poly.coefficients = batched;
        if (add(poly.evaluate(Fr::zero()), poly.evaluate(Fr::one())) != b->running_claim) return JOLT_ERR_ROUND_CHECK;
        Fr challenge;
        if (b->round_transcript) {  // the host's real transcript (spongefish in the reference): compressed poly in, challenge out
            std::vector<jolt_fr_t> compressed;
            compressed.reserve(poly.coefficients.size());
            for (size_t k = 0; k < poly.coefficients.size(); ++k) {
                if (k == 1) continue;  // recorder.rs:118-130: the linear term is omitted
                jolt_fr_t c;
                fr_to_abi(&c, poly.coefficients[k]);
                compressed.push_back(c);
            }
            jolt_fr_t ch;
            JOLT_TRY(b->round_transcript(b->round_transcript_user, compressed.data(), compressed.size(), &ch));
            challenge = fr_from_abi(&ch);
            JOLT_REQUIRE(b->ctx, fr_is_canonical(challenge), "transcript callback returned a non-canonical challenge");
        } else {
            b->transcript.append_round_poly(kSumcheckRoundLabel, poly.coefficients.data(), poly.coefficients.size());
            challenge = b->full_width ? b->transcript.challenge_scalar() : b->transcript.challenge();
        }
        b->running_claim = poly.evaluate(challenge);
        b->challenges.push_back(challenge);
        b->round_polys.push_back(poly);
        for (size_t a = 0; a < active.size(); ++a) {
            size_t i = active[a];
            b->member_claims[i] = msgs[a].evaluate(challenge);
            b->pending[i] = challenge;
            b->has_pending[i] = true;
        }
        b->round += 1;
    }
    return JOLT_OK;
}

// Deliver every pending bind (ProveRounds::finish_rounds at the end of the batch, or the phase switch of a sharded
// batch: the local tables must carry the last local challenge before their single remaining entries are gathered).
extern "C" int32_t jolt_host_batch_flush_binds(jolt_batch* b, jolt_member* const* members, jolt_fr_t* binds_out, int32_t* has_bind_out) {
    if (!b) return JOLT_ERR_INVALID_ARG;
    std::vector<jolt_member*> ms;
    std::vector<jolt_fr_t> store(b->n);
    std::vector<const jolt_fr_t*> ptrs;
    for (size_t i = 0; i < b->n; ++i) {
        if (has_bind_out) has_bind_out[i] = b->has_pending[i] ? 1 : 0;
        if (!b->has_pending[i]) continue;
        fr_to_abi(&store[i], b->pending[i]);
        if (binds_out) binds_out[i] = store[i];
        note_bind(b, i, b->pending[i]);
        b->has_pending[i] = false;
        if (members) { ms.push_back(members[i]); ptrs.push_back(&store[i]); }
    }
    if (members && !ms.empty()) return jolt_round_group_finish(b->ctx, ms.data(), ms.size(), ptrs.data());
    return JOLT_OK;
}

extern "C" int32_t jolt_host_batch_split_eq_scalar(const jolt_batch* b, size_t member, jolt_fr_t* out) {
    if (!b || member >= b->n || !out) return JOLT_ERR_INVALID_ARG;
    fr_to_abi(out, b->current_scalar[member]);
    return JOLT_OK;
}

extern "C" int32_t jolt_host_batch_end(jolt_batch* b, jolt_fr_t* out_polys, jolt_fr_t* out_challenges, jolt_fr_t* out_member_claims,
                                       jolt_fr_t* out_final_claim) {
    if (!b) return JOLT_ERR_INVALID_ARG;
    const size_t stride = b->max_degree + 1;
    Fr zero = Fr::zero();
    if (out_polys)
        for (size_t r = 0; r < b->round_polys.size(); ++r)
            for (size_t k = 0; k < stride; ++k)
                fr_to_abi(&out_polys[r * stride + k], k < b->round_polys[r].coefficients.size() ? b->round_polys[r].coefficients[k] : zero);
    if (out_challenges)
        for (size_t r = 0; r < b->challenges.size(); ++r) fr_to_abi(&out_challenges[r], b->challenges[r]);
    if (out_member_claims)
        for (size_t i = 0; i < b->n; ++i) fr_to_abi(&out_member_claims[i], b->member_claims[i]);
    if (out_final_claim) fr_to_abi(out_final_claim, b->running_claim);
    delete b;
    return JOLT_OK;
}
