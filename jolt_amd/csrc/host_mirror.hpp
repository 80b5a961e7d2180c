// jolt_amd/csrc/host_mirror.hpp -- host-side mirror of the reference interfaces that CALL the hot path.
//
// The reference's host code is Rust and stays Rust in a real deployment; this image has no Rust
// toolchain, so the callers are restated in C++ ABOVE the C ABI of include/jolt_hip.h, with the reference's names,
// argument meaning and error behaviour, so that the parity tests read like the reference's own:
//   ProveRounds / RoundScheduler / SequentialRounds / prove_batch   crates/jolt-sumcheck/src/prover.rs:52-362
//   BatchMember / BatchPrelude                                       crates/jolt-sumcheck/src/batch.rs:23-72
//   UnivariatePoly::{from_evals, evaluate, compress}                 crates/jolt-poly/src/univariate.rs
//   GruenSplitEqPolynomial::gruen_poly_deg_3                         crates/jolt-poly/src/split_eq.rs:383-417
//   Transcript                                                       crates/jolt-transcript/src/legacy.rs:55-100
#pragma once
#include <functional>
#include <memory>
#include <vector>

#include "ctx.hpp"

// jolt_round_group_prove with a host callback that runs after the round has been enqueued and before its sums are awaited
// (capi.hip): work that does not depend on the sums overlaps with the device.
int32_t jolt_internal_round_group_prove(jolt_ctx* ctx, jolt_member* const* members, size_t n, const jolt_fr_t* const* binds, jolt_fr_t* evals_out,
                                        size_t cap, const std::function<void()>* overlap);

namespace jolt_host {

using jolt::Fr;

struct UnivariatePoly {
    std::vector<Fr> coefficients;
    static UnivariatePoly from_evals(const Fr* evals, size_t n);  // univariate.rs:198-202
    Fr evaluate(const Fr& x) const;
    size_t degree() const;  // index of the last stored coefficient (univariate.rs `degree`)
};

// SumcheckError (crates/jolt-sumcheck/src/error.rs) as status codes of include/jolt_hip.h
struct SumcheckError {
    int32_t status = JOLT_OK;
    size_t round = 0;
};

// Fiat-Shamir surface the path needs (append bytes / squeeze a challenge); the real Blake2b/Keccak transcript of
// crates/jolt-transcript plugs in here.
struct Transcript {
    virtual ~Transcript() = default;
    virtual void append_bytes(const uint8_t* b, size_t n) = 0;
    virtual Fr challenge() = 0;         // Transcript::challenge: 16 bytes -> from_challenge_bytes (125-bit shape)
    virtual Fr challenge_scalar() = 0;  // Transcript::challenge_scalar: 16 bytes -> from_scalar_challenge_bytes
    void append_fr(const Fr& v);        // canonical 32-byte LE (mod.rs:116-123)
};

// Deterministic test transcript; re-implemented from the SPEC comment in oracle/mock_transcript.h.
struct MockTranscript final : Transcript {
    uint64_t s[4];
    explicit MockTranscript(uint64_t label);
    void append_bytes(const uint8_t* b, size_t n) override;
    Fr challenge() override;
    Fr challenge_scalar() override;

   private:
    void absorb_word(uint64_t w);
    void draw16(uint8_t out[16]);
};

// prover.rs:52-72
struct ProveRounds {
    virtual ~ProveRounds() = default;
    virtual size_t num_rounds() const = 0;
    virtual int32_t prove_round(const Fr* bind, size_t round, const Fr& previous_claim, UnivariatePoly* out) = 0;
    virtual int32_t finish_rounds(const Fr& bind) = 0;
};

// A batch member whose tables live on the GPU: calls jolt_member_prove_round / jolt_member_finish and assembles the
// round message exactly as the reference kernels do (from_evals + round check, naive.rs:298-309; gruen_poly_deg_3 for
// the split-eq member, ram_hamming_booleanity.rs:128-135).
struct DeviceMember final : ProveRounds {
    jolt_member* m;
    explicit DeviceMember(jolt_member* member) : m(member) {}
    size_t num_rounds() const override;
    int32_t prove_round(const Fr* bind, size_t round, const Fr& previous_claim, UnivariatePoly* out) override;
    int32_t finish_rounds(const Fr& bind) override;
    // message assembly from the device sums (shared with the grouped scheduler)
    int32_t assemble(const Fr* evals, const Fr& previous_claim, UnivariatePoly* out, const Fr* inv_l1 = nullptr) const;
    // l(1) = scalar * w_i of the NEXT round message, given the bind that the round applies first (split-eq members)
    bool next_l1(bool has_bind, const Fr& bind, Fr* l1) const;
    size_t n_evals() const;
};

// prover.rs:74-107
struct MemberRound {
    size_t index, local_round;
    bool has_bind;
    Fr bind;
    Fr claim;
    ProveRounds* member;
    bool has_message;
    UnivariatePoly message;
};
struct MemberFinish {
    Fr bind;
    ProveRounds* member;
};
// prover.rs:110-120
struct RoundScheduler {
    virtual ~RoundScheduler() = default;
    virtual int32_t batch_prove_round(std::vector<MemberRound>& work) = 0;
    virtual int32_t batch_finish_rounds(std::vector<MemberFinish>& finishes) = 0;
};
struct SequentialRounds final : RoundScheduler {  // prover.rs:124-146
    int32_t batch_prove_round(std::vector<MemberRound>& work) override;
    int32_t batch_finish_rounds(std::vector<MemberFinish>& finishes) override;
};
// What BuildRoundScheduler (crates/jolt-kernels/src/backend.rs:68-70) would return for this backend: all device members
// of a round are enqueued back to back and fetched with ONE copy + ONE sync (jolt_round_group_prove).
struct DeviceGroupedRounds final : RoundScheduler {
    jolt_ctx* ctx;
    explicit DeviceGroupedRounds(jolt_ctx* c) : ctx(c) {}
    int32_t batch_prove_round(std::vector<MemberRound>& work) override;
    int32_t batch_finish_rounds(std::vector<MemberFinish>& finishes) override;
};

// batch.rs:23-72
struct BatchMember {
    Fr input_claim, coefficient;
    size_t rounds, offset;
};
struct BatchPrelude {
    std::vector<BatchMember> members;
    Fr claimed_sum;
    size_t max_num_vars, max_degree;
    static BatchPrelude make(std::vector<BatchMember> members, size_t max_num_vars, size_t max_degree);
};
struct ProvedBatch {  // prover.rs:152-157
    std::vector<Fr> challenges;
    Fr final_claim;
    std::vector<Fr> member_claims;
    std::vector<UnivariatePoly> round_polys;  // what ClearSumcheckRecorder keeps (recorder.rs:126)
};

// prover.rs:193-362.  `full_width_challenges` selects challenge_scalar() (test hook for the non-shifted bind path).
int32_t prove_batch(const BatchPrelude& prelude, std::vector<ProveRounds*>& members, RoundScheduler& scheduler, Transcript& transcript,
                    bool full_width_challenges, ProvedBatch* out, SumcheckError* err);

Fr fr_mul_pow_2(Fr a, size_t k);
// `inv_l1` (optional): 1 / (current_scalar * point_i), computed by the caller while the device was busy with the round
// (the one field inversion of the message assembly; it does not depend on the round sums).  Verified before use.
int32_t gruen_poly_deg_3(const Fr& current_scalar, const Fr& point_i, const Fr& q_constant, const Fr& q_quadratic, const Fr& s0_plus_s1,
                         UnivariatePoly* out, const Fr* inv_l1 = nullptr);
int32_t gruen_poly_from_q(const Fr& current_scalar, const Fr& point_i, const Fr* q_evals, size_t dq, const Fr& s0_plus_s1, UnivariatePoly* out,
                          const Fr* inv_l1 = nullptr);
Fr inverse_or_given(const Fr& x, const Fr* given);
void fr_to_bytes_le(const Fr& a, uint8_t out[32]);
Fr fr_from_challenge_bytes(const uint8_t* b, size_t n);
Fr fr_from_scalar_challenge_bytes(const uint8_t* b, size_t n);

}  // namespace jolt_host
