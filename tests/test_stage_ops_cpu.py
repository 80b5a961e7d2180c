"""The stage-operator CONTRACT and its two drivers on the CPU (no device): jolt_stage_op_{num_rounds, degree, input_claim, prove_round, finish_rounds, output_claims, window},
jolt_host_prove_batch_ops (prove_batch, crates/jolt-sumcheck/src/prover.rs:193-362, over operators) and jolt_host_stage_op_prove_alone, exercised through the host-only
dense-member operator (jolt_stage_host_expr_create: NaiveSumcheckProver over host tables) against the oracle's prove_batch over the same members -- the device-backed operators
of jolt_amd/csrc/stage_ops.hip go through exactly these code paths in the GPU suite (tests/test_gpu_extended*.py)."""
import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import ffi
from util import rand_fr


def members(seed, n_vars, shape):
    """(host operator, oracle member, degree) over the same random tables; shape: list of terms as lists of table indices"""
    n_tables = 1 + max(i for t in shape for i in t)
    tabs = [rand_fr(1 << n_vars, seed + k) for k in range(n_tables)]
    terms = [(rand_fr(1, seed + 50 + k)[0], list(t)) for k, t in enumerate(shape)]
    degree = max(len(t) for t in shape)
    return ffi.stage_host_expr(tabs, terms, degree), O.Member.expr(tabs, terms, degree), degree, tabs, terms


@pytest.mark.parametrize("n_vars,shape", [(1, [[0, 1]]), (5, [[0, 1], [0, 2]]), (7, [[0, 1, 2], [3]]), (4, [[0], [1, 1]])])
def test_one_operator_in_a_batch_is_the_oracle_batch(n_vars, shape):
    op, orc, degree, _, _ = members(100 + n_vars, n_vars, shape)
    assert op.rounds == n_vars and op.degree == degree
    claim = op.input_claim()
    assert np.array_equal(claim, orc.input_claim())
    one = O.to_mont([1])[0]
    got = ffi.prove_batch_ops([op], [claim], [one], [0], n_vars, degree, label=3)
    want = O.prove_batch([orc], [claim], [one], [0], n_vars, degree, label=3)
    for key in ("polys", "challenges", "member_claims", "final_claim"):
        assert np.array_equal(got[key], want[key]), key
    assert np.array_equal(np.stack(op.output_claims()), orc.final_values())
    op.destroy()


def test_a_batch_of_operators_with_different_round_counts_is_the_oracle_batch():
    """three members of 6, 4 and 6 rounds, degrees 2 / 3 / 1, random batching coefficients, the short one in a later window: prover.rs:244-282 (inactive rounds halve the
    member's claim and contribute a constant)"""
    def fresh():
        a, b, c = members(11, 6, [[0, 1], [2, 3]]), members(12, 4, [[0, 1, 2]]), members(13, 6, [[0]])
        return [a[0], b[0], c[0]], [a[1], b[1], c[1]]
    ops, orcs = fresh()
    claims = [o.input_claim() for o in ops]
    coeffs = rand_fr(3, 99)
    got = ffi.prove_batch_ops(ops, claims, coeffs, [0, 2, 0], 6, 3, label=8)  # the short member's window ends with the batch (offset = max_num_vars - rounds)
    want = O.prove_batch(orcs, claims, coeffs, [0, 2, 0], 6, 3, label=8)
    for key in ("polys", "challenges", "member_claims", "final_claim"):
        assert np.array_equal(got[key], want[key]), key
    acc = np.zeros(4, dtype=np.uint64)
    for cf, mc in zip(coeffs, got["member_claims"]):
        acc = O.fr_add(acc, O.fr_mul(cf, mc))
    assert np.array_equal(np.asarray(acc).reshape(4), got["final_claim"])
    for o in ops:
        o.destroy()
    # a window that does not end with the batch hands the member a claim scaled by 2^(max - rounds - offset): its own round check refuses it, on both sides
    ops, orcs = fresh()
    with pytest.raises(ffi.JoltError) as e:
        ffi.prove_batch_ops(ops, claims, coeffs, [0, 0, 0], 6, 3, label=8)
    assert e.value.status == 8  # JOLT_ERR_ROUND_CHECK
    with pytest.raises(RuntimeError):
        O.prove_batch(orcs, claims, coeffs, [0, 0, 0], 6, 3, label=8)
    for o in ops:
        o.destroy()


def test_alone_driver_and_round_windows():
    """jolt_host_stage_op_prove_alone absorbs every coefficient and binds with the transcript's challenge; an operator driven as two windows (jolt_stage_op_window: the
    first window's last challenge is carried into the second's first round) sends the same messages as the operator driven whole under the same transcript"""
    n_vars, shape = 6, [[0, 1], [2]]
    whole = members(21, n_vars, shape)[0]
    claim = whole.input_claim()
    tr = ffi.HostTranscript(4)
    w = whole.prove_alone(tr, claim)
    tr.close()
    assert len(w["polys"]) == n_vars and all(p.shape == (3, 4) for p in w["polys"])
    # the messages satisfy the round check against the running claim, which the driver recomputes from the transcript's challenges
    running = claim
    for poly, r in zip(w["polys"], w["challenges"]):
        at0, at1 = (np.asarray(O.univariate_evaluate(poly, x)).reshape(1, 4) for x in (np.zeros(4, dtype=np.uint64), O.to_mont([1])[0]))
        assert np.array_equal(np.asarray(O.fr_add(at0, at1)).reshape(4), np.asarray(running).reshape(4))
        running = np.asarray(O.univariate_evaluate(poly, r)).reshape(4)
    assert np.array_equal(running, w["final_claim"])
    parent = members(21, n_vars, shape)[0]
    first, second = parent.window(0, 2), parent.window(2, n_vars - 2)
    tr = ffi.HostTranscript(4)
    f = first.prove_alone(tr, claim)
    s = second.prove_alone(tr, f["final_claim"])
    tr.close()
    assert all(np.array_equal(x, y) for x, y in zip(f["polys"] + s["polys"], w["polys"]))
    assert np.array_equal(np.concatenate([f["challenges"], s["challenges"]]), w["challenges"]) and np.array_equal(s["final_claim"], w["final_claim"])
    assert np.array_equal(np.stack(parent.output_claims()), np.stack(whole.output_claims()))
    for o in (first, second, parent, whole):
        o.destroy()


def test_contract_errors():
    """a wrong running claim is JOLT_ERR_ROUND_CHECK (the reference kernels' hard self-check, naive.rs:298-306); output claims before the last bind are
    JOLT_ERR_NOT_FULLY_BOUND; a window beyond the rounds, a non-canonical bind and a batch that disagrees with the operator's rounds are refused"""
    op, _, degree, tabs, terms = members(31, 4, [[0, 1]])
    claim = op.input_claim()
    with pytest.raises(ffi.JoltError) as e:
        op.prove_round(None, 0, O.fr_add(claim, O.to_mont([1])[0]))
    assert e.value.status == 8  # JOLT_ERR_ROUND_CHECK
    with pytest.raises(ffi.JoltError):
        op.output_claims()
    with pytest.raises(ffi.JoltError):
        op.window(3, 2)
    bad = np.full(4, 2**64 - 1, dtype=np.uint64)
    poly = op.prove_round(None, 0, claim)
    assert poly.shape == (degree + 1, 4)
    with pytest.raises(ffi.JoltError):
        op.prove_round(bad, 1, claim)
    with pytest.raises(ffi.JoltError):
        ffi.prove_batch_ops([op], [claim], [O.to_mont([1])[0]], [0], 3, degree)  # BatchMemberRoundsMismatch / WindowOutOfRange
    with pytest.raises(ffi.JoltError):
        ffi.stage_host_expr(tabs, [(terms[0][0], [0, 5])], 2)  # a factor that names no table
    op.destroy()


@pytest.mark.parametrize("kind", [ffi.TRANSCRIPT_BLAKE2B, ffi.TRANSCRIPT_KECCAK, ffi.TRANSCRIPT_BLAKE2B_SPONGE])
def test_a_batch_under_the_reference_transcripts_is_the_oracle_batch(kind):
    """prove_batch over operators with the engine in the label's two top bits (LegacyBlake2bTranscript, KeccakTranscript, the Blake2b512 sponge): the library's C++ transcripts
    and the oracle's C ones draw the same challenges from the same messages -- and not the test transcript's"""
    a, b = members(21, 6, [[0, 1], [2, 3]]), members(22, 4, [[0, 1, 2]])
    ops, orcs = [a[0], b[0]], [a[1], b[1]]
    claims = [o.input_claim() for o in ops]
    coeffs = rand_fr(2, 77)
    want = O.prove_batch(orcs, claims, coeffs, [0, 2], 6, 3, label=kind | 5)
    got = ffi.prove_batch_ops(ops, claims, coeffs, [0, 2], 6, 3, label=kind | 5)
    for key in ("polys", "challenges", "member_claims", "final_claim"):
        assert np.array_equal(got[key], want[key]), key
    for o in ops:
        o.destroy()
    ops = [members(21, 6, [[0, 1], [2, 3]])[0], members(22, 4, [[0, 1, 2]])[0]]
    other = ffi.prove_batch_ops(ops, claims, coeffs, [0, 2], 6, 3, label=5)
    assert not np.array_equal(other["challenges"], got["challenges"])
    for o in ops:
        o.destroy()


def test_a_host_proof_replays_under_a_hashlib_verifier():
    """the bytes the reference's verifier reads (crates/jolt-sumcheck/src/recorder.rs:118-130, round_proof.rs:129-143, crates/jolt-transcript/src/digest.rs:84-189): from the
    compressed round polynomials and the claimed sum alone, a verifier written here over hashlib.blake2b recovers each linear coefficient, absorbs
    LabelWithCount("sumcheck_poly", d) + d big-endian coefficients, squeezes the 125-bit challenge, and arrives at the prover's challenges and final claim -- no GPU"""
    import hashlib
    R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    to_int = lambda x: int.from_bytes(O.fr_to_bytes_le(x), "little")
    op, _, degree, _, _ = members(31, 7, [[0, 1, 2], [3]])
    claim0 = op.input_claim()
    coeff = rand_fr(1, 5)[0]
    label = ffi.TRANSCRIPT_BLAKE2B | 4242
    out = ffi.prove_batch_ops([op], [claim0], [coeff], [0], 7, degree, label=label)
    op.destroy()
    claim = to_int(coeff) * to_int(claim0) % R
    state, n_rounds = hashlib.blake2b(b"jolt-amd/4242".ljust(32, b"\0"), digest_size=32).digest(), 0
    inv_r = pow(2**256, -1, R)
    for r in range(7):
        coeffs = [to_int(c) for c in out["polys"][r]]
        while len(coeffs) > 2 and coeffs[-1] == 0:
            coeffs.pop()
        stored = [coeffs[0]] + coeffs[2:]
        assert (claim - 2 * stored[0] - sum(stored[1:])) % R == coeffs[1], r
        for payload in [b"sumcheck_poly".ljust(24, b"\0") + len(stored).to_bytes(8, "big")] + [c.to_bytes(32, "big") for c in stored]:
            state, n_rounds = hashlib.blake2b(state + bytes(28) + n_rounds.to_bytes(4, "big") + payload, digest_size=32).digest(), n_rounds + 1
        state, n_rounds = hashlib.blake2b(state + bytes(28) + n_rounds.to_bytes(4, "big"), digest_size=32).digest(), n_rounds + 1
        v = int.from_bytes(state[:16], "little")
        challenge = (((v & (2**64 - 1)) << 128) | (((v >> 64) & (2**61 - 1)) << 192)) * inv_r % R
        assert challenge == to_int(out["challenges"][r]), r
        claim = sum(c * pow(challenge, k, R) for k, c in enumerate(coeffs)) % R
    assert claim == to_int(out["final_claim"])
