"""The hot path under the REFERENCE's Fiat-Shamir transcripts (the engine rides in the two top bits of every transcript label: include/jolt_hip.h): the catalogue's
batched sumchecks, the stage operators and the HyperKZG opening on the device against the oracle, with challenges drawn from LegacyBlake2bTranscript (the transcript
of the reference's benchmark profile, crates/jolt-prover/src/profile.rs:69) and KeccakTranscript -- and a sumcheck proof replayed by a verifier written here over
hashlib alone, the way the reference's verifier reads it (crates/jolt-sumcheck/src/recorder.rs:118-130, round_proof.rs:129-143, digest.rs:84-189)."""
import hashlib

import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import ffi
from jolt_amd.stages import DeviceExtended
from jolt_amd.workload import DeviceWorkload
from test_gpu_extended import same
from test_gpu_msm import same_point
from util import rand_challenge, rand_fr
from workload_oracle import OracleExtended, OracleWorkload

pytestmark = pytest.mark.gpu
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
KINDS = [ffi.TRANSCRIPT_BLAKE2B, ffi.TRANSCRIPT_KECCAK]


def fr_int(a):
    return int.from_bytes(O.fr_to_bytes_le(a), "little")


@pytest.mark.parametrize("kind", KINDS)
def test_catalogue_stages_under_the_reference_transcripts(kind):
    ctx = ffi.Context(0)
    dev, orc = DeviceWorkload(ctx, 9, seed=11), OracleWorkload(9, seed=11)
    want, got = orc.prove(label=kind | 100), dev.prove(label=kind | 100)
    other = dev.prove(label=100)  # the test transcript draws other challenges from the same messages
    for stage in want:
        for key in ("polys", "challenges", "final_claim"):
            assert np.array_equal(got[stage][key], want[stage][key]), (stage, key)
        assert not np.array_equal(got[stage]["challenges"], other[stage]["challenges"])
    dev.close()
    ctx.close()


def test_a_device_sumcheck_proof_replays_under_a_hashlib_verifier():
    """stage 3 of the catalogue (three relations, degree 3) proved on the device under LegacyBlake2bTranscript; the verifier below knows only the compressed round
    polynomials and the claimed sum: it rebuilds each linear coefficient from s(0) + s(1) = claim, absorbs LabelWithCount("sumcheck_poly", d) + the d stored
    coefficients as 32 big-endian bytes each, squeezes the 125-bit challenge, and must arrive at the device's challenges and final claim"""
    ctx = ffi.Context(0)
    n_vars, label = 8, ffi.TRANSCRIPT_BLAKE2B | 901
    dev = DeviceWorkload(ctx, n_vars, seed=3)
    out = dev.prove(label=label)[3]
    idxs = dev.stages[3]
    claim = 0
    for i in idxs:
        claim = (claim + fr_int(dev.batch_coeffs[i]) * fr_int(dev.claims[i])) % R_MOD
    text = b"jolt-amd/%d" % ((label + 3) & ((1 << 62) - 1))
    state, n_rounds = hashlib.blake2b(text.ljust(32, b"\0"), digest_size=32).digest(), 0
    inv_r = pow(2**256, -1, R_MOD)
    for r in range(n_vars):
        coeffs = [fr_int(c) for c in out["polys"][r]]
        while len(coeffs) > 2 and coeffs[-1] == 0:
            coeffs.pop()
        stored = [coeffs[0]] + coeffs[2:]  # CompressedUniPoly: the linear term is not sent
        linear = (claim - 2 * stored[0] - sum(stored[1:])) % R_MOD
        assert linear == coeffs[1], r
        for payload in [b"sumcheck_poly".ljust(24, b"\0") + len(stored).to_bytes(8, "big")] + [c.to_bytes(32, "big") for c in stored]:
            state = hashlib.blake2b(state + bytes(28) + n_rounds.to_bytes(4, "big") + payload, digest_size=32).digest()
            n_rounds += 1
        state = hashlib.blake2b(state + bytes(28) + n_rounds.to_bytes(4, "big"), digest_size=32).digest()
        n_rounds += 1
        v = int.from_bytes(state[:16], "little")
        challenge = (((v & (2**64 - 1)) << 128) | (((v >> 64) & (2**61 - 1)) << 192)) * inv_r % R_MOD
        assert challenge == fr_int(out["challenges"][r]), r
        claim = sum(c * pow(challenge, k, R_MOD) for k, c in enumerate(coeffs)) % R_MOD
    assert claim == fr_int(out["final_claim"])
    dev.close()
    ctx.close()


@pytest.mark.parametrize("kind", KINDS)
def test_stage_operators_under_the_reference_transcripts(kind):
    ctx = ffi.Context(0)
    kw = dict(n_tables=12)
    dev = DeviceExtended(ctx, 9, seed=30, **kw)
    got = dev.prove(label=kind | 40)
    want = OracleExtended(9, seed=30, **kw).prove(label=kind | 40)
    address_domain = {"bytecode_read_raf", "ram_raf_evaluation", "ram_output_check", "hamming_weight", "booleanity_cycle"}
    for name in got:
        same(got[name], {k: v for k, v in want[name].items() if k != "claim" or name in address_domain}, name)
    dev.close()
    ctx.close()


@pytest.mark.parametrize("kind", KINDS)
def test_hyperkzg_opening_under_the_reference_transcripts(kind):
    ctx = ffi.Context(0)
    ell = 7
    n = 1 << ell
    host_srs = O.srs_setup_from_secret(rand_fr(1, 77)[0], n + 1)
    srs = ctx.srs_upload(host_srs)
    evals, point = rand_fr(n, 78), np.stack([rand_challenge(80 + k) for k in range(ell)])
    got = ctx.hyperkzg_open(srs, ctx.upload(evals), point, label=kind | 9)
    want = O.hyperkzg_open(host_srs, evals, point, label=kind | 9)
    assert np.array_equal(got["challenges"], want["challenges"])
    assert np.array_equal(got["v"], want["v"])
    for i in range(ell - 1):
        assert same_point(got["com"][i], want["com"][i])
    for t in range(3):
        assert same_point(got["w"][t], want["w"][t])
    if kind == ffi.TRANSCRIPT_BLAKE2B:  # the first challenge from hashlib: the ell - 1 level commitments in the compressed arkworks encoding, then Transcript::challenge
        state, n_rounds = hashlib.blake2b(b"jolt-amd/9".ljust(32, b"\0"), digest_size=32).digest(), 0
        for c in want["com"]:
            state = hashlib.blake2b(state + bytes(28) + n_rounds.to_bytes(4, "big") + O.g1_serialize_compressed(c), digest_size=32).digest()
            n_rounds += 1
        state = hashlib.blake2b(state + bytes(28) + n_rounds.to_bytes(4, "big"), digest_size=32).digest()
        v = int.from_bytes(state[:16], "little")
        r = (((v & (2**64 - 1)) << 128) | (((v >> 64) & (2**61 - 1)) << 192)) * pow(2**256, -1, R_MOD) % R_MOD
        assert r == fr_int(got["challenges"][0])
    ctx.close()
