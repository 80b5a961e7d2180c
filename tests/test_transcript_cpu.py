"""The reference's Fiat-Shamir transcripts on both sides of the parity tests (oracle/mock_transcript.h, jolt_amd/csrc/host_mirror.hip).

Pins, in order of strength:
  * the reference's OWN known-answer vectors (crates/jolt-transcript/tests/{keccak,blake2b}_tests.rs, extracted by tests/golden/extract_transcript_kats.py): PROTOCOL_ID,
    the session framing, the append framing, the sponge constructions and the 125-bit challenge decoder, for the oracle AND the product;
  * the hash primitives against their specifications: BLAKE2b against RFC 7693 appendix A and hashlib at every block-boundary length, Keccak-f[1600] through SHA3-256
    built on it against hashlib;
  * LegacyBlake2bTranscript (the transcript of the reference's benchmark profile, crates/jolt-prover/src/profile.rs:69) against a third, independent restatement of
    crates/jolt-transcript/src/digest.rs written here over hashlib.blake2b;
  * oracle == product for every engine over random append / challenge sequences (two separately written implementations)."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import ffi
from util import rand_fr

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_transcript_kats.json")))["cases"]
ENGINE = {"keccak_sponge": 2, "blake2b512_sponge": 3}
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def fr_int(a):  # canonical integer of a Montgomery element
    return int.from_bytes(O.fr_to_bytes_le(a), "little")


@pytest.mark.parametrize("case", KATS, ids=[c["test"] for c in KATS])
def test_reference_known_answer_vectors(case):
    """Transcript::new(b"Jolt"); append_bytes(12345u64 BE); challenge() -- the bytes the reference's tests hold"""
    want = int.from_bytes(bytes(case["challenge_le_bytes"]), "little") % R_MOD
    kind = ENGINE[case["engine"]]
    for make in (lambda: O.MockTranscript(case["label"].encode(), kind=kind), lambda: ffi.HostTranscript(case["label"].encode(), kind=kind)):
        t = make()
        t.append_bytes(case["append_u64_be"].to_bytes(8, "big"))
        assert fr_int(t.challenge()) == want


def test_blake2b_against_rfc7693_and_hashlib():
    abc = bytes.fromhex("ba80a53f981c4d0d6a2797b69f12f6e94c212f14685ac4b74b12bb6fdbffa2d17d87c5392aab792dc252d5de4533cc9518d38aa8dbf1925ab92386edd4009923")  # RFC 7693 appendix A
    for digest in (O.blake2b_digest, ffi.host_blake2b):
        assert digest(b"abc", 64) == abc
        for n in (0, 1, 31, 32, 64, 96, 127, 128, 129, 255, 256, 257, 1000):
            data = bytes((i * 7 + 3) & 255 for i in range(n))
            for outlen in (32, 64):
                assert digest(data, outlen) == hashlib.blake2b(data, digest_size=outlen).digest(), (n, outlen)


def test_keccak_f1600_through_sha3_256():
    def sha3_256(data, permute):
        rate, state = 136, bytearray(200)
        padded = bytearray(data) + b"\x06"
        padded += bytes(-len(padded) % rate)
        padded[-1] |= 0x80
        for off in range(0, len(padded), rate):
            for i in range(rate):
                state[i] ^= padded[off + i]
            state = bytearray(permute(bytes(state)))
        return bytes(state[:32])

    def orc_permute(st):
        import ctypes as C
        buf = (C.c_uint8 * 200).from_buffer_copy(st)
        O.lib().orc_keccak_f1600_permute(buf)
        return bytes(buf)

    for permute in (orc_permute, ffi.host_keccak_f1600):
        for n in (0, 5, 135, 136, 137, 300):
            data = bytes((i * 11 + 1) & 255 for i in range(n))
            assert sha3_256(data, permute) == hashlib.sha3_256(data).digest(), n


class DigestModel:
    """crates/jolt-transcript/src/digest.rs:84-189 over hashlib: a third restatement, independent of both C implementations"""

    def __init__(self, label):
        self.state, self.n_rounds = hashlib.blake2b(label.ljust(32, b"\0"), digest_size=32).digest(), 0

    def _hasher(self):
        return hashlib.blake2b(self.state + bytes(28) + self.n_rounds.to_bytes(4, "big"), digest_size=32)

    def append_bytes(self, b):
        h = self._hasher()
        h.update(b)
        self.state, self.n_rounds = h.digest(), self.n_rounds + 1

    def challenge_bytes16(self):
        self.state, self.n_rounds = self._hasher().digest(), self.n_rounds + 1
        return self.state[:16]


def challenge_from_bytes(b16):  # Fr::from_challenge_bytes (crates/jolt-field/src/bn254/mod.rs:171-184): 125 bits in the two HIGH limbs of the Montgomery representation
    v = int.from_bytes(b16, "little")
    lo, hi = v & (2**64 - 1), (v >> 64) & (2**61 - 1)
    return ((lo << 128) | (hi << 192)) * pow(2**256, -1, R_MOD) % R_MOD


def test_legacy_blake2b_transcript_against_the_model():
    rng = np.random.default_rng(7)
    vals = rand_fr(8, 11)
    for label in (b"Jolt", b"", b"prove-driver-twin", b"x" * 32):
        model, orc, mine = DigestModel(label), O.MockTranscript(label, kind=1), ffi.HostTranscript(label, kind=1)
        assert orc.state() == mine.state() == model.state
        for step in range(40):
            what = int(rng.integers(0, 5))
            if what == 0:
                data = bytes(rng.integers(0, 256, int(rng.integers(0, 300)), dtype=np.uint8))
                for t in (model, orc, mine):
                    t.append_bytes(data)
            elif what == 1:  # a field element: 32 big-endian bytes (legacy.rs:116-123)
                v = vals[step % 8]
                model.append_bytes(fr_int(v).to_bytes(32, "big"))
                orc.append_fr(v)
                mine.append(v)
            elif what == 2:  # LabelWithCount (legacy.rs:180-195)
                model.append_bytes(b"sumcheck_poly".ljust(24, b"\0") + (step).to_bytes(8, "big"))
                orc.append_label_with_count(b"sumcheck_poly", step)
                mine.append_label(b"sumcheck_poly", step)
            elif what == 3:  # a compressed labelled round polynomial (round_proof.rs:129-143)
                coeffs = vals[:4]
                model.append_bytes(b"sumcheck_poly".ljust(24, b"\0") + (3).to_bytes(8, "big"))
                for k in (0, 2, 3):
                    model.append_bytes(fr_int(coeffs[k]).to_bytes(32, "big"))
                orc.append_round_poly(coeffs)
                mine.append_round_poly(coeffs)
            else:
                want = challenge_from_bytes(model.challenge_bytes16())
                assert fr_int(orc.challenge()) == want and fr_int(mine.challenge()) == want
            assert orc.state() == mine.state() == model.state
        model.append_bytes(b"opening_claim".ljust(32, b"\0"))  # Label (legacy.rs:146-163)
        orc.append_label(b"opening_claim")
        mine.append_label(b"opening_claim")
        model.append_bytes(bytes(24) + (77).to_bytes(8, "big"))  # U64Word (legacy.rs:197-209)
        orc.append_u64_word(77)
        mine.append_u64_word(77)
        assert orc.state() == mine.state() == model.state


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_oracle_and_product_transcripts_agree(kind):
    """integer labels (the engine in the two top bits) and byte labels; appends across the sponge's rate (136 bytes) and the digest's block (128), challenges of both
    shapes back to back (the Blake2b sponge serves the second from the unused tail of its block), state()"""
    rng = np.random.default_rng(100 + kind)
    vals = rand_fr(6, 5 + kind)
    pairs = [(O.MockTranscript((kind << 62) | 4711), ffi.HostTranscript((kind << 62) | 4711))]
    if kind:
        pairs.append((O.MockTranscript(b"Jolt", kind=kind), ffi.HostTranscript(b"Jolt", kind=kind)))
    for orc, mine in pairs:
        for step in range(60):
            what = int(rng.integers(0, 4))
            if what == 0:
                data = bytes(rng.integers(0, 256, int(rng.integers(0, 400)), dtype=np.uint8))
                orc.append_bytes(data)
                mine.append_bytes(data)
            elif what == 1:
                orc.append_fr(vals[step % 6])
                mine.append(vals[step % 6])
            elif what == 2:
                orc.append_round_poly(vals[:5], b"uniskip_poly")
                mine.append_round_poly(vals[:5], b"uniskip_poly")
            else:
                for full in (False, True, False):
                    a = orc.challenge_scalar() if full else orc.challenge()
                    assert np.array_equal(a, mine.challenge(full_width=full))
            assert orc.state() == mine.state()


def test_transcript_entry_points_refuse_bad_arguments():
    with pytest.raises(Exception):
        ffi.HostTranscript(b"x" * 33, kind=1)
    with pytest.raises(Exception):
        ffi.HostTranscript(b"Jolt", kind=0)
    with pytest.raises(ValueError):
        O.MockTranscript(b"x" * 33, kind=1)
    t = ffi.HostTranscript(b"Jolt", kind=2)
    with pytest.raises(Exception):
        t.append_label(b"y" * 25, 1)
    with pytest.raises(Exception):
        t.append(np.full((1, 4), 2**64 - 1, dtype=np.uint64))  # not canonical


@pytest.mark.parametrize("kind", [1, 2, 3])
@pytest.mark.parametrize("side", ["oracle", "product"])
def test_the_references_transcript_property_tests(kind, side):
    """crates/jolt-transcript/tests/common/mod.rs:24-175 (`transcript_tests!`, instantiated there for every engine) re-run on both implementations: determinism, domain
    separation by label, unique successive challenges, appends change the state, order and data sensitivity, an EMPTY append changes the state (and does so
    reproducibly), large data, a 32-byte label is accepted and a 33-byte one refused, challenge vectors are distinct"""
    if side == "oracle":
        new = lambda label: O.MockTranscript(label, kind=kind)
        app = lambda t, b: t.append_bytes(b)
    else:
        new = lambda label: ffi.HostTranscript(label, kind=kind)
        app = lambda t, b: t.append_bytes(b)
    ch = lambda t: fr_int(t.challenge())

    def after(label, *chunks):
        t = new(label)
        for c in chunks:
            app(t, c)
        return ch(t)

    assert after(b"test", b"hello") == after(b"test", b"hello")  # test_determinism
    assert after(b"label_a") != after(b"label_b")  # test_domain_separation
    t = new(b"test")
    seen = [ch(t) for _ in range(100)]
    assert len(set(seen)) == 100  # test_challenge_uniqueness
    assert after(b"test") != after(b"test", b"data")  # test_append_changes_state
    assert after(b"test", b"a", b"b") != after(b"test", b"b", b"a")  # test_order_matters
    assert after(b"test", b"data1") != after(b"test", b"data2")  # test_data_sensitivity
    assert after(b"test", b"") != after(b"test")  # test_empty_bytes: the framing absorbs a length / a round word even for no payload
    assert after(b"test", b"") == after(b"test", b"")
    assert after(b"test", bytes(range(256)) * 40) == after(b"test", bytes(range(256)) * 40)  # test_large_data
    assert after(b"ab", b"c") != after(b"a", b"bc")
    assert after(b"x" * 32) == after(b"x" * 32)  # test_max_valid_label
    with pytest.raises(Exception):
        new(b"x" * 33)  # test_label_too_long (the reference panics; here an error status)
    t = new(b"test")
    assert len({ch(t) for _ in range(5)}) == 5  # test_challenge_vector
