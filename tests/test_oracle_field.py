"""Pin the CPU oracle's Fr/Fq layer against the reference's golden vectors and a big-integer model.

Golden vectors: tests/golden/bn254_golden_bytes.json (extracted from
/root/reference/crates/jolt-field/tests/golden_bytes.rs:68-254).
Differential model: Python ints mod r, the same strategy as
/root/reference/crates/jolt-field/tests/bn254_differential.rs:76-272 (num-bigint there).
"""
import json
import os
import random

import numpy as np
import pytest

import oracle_lib as O
from util import rand_fr

R, Q = O.R_MOD, O.Q_MOD
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bn254_golden_bytes.json")))["tables"]


def test_golden_fr_bytes():
    # golden_bytes.rs:343-346 fr_bytes_match_fixtures: from_bytes_le_reduced(input).to_bytes_le() == expected
    for inp, exp in GOLD["FIX_BN254_FR"]:
        e = O.fr_from_bytes_le_reduced(bytes.fromhex(inp))
        assert O.fr_to_bytes_le(e).hex() == exp


def test_golden_fr_challenges():
    # golden_bytes.rs:353-356
    for inp, exp in GOLD["FIX_BN254_FR_CHALLENGE"]:
        assert O.fr_to_bytes_le(O.fr_from_challenge_bytes(bytes.fromhex(inp))).hex() == exp
    for inp, exp in GOLD["FIX_BN254_FR_SCALAR_CHALLENGE"]:
        assert O.fr_to_bytes_le(O.fr_from_scalar_challenge_bytes(bytes.fromhex(inp))).hex() == exp


def test_golden_fq_bytes():
    # golden_bytes.rs:348-351: the oracle's Fq Montgomery layer must reproduce the canonical encodings
    for inp, exp in GOLD["FIX_BN254_FQ"]:
        v = int.from_bytes(bytes.fromhex(inp), "little") % Q
        canon = np.array([O.int_to_limbs(v)], dtype=np.uint64)
        back = O.fq_to_canonical(O.fq_from_canonical(canon))
        assert O.limbs_to_int(back[0]).to_bytes(32, "little").hex() == exp


def test_golden_fq_challenges():
    # Fq arm (mod.rs:262): checked from_bigint of [0,0,low,high] -> ordinary integer value
    for inp, exp in GOLD["FIX_BN254_FQ_CHALLENGE"]:
        b = bytes.fromhex(inp)[:16].ljust(16, b"\0")
        v = int.from_bytes(b, "little")
        low, high = v & (2**64 - 1), (v >> 64) & ((2**64 - 1) >> 3)
        val = ((low << 128) | (high << 192)) % Q
        assert val.to_bytes(32, "little").hex() == exp
    for inp, exp in GOLD["FIX_BN254_FQ_SCALAR_CHALLENGE"]:
        v = int.from_bytes(bytes.fromhex(inp), "big") % Q
        assert v.to_bytes(32, "little").hex() == exp


def _rand_elems(rng, n, mod):
    edge = [0, 1, 2, mod - 1, mod - 2, (mod - 1) // 2, 2**64 - 1, 2**128 - 1, 2**253]
    vals = edge + [rng.randrange(mod) for _ in range(n - len(edge))]
    return vals


@pytest.mark.parametrize("mod,add,sub,mul,inv", [
    (R, "fr_add", "fr_sub", "fr_mul", "fr_inv"),
    (Q, "fq_add", "fq_sub", "fq_mul", "fq_inv"),
])
def test_differential_ring_ops(mod, add, sub, mul, inv):
    rng = random.Random(0xB254B254)
    a_i, b_i = _rand_elems(rng, 300, mod), list(reversed(_rand_elems(rng, 300, mod)))
    a, b = O.to_mont(a_i, mod), O.to_mont(b_i, mod)
    assert O.from_mont(getattr(O, add)(a, b), mod) == [(x + y) % mod for x, y in zip(a_i, b_i)]
    assert O.from_mont(getattr(O, sub)(a, b), mod) == [(x - y) % mod for x, y in zip(a_i, b_i)]
    assert O.from_mont(getattr(O, mul)(a, b), mod) == [(x * y) % mod for x, y in zip(a_i, b_i)]
    got = O.from_mont(getattr(O, inv)(a), mod)
    assert got == [pow(x, -1, mod) if x else 0 for x in a_i]
    # results stay canonical (< p) in Montgomery form
    for arr in (getattr(O, add)(a, b), getattr(O, mul)(a, b)):
        assert all(O.limbs_to_int(row) < mod for row in arr)


def test_from_int_and_small_scalar_mul():
    # bn254_differential.rs: from_u64/u128/i64/i128, mul_u64/mul_u128 vs the integer model;
    # mont.rs:703-717 kernel_matches_arkworks incl. the 2^14 table boundary
    rng = random.Random(7)
    u64s = [0, 1, 2, (1 << 14) - 1, 1 << 14, 2**63, 2**64 - 1] + [rng.randrange(2**64) for _ in range(100)]
    assert O.from_mont(O.fr_from_u64(u64s)) == [v % R for v in u64s]
    i64s = [0, -1, 1, -(2**63), 2**63 - 1] + [rng.randrange(-2**63, 2**63) for _ in range(100)]
    assert O.from_mont(O.fr_from_i64(i64s)) == [v % R for v in i64s]
    for v in [0, 1, 2**64, 2**128 - 1] + [rng.randrange(2**128) for _ in range(50)]:
        assert O.from_mont(O.fr_from_u128(v)) == [v % R]
        assert O.from_mont(O.fr_from_i128(-v)) == [(-v) % R]
    for _ in range(200):
        a = rng.randrange(R)
        am = O.to_mont([a])[0]
        b = rng.choice([0, 1, rng.randrange(2**64)])
        c = rng.choice([0, 1, rng.randrange(2**128)])
        assert O.from_mont(O.fr_mul_u64(am, b)) == [a * b % R]
        assert O.from_mont(O.fr_mul_u128(am, c)) == [a * c % R]
    assert O.from_mont(O.fr_mul_pow_2(O.to_mont([5])[0], 70)) == [5 * 2**70 % R]


def test_from_montgomery_reduce_and_wide_accumulator():
    # mont.rs:719-735 montgomery_reduce_roundtrip and 630-660 accumulator tests, against the integer model
    rng = random.Random(8)
    rinv = pow(O.MONT_R, -1, R)
    for L in (8, 9, 10):
        for _ in range(50):
            v = rng.randrange(2 ** (64 * L)) if L > 8 else rng.randrange(R * 2**256)
            limbs = np.array(O.int_to_limbs(v, L), dtype=np.uint64)
            got = O.limbs_to_int(O.fr_from_montgomery_reduce(limbs))
            assert got == v * rinv % R
    a_i = [rng.randrange(R) for _ in range(1000)]
    b_i = [rng.randrange(R) for _ in range(1000)]
    adds = [rng.randrange(R) for _ in range(37)]
    got = O.from_mont(O.wide_accumulate(O.to_mont(a_i), O.to_mont(b_i), O.to_mont(adds)))
    assert got == [(sum(x * y for x, y in zip(a_i, b_i)) + sum(adds)) % R]


def test_spread_recipe_matches_model():
    # mont.rs:612-616 `spread(seed)`: a*b^2 + a with odd 64-bit a, b
    for seed in range(50):
        a = (seed * 0x9E3779B97F4A7C15) % 2**64 | 1
        b = (seed * 0xBF58476D1CE4E5B9) % 2**64 | 1
        am, bm = O.fr_from_u64([a]), O.fr_from_u64([b])
        got = O.fr_add(O.fr_mul(O.fr_mul(am, bm), bm), am)
        assert O.from_mont(got) == [(a * b * b + a) % R]


def test_small_scalar_accumulator_restatement_equals_plain_sums():
    """FrSmallScalarAccumulator (mont.rs:343-427) as restated in oracle/fr.c: positive / negative five-limb sums with ONE Barrett
    reduction give the same value as the sum of reduced products -- over the scalar corners 0, +-1, i64::MIN / MAX.  The five
    limbs wrap at 2^320 (add_assign_trunc), i.e. the sums of |scalars| must stay below ~2^66: full-range scalars for a handful of
    terms, 48-bit scalars for long sums (the witness columns it is used on are flags and register-sized values)."""
    rng = np.random.default_rng(9)
    for n in (1, 2, 5, 300, 5000):
        values = rand_fr(n, 40 + n)
        if n <= 5:
            scalars = rng.integers(-2**63, 2**63 - 1, size=n, dtype=np.int64)
            scalars[: min(n, 5)] = [0, 1, -1, -2**63, 2**63 - 1][: min(n, 5)]
        else:
            scalars = rng.integers(-2**48, 2**48, size=n, dtype=np.int64)
        want = np.zeros((1, 4), dtype=np.uint64)
        for k in range(n):
            want = O.fr_add(want, O.fr_mul(values[k].reshape(1, 4), O.fr_from_i64(scalars[k: k + 1])))
        assert np.array_equal(O.small_scalar_accumulate(values, scalars), want[0]), n
