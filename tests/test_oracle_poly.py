"""Pin the oracle's jolt-poly restatement with the identities the reference's own tests use
(/root/reference/crates/jolt-poly/src/dense.rs:578-1119, eq.rs:475-756, split_eq.rs:527-618, lt.rs tests,
eq_plus_one.rs tests) against an independent Python big-integer model."""
import random

import numpy as np

import oracle_lib as O

R = O.R_MOD


def rand_table(rng, n):
    return [rng.randrange(R) for _ in range(n)]


def eq_model(r, x_bits):
    acc = 1
    for ri, xi in zip(r, x_bits):
        acc = acc * (ri if xi else (1 - ri)) % R
    return acc


def bits_msb(i, n):
    return [(i >> (n - 1 - k)) & 1 for k in range(n)]


def test_bind_both_orders_vs_model():
    rng = random.Random(1)
    for n in (1, 2, 5, 8):
        t = rand_table(rng, 1 << n)
        r = rng.randrange(R)
        tm, rm = O.to_mont(t), O.to_mont([r])[0]
        half = len(t) // 2
        assert O.from_mont(O.bind_high_to_low(tm, rm)) == [(t[i] + r * (t[i + half] - t[i])) % R for i in range(half)]
        assert O.from_mont(O.bind_low_to_high(tm, rm)) == [(t[2 * i] + r * (t[2 * i + 1] - t[2 * i])) % R for i in range(half)]


def test_bind_all_vars_equals_evaluate():
    # dense.rs tests: binding every variable (HighToLow, point order) == evaluate(point);
    # LowToHigh binds consume the point back to front
    rng = random.Random(2)
    n = 6
    t = rand_table(rng, 1 << n)
    point = [rng.randrange(R) for _ in range(n)]
    tm, pm = O.to_mont(t), O.to_mont(point)
    want = sum(t[i] * eq_model(point, bits_msb(i, n)) for i in range(1 << n)) % R
    assert O.from_mont(O.poly_evaluate(tm, pm)) == [want]
    cur = tm
    for k in range(n):
        cur = O.bind_high_to_low(cur, pm[k])
    assert O.from_mont(cur) == [want]
    cur = tm
    for k in reversed(range(n)):
        cur = O.bind_low_to_high(cur, pm[k])
    assert O.from_mont(cur) == [want]


def test_bind_to_field_u64():
    rng = random.Random(3)
    t = [rng.randrange(2**64) for _ in range(16)]
    r = rng.randrange(R)
    got = O.from_mont(O.bind_to_field_u64(np.array(t, dtype=np.uint64), O.to_mont([r])[0]))
    assert got == [(t[i] + r * (t[i + 8] - t[i])) % R for i in range(8)]


def test_eq_tables():
    rng = random.Random(4)
    for n in (0, 1, 3, 7):
        r = [rng.randrange(R) for _ in range(n)]
        rm = O.to_mont(r) if n else np.zeros((0, 4), dtype=np.uint64)
        want = [eq_model(r, bits_msb(i, n)) for i in range(1 << n)]
        assert O.from_mont(O.eq_evals(rm)) == want
        assert O.from_mont(O.eq_evaluations(rm)) == want
        s = rng.randrange(R)
        assert O.from_mont(O.eq_evals(rm, O.to_mont([s])[0])) == [w * s % R for w in want]
        if n >= 3:
            full = O.eq_evals(rm)
            for block in (1, 2, 4):
                for start in range(0, 1 << n, block):
                    got = O.eq_evals_aligned_block(rm, start, block)
                    assert np.array_equal(got, full[start:start + block])
    x = [rng.randrange(R) for _ in range(5)]
    y = [rng.randrange(R) for _ in range(5)]
    want = 1
    for a, b in zip(x, y):
        want = want * (a * b + (1 - a) * (1 - b)) % R
    assert O.from_mont(O.eq_mle(O.to_mont(x), O.to_mont(y))) == [want]


def test_lt_and_eq_plus_one_boolean_truth_tables():
    # lt.rs boolean_correctness, eq_plus_one.rs tests
    for n in (1, 3, 4):
        for r_int in range(1 << n):
            rb = O.to_mont(bits_msb(r_int, n))
            assert O.from_mont(O.lt_evals(rb)) == [1 if x < r_int else 0 for x in range(1 << n)]
            eq, eq1 = O.eq_plus_one_evals(rb)
            assert O.from_mont(eq) == [1 if x == r_int else 0 for x in range(1 << n)]
            assert O.from_mont(eq1) == [1 if x == r_int + 1 else 0 for x in range(1 << n)]


def test_lt_random_point_matches_formula():
    # lt.rs:119-137 evaluate formula: sum_i (1-x_i) r_i eq(x[..i], r[..i])
    rng = random.Random(5)
    n = 5
    r = [rng.randrange(R) for _ in range(n)]
    table = O.from_mont(O.lt_evals(O.to_mont(r)))
    for j in range(1 << n):
        x = bits_msb(j, n)
        lt, pre = 0, 1
        for xi, ri in zip(x, r):
            lt = (lt + (1 - xi) * ri * pre) % R
            pre = pre * (xi * ri + (1 - xi) * (1 - ri)) % R
        assert table[j] == lt


def test_univariate_from_evals_and_evaluate():
    rng = random.Random(6)
    for n in (2, 3, 4, 6):
        coeffs = [rng.randrange(R) for _ in range(n)]
        evals = [sum(c * pow(x, k, R) for k, c in enumerate(coeffs)) % R for x in range(n)]
        got = O.univariate_from_evals(O.to_mont(evals))
        assert O.from_mont(got) == coeffs
        x = rng.randrange(R)
        assert O.from_mont(O.univariate_evaluate(got, O.to_mont([x])[0])) == [sum(c * pow(x, k, R) for k, c in enumerate(coeffs)) % R]


def test_split_eq_dims_cover_remaining_vars():
    # split_eq.rs:214-236,339-350: E_out x E_in always spans the n - bound - 1 variables not yet bound
    for n in range(1, 12):
        for bound in range(n):
            ob, ib = O.split_eq_current_dims(n, bound)
            assert ob + ib == n - bound - 1
