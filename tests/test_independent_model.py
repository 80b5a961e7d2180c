"""The oracle (oracle/*.c) against tests/independent_model.py -- a second model written from the curve definition and the reference's .rs files
with different machinery (complete projective addition, LSB-first scalar multiplication, plain integers) -- on >= 100 deterministic cases per
layer.  The reference holds no vectors above the field layer; this removes the single-author risk from the oracle's G1 / MSM /
polynomial / KZG restatements, it does not replace the reference binary."""
import numpy as np
import pytest

import independent_model as M
import oracle_lib as O


def fr(v):
    return O.to_mont([int(x) % M.R for x in (v if isinstance(v, (list, tuple)) else [v])])


def fr_int(a):
    return [int(x) for x in O.from_mont(np.asarray(a).reshape(-1, 4))]


def to_oracle_point(p):
    """projective (X : Y : Z) -> the oracle's Jacobian Montgomery limbs (x, y, 1) or its identity"""
    a = M.affine(p)
    if a is None:
        return O.g1_identity()
    out = np.zeros(12, dtype=np.uint64)
    out[0:4], out[4:8], out[8:12] = O.to_mont([a[0]], M.Q)[0], O.to_mont([a[1]], M.Q)[0], O.to_mont([1], M.Q)[0]
    return out


def same(oracle_point, p):
    return O.g1_eq(oracle_point, to_oracle_point(p))


def lcg(seed):
    x = seed
    while True:
        x = (x * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        yield x


def wide(gen):  # a full-width scalar from four 64-bit draws
    return (next(gen) | next(gen) << 64 | next(gen) << 128 | next(gen) << 192) % M.R


def test_group_law_on_120_cases():
    gen = lcg(1)
    g = O.g1_generator()
    assert M.on_curve(M.G) and M.affine(M.pmul(M.G, M.R - 1)) == (1, M.Q - 2) and M.affine(M.pmul(M.G, M.R)) is None
    for case in range(120):
        a, b = wide(gen), wide(gen) if case % 7 else (M.R - 1, 1, 0, 2)[case % 4]
        pa, pb = M.pmul(M.G, a), M.pmul(M.G, b)
        oa, ob = O.g1_scalar_mul(g, fr(a)[0]), O.g1_scalar_mul(g, fr(b)[0])
        assert same(oa, pa) and same(ob, pb) and M.on_curve(pa)
        assert same(O.g1_add(oa, ob), M.padd(pa, pb))
        assert same(O.g1_double(oa), M.padd(pa, pa))
        assert same(O.g1_add(oa, O.g1_neg(oa)), M.INF)
        assert same(O.g1_scalar_mul(oa, fr(b)[0]), M.pmul(M.G, a * b % M.R))


def test_msm_on_100_cases():
    gen = lcg(2)
    powers_m = M.srs(wide(gen), 24)
    powers_o = np.stack([to_oracle_point(p) for p in powers_m])
    corner = [0, 1, M.R - 1, (M.R - 1) // 2, (M.R + 1) // 2, (1 << 64) - 1, 1 << 64, 15, 16, 17]
    for case in range(100):
        n = 1 + next(gen) % 24
        scalars = [corner[next(gen) % len(corner)] if (case + i) % 5 == 0 else (wide(gen) if case % 3 else next(gen)) for i in range(n)]
        want = M.msm_naive(powers_m[:n], scalars)
        assert M.affine(want) == M.affine(M.msm_pippenger(powers_m[:n], scalars))
        sc = fr(scalars)
        assert same(O.g1_msm_pippenger(powers_o[:n], sc), want), case
        assert same(O.g1_msm_naive(powers_o[:n], sc), want), case


def test_polynomial_layer_on_100_cases():
    for case in range(100):
        n = 1 + case % 6
        table = M.dense_member_table(n, 1000 + case)  # DenseMember::with_sum's table recipe
        point = M.synthetic_point(n, 77 + case)       # parity.rs synthetic_point
        t, p = fr(table), fr(point)
        assert fr_int(O.bind_low_to_high(t, p[0])) == M.bind_low_to_high(table, point[0])
        assert fr_int(O.bind_high_to_low(t, p[0])) == M.bind_high_to_low(table, point[0])
        assert fr_int(O.eq_evals(p)) == M.eq_table(point)
        assert fr_int(O.poly_evaluate(t, p)) == [M.evaluate(table, point)]
        # binding all variables low-to-high evaluates at the reversed point
        cur = table
        for x in reversed(point):
            cur = M.bind_low_to_high(cur, x)
        assert cur == [M.evaluate(table, point)]


def test_kzg_pieces_on_100_cases():
    gen = lcg(3)
    beta = wide(gen)
    powers_m = M.srs(beta, 32)
    powers_o = np.stack([to_oracle_point(p) for p in powers_m])
    assert all(same(a, b) for a, b in zip(O.srs_setup_from_secret(fr(beta)[0], 8), powers_m[:8]))
    for case in range(100):
        ell = 1 + case % 5
        n = 1 << ell
        coeffs = [wide(gen) if case % 2 else next(gen) for _ in range(n)]
        u = wide(gen)
        c = fr(coeffs)
        assert fr_int(O.kzg_eval_univariate(c, fr(u)[0])) == [M.eval_univariate(coeffs, u)]
        assert fr_int(O.kzg_witness_polynomial(c, fr(u)[0])) == M.witness_polynomial(coeffs, u)
        if case % 4 == 0:  # commitments are the slow part of the model: every fourth case
            want = M.kzg_commit(coeffs, powers_m)
            assert same(O.kzg_commit(c, powers_o), want)
            assert M.affine(want) == M.affine(M.pmul(M.G, M.eval_univariate(coeffs, beta)))  # commit(p) = p(beta) G
        point = M.synthetic_point(ell, case)
        got = O.hyperkzg_fold_polynomials(c, fr(point))
        for a, b in zip(got, M.fold_polynomials(coeffs, point)):
            assert fr_int(a) == b


@pytest.mark.parametrize("ell", [1, 2, 3, 4, 5])
def test_full_openings_given_the_oracle_challenges(ell):
    """the whole opening, four seeds per size: level commitments, all 3 * ell evaluations and the three witness commitments from the model
    (challenges r, q taken from the oracle's transcript, which the model does not restate)"""
    gen = lcg(10 + ell)
    n = 1 << ell
    for seed in range(4):
        beta = wide(gen)
        powers_m = M.srs(beta, n)
        powers_o = np.stack([to_oracle_point(p) for p in powers_m])
        evals = M.dense_member_table(ell, 5000 + seed) if seed % 2 else [wide(gen) for _ in range(n)]
        point = [wide(gen) for _ in range(ell)]
        out = O.hyperkzg_open(powers_o, fr(evals), fr(point), label=seed)
        r, q = fr_int(out["challenges"][0])[0], fr_int(out["challenges"][1])[0]
        coms, v, ws = M.open_given_challenges(powers_m, evals, point, r, q)
        assert len(coms) == ell - 1 and all(same(a, b) for a, b in zip(out["com"], coms))
        for t in range(3):
            assert fr_int(out["v"][t]) == v[t]
            assert same(out["w"][t], ws[t])
