"""Pin the oracle's BN254 G1 / MSM / HyperKZG restatement.  The reference holds no golden points
(SURVEY.md 8c), so the pins are: an independent Python affine big-int model of y^2 = x^3 + 3, the group laws and
msm == sum s_i P_i (/root/reference/crates/jolt-crypto/tests/group_laws.rs:69-78,135-146), the KZG division
identity (/root/reference/crates/jolt-hyperkzg/src/kzg.rs:229-264) and commit(p) == p(beta) G."""
import random

import numpy as np

import oracle_lib as O
from util import rand_fr

R, Q = O.R_MOD, O.Q_MOD


# ---- independent affine model (None = infinity)
def aff_add(P, S):
    if P is None: return S
    if S is None: return P
    (x1, y1), (x2, y2) = P, S
    if x1 == x2:
        if (y1 + y2) % Q == 0: return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, Q) % Q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
    x3 = (lam * lam - x1 - x2) % Q
    return (x3, (lam * (x1 - x3) - y1) % Q)


def aff_mul(k, P):
    acc = None
    while k:
        if k & 1: acc = aff_add(acc, P)
        P = aff_add(P, P)
        k >>= 1
    return acc


def to_model(p):
    if O.g1_is_identity(p): return None
    a = O.g1_to_affine(p)[0]
    xs = O.from_mont(a[:4], Q)[0]
    ys = O.from_mont(a[4:], Q)[0]
    return (xs, ys)


G = (1, 2)


def test_generator_and_small_multiples_vs_affine_model():
    g = O.g1_generator()
    assert O.g1_on_curve(g) and to_model(g) == G
    acc = O.g1_identity()
    for k in range(1, 20):
        acc = O.g1_add(acc, g)
        assert O.g1_on_curve(acc)
        assert to_model(acc) == aff_mul(k, G)
    rng = random.Random(21)
    for _ in range(5):
        k = rng.randrange(R)
        assert to_model(O.g1_scalar_mul(g, O.to_mont([k])[0])) == aff_mul(k, G)


def test_public_known_answers():
    """Known-answer pins that do not come from this repository: 2*(1,2) on alt_bn128 as it appears in the EIP-196 precompile test
    vectors, and r*G = infinity for the BN254 group order (the modulus pinned by crates/jolt-field/tests/bn254_differential.rs:21-36)."""
    g = O.g1_generator()
    two_g = (0x030644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3,
             0x15ed738c0e0a7c92e7845f96b2ae9c0a68a6a449e3538fc7ff3ebf7a5a18a2c4)
    assert to_model(O.g1_double(g)) == two_g
    assert to_model(O.g1_add(g, g)) == two_g
    assert to_model(O.g1_scalar_mul(g, O.to_mont([2])[0])) == two_g
    assert O.g1_is_identity(O.g1_scalar_mul(g, O.to_mont([0])[0]))
    # (r - 1) G = -G, hence r G = infinity
    minus_g = O.g1_scalar_mul(g, O.to_mont([R - 1])[0])
    assert to_model(minus_g) == (1, Q - 2)
    assert O.g1_is_identity(O.g1_add(minus_g, g))
    # compressed form of G: x = 1, y = 2 is the smaller root -> no sign flag
    assert O.g1_serialize_compressed(g) == (1).to_bytes(32, "little")


def test_group_laws_and_corner_cases():
    g = O.g1_generator()
    ident = O.g1_identity()
    p = O.g1_scalar_mul(g, O.to_mont([12345])[0])
    q = O.g1_scalar_mul(g, O.to_mont([67890])[0])
    assert O.g1_eq(O.g1_add(p, q), O.g1_add(q, p))
    assert O.g1_eq(O.g1_add(p, ident), p) and O.g1_eq(O.g1_add(ident, p), p)
    assert O.g1_is_identity(O.g1_add(p, O.g1_neg(p)))          # P + (-P)
    assert O.g1_eq(O.g1_add(p, p), O.g1_double(p))              # P + P through the add path
    assert O.g1_is_identity(O.g1_double(ident))
    # r*G = identity
    assert O.g1_is_identity(O.g1_scalar_mul(g, O.to_mont([0])[0]))
    # different projective representatives compare equal as points
    p2 = O.g1_add(O.g1_add(p, q), O.g1_neg(q))
    assert O.g1_eq(p, p2) and not np.array_equal(p, p2)


def test_compressed_serialization_flags():
    g = O.g1_generator()
    b = O.g1_serialize_compressed(g)
    assert b[:31] == (1).to_bytes(31, "little") and b[31] & 0xC0 == 0   # y = 2 <= q-2 -> positive
    nb = O.g1_serialize_compressed(O.g1_neg(g))
    assert nb[31] & 0x80 and nb[:31] == b[:31]
    ib = O.g1_serialize_compressed(O.g1_identity())
    assert ib[31] == 0x40 and ib[:31] == bytes(31)


def test_msm_equals_sum_of_scalar_muls():
    rng = random.Random(22)
    g = O.g1_generator()
    for n in (0, 1, 2, 33, 70):
        ks = [rng.randrange(1, R) for _ in range(n)]
        bases = np.stack([O.g1_scalar_mul(g, O.to_mont([k])[0]) for k in ks]) if n else O.g1_array(0)
        svals = [rng.choice([0, 1, R - 1, rng.randrange(2**64), rng.randrange(R)]) for _ in range(n)]
        scalars = O.to_mont(svals) if n else O.fr_array(0)
        got = O.g1_msm_pippenger(bases, scalars)
        assert O.g1_eq(got, O.g1_msm_naive(bases, scalars))
        want = aff_mul(sum(k * s for k, s in zip(ks, svals)) % R, G)
        assert to_model(got) == want
    # repeated bases exercise the P+P / P+(-P) bucket corner cases
    b = O.g1_scalar_mul(g, O.to_mont([5])[0])
    bases = np.stack([b, b, O.g1_neg(b), b, O.g1_identity()])
    scalars = O.to_mont([7, 7, 7, 3, 11])
    assert to_model(O.g1_msm_pippenger(bases, scalars)) == aff_mul(5 * 10, G)


def test_kzg_commit_and_division_identity():
    rng = random.Random(23)
    beta = rng.randrange(R)
    n = 16
    srs = O.srs_setup_from_secret(O.to_mont([beta])[0], n + 1)
    assert to_model(srs[3]) == aff_mul(pow(beta, 3, R), G)
    coeffs = [rng.randrange(R) for _ in range(n)]
    cm = O.to_mont(coeffs)
    com = O.kzg_commit(cm, srs)
    pbeta = sum(c * pow(beta, i, R) for i, c in enumerate(coeffs)) % R
    assert to_model(com) == aff_mul(pbeta, G)
    # kzg.rs:229-264: f(x) = h(x) (x-u) + f(u)
    u = rng.randrange(R)
    h = O.from_mont(O.kzg_witness_polynomial(cm, O.to_mont([u])[0]))
    fu = O.from_mont(O.kzg_eval_univariate(cm, O.to_mont([u])[0]))[0]
    assert fu == sum(c * pow(u, i, R) for i, c in enumerate(coeffs)) % R
    x = rng.randrange(R)
    hx = sum(c * pow(x, i, R) for i, c in enumerate(h)) % R
    fx = sum(c * pow(x, i, R) for i, c in enumerate(coeffs)) % R
    assert fx == (hx * (x - u) + fu) % R
    try:
        O.kzg_commit(O.to_mont([1] * (n + 2)), srs)
        assert False
    except ValueError:
        pass


def test_hyperkzg_open_fold_consistency():
    # scheme.rs:226-240: 2 r y_next = r (1-x)(y_pos+y_neg) + x (y_pos-y_neg) on every level, last y_next = eval
    rng = random.Random(24)
    ell = 4
    n = 1 << ell
    beta = rng.randrange(R)
    srs = O.srs_setup_from_secret(O.to_mont([beta])[0], n + 1)
    evals = [rng.randrange(R) for _ in range(n)]
    point = [rng.randrange(R) for _ in range(ell)]
    em, pm = O.to_mont(evals), O.to_mont(point)
    polys = O.hyperkzg_fold_polynomials(em, pm)
    assert [len(p) for p in polys] == [16, 8, 4, 2]
    claimed = O.from_mont(O.poly_evaluate(em, pm))[0]
    out = O.hyperkzg_open(srs, em, pm, label=5)
    r = O.from_mont(out["challenges"][0])[0]
    v = [O.from_mont(out["v"][t]) for t in range(3)]
    y_sq = v[2] + [claimed]
    for lvl in range(ell):
        x = point[ell - 1 - lvl]
        lhs = 2 * r * y_sq[lvl + 1] % R
        rhs = (r * (1 - x) * (v[0][lvl] + v[1][lvl]) + x * (v[0][lvl] - v[1][lvl])) % R
        assert lhs == rhs, lvl
    # commitments are commitments of the folds; witnesses satisfy w_t * (beta - u_t) + B(u_t) G = B(beta) G
    for i in range(1, ell):
        pb = sum(c * pow(beta, k, R) for k, c in enumerate(O.from_mont(polys[i]))) % R
        assert to_model(out["com"][i - 1]) == aff_mul(pb, G)
    q = O.from_mont(out["challenges"][1])[0]
    Bc = [0] * n
    for j, p in enumerate(polys):
        for k, c in enumerate(O.from_mont(p)):
            Bc[k] = (Bc[k] + pow(q, j, R) * c) % R
    Bbeta = sum(c * pow(beta, k, R) for k, c in enumerate(Bc)) % R
    us = [r, (-r) % R, r * r % R]
    for t in range(3):
        Bu = sum(c * pow(us[t], k, R) for k, c in enumerate(Bc)) % R
        wt = (Bbeta - Bu) * pow(beta - us[t], -1, R) % R
        assert to_model(out["w"][t]) == aff_mul(wt, G)


def test_baseline_parallel_msm_and_grid_pieces_equal_the_serial_restatements():
    """bench.py's cpu_baseline leg runs the oracle's MSM / HyperKZG on all host cores (OpenMP tasks over windows x point chunks):
    same points as the serial restatements, the one-hot grid sum equals kzg_commit of the embedded 0/1 coefficient vector, and the
    joint polynomial equals the RLC of the embedded polynomials."""
    n = 600
    beta = rand_fr(1, 900)[0]
    srs = O.srs_setup_from_secret(beta, n)
    bases = O.baseline_prepare_bases(srs)
    # signed-digit corner scalars: 0, 1, r - 1, the two halves of r, window boundaries of c = 8 (n = 600), all-ones windows
    corner = O.to_mont([0, 1, R - 1, (R - 1) // 2, (R + 1) // 2, 127, 128, 129, 255, 256, (1 << 64) - 1, 1 << 64, (1 << 128) - 1, (1 << 253) + 5, R - 2])
    mixed = np.concatenate([corner, rand_fr(n - corner.shape[0], 909)])
    for scalars in (rand_fr(n, 901), O.fr_from_u64(np.arange(n, dtype=np.uint64) * 7 % 5), mixed,
                    O.fr_from_u64(np.random.default_rng(910).integers(0, 2**64, size=n, dtype=np.uint64))):
        assert O.g1_eq(O.baseline_msm(bases, scalars), O.g1_msm_pippenger(srs, scalars))
    assert O.g1_eq(O.baseline_msm(bases[10:], rand_fr(50, 902)), O.g1_msm_pippenger(srs[10:60], rand_fr(50, 902)))
    # several MSMs of different lengths as one task pool (the level commitments of an opening); a repeated base makes buckets double
    many = [rand_fr(600, 911), rand_fr(300, 912), O.fr_from_i64(np.arange(-75, 75, dtype=np.int64)), rand_fr(1, 913), mixed[:40]]
    got_many = O.baseline_msm_many(bases, many)
    for g, sc in zip(got_many, many):
        assert O.g1_eq(g, O.g1_msm_pippenger(srs[: sc.shape[0]], sc))
    dup = np.concatenate([srs[:1]] * 64)  # 64 copies of one base, equal scalars: every addition into the bucket is a doubling / identity case
    dbases = O.baseline_prepare_bases(dup)
    same = np.repeat(rand_fr(1, 914), 64, axis=0)
    assert O.g1_eq(O.baseline_msm(dbases, same), O.g1_msm_naive(dup, same))
    bases = O.baseline_prepare_bases(srs)
    ell = 5
    evals, point = rand_fr(1 << ell, 903), rand_fr(ell, 904)
    want = O.hyperkzg_open(srs, evals, point, label=3)
    O.baseline_use_parallel_msm(True)
    try:
        got = O.hyperkzg_open(bases, evals, point, label=3)
    finally:
        O.baseline_use_parallel_msm(False)
    assert np.array_equal(got["v"], want["v"]) and np.array_equal(got["challenges"], want["challenges"])
    for a, b in zip(list(got["com"]) + list(got["w"]), list(want["com"]) + list(want["w"])):
        assert O.g1_eq(a, b)
    T, K = 32, 16
    rng = np.random.default_rng(905)
    idx = rng.integers(0, K, size=(3, T), dtype=np.uint8)
    idx[1, rng.random(T) < 0.5] = 0xFF
    one = O.to_mont([1])[0]
    s, dense, ds = rand_fr(3, 906), [rand_fr(T, 907)], rand_fr(1, 908)
    want_j = np.zeros((K * T, 4), dtype=np.uint64)
    kt_bases = O.baseline_prepare_bases(srs[: K * T])  # the grid's bases exactly: the one-hot sums address them as k * T + j
    sums = O.baseline_grid_onehot_sums(kt_bases, idx)
    for p in range(3):
        emb = np.zeros((K * T, 4), dtype=np.uint64)
        hot = idx[p] != 0xFF
        emb[idx[p][hot].astype(np.int64) * T + np.nonzero(hot)[0]] = one
        assert O.g1_eq(sums[p], O.kzg_commit(emb, srs[: K * T]))
        want_j = O.fr_add(want_j, O.fr_mul(emb, np.repeat(s[p].reshape(1, 4), K * T, axis=0)))
    want_j[:T] = O.fr_add(want_j[:T], O.fr_mul(dense[0], np.repeat(ds[0].reshape(1, 4), T, axis=0)))
    assert np.array_equal(O.baseline_grid_joint(idx, K, s, dense, ds), want_j)
