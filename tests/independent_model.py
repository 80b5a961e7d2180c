"""A SECOND model of the layers above the field, independent of oracle/*.c and of the affine model inside tests/test_oracle_g1.py
(TEST INFRASTRUCTURE).  Written from the curve's public definition (BN254 G1: y^2 = x^3 + 3 over F_q, generator (1, 2), prime order r) and from
the reference's Rust sources only:
    multilinear bind / evaluate / eq table        crates/jolt-poly/src/dense.rs:188-303,340-366, src/eq.rs:50-98
    kzg_commit / witness polynomial / eval        crates/jolt-hyperkzg/src/kzg.rs:15-59
    fold_polynomials / open / kzg_open_batch      crates/jolt-hyperkzg/src/scheme.rs:88-158, kzg.rs:69-126
    JoltGroup::msm contract                       crates/jolt-crypto/src/ec/group.rs:63-70
Deliberately different machinery from the oracle's: homogeneous projective points with the COMPLETE addition law for j = 0 curves (Renes,
Costello, Batina 2016, Algorithm 7 with b3 = 9: no special cases to get wrong), least-significant-bit-first double-and-add, a fixed
4-bit unsigned Pippenger, plain Python integers everywhere (no Montgomery form).  Inputs follow the reference's deterministic recipes:
`DenseMember::with_sum` tables (crates/jolt-sumcheck/src/tests.rs:1129-1135: seed + 31 i + 11) and `synthetic_point` challenges
(crates/jolt-kernels/src/optimized/parity.rs:25-34: seed * 6364136223846793005 + 2 i + 3 mod 2^64)."""

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
B3 = 9
G = (1, 2, 1)
INF = (0, 1, 0)


# ---- recipes ---------------------------------------------------------------------------------------------------------------------
def dense_member_table(num_rounds, seed):
    return [(seed + 31 * i + 11) % R for i in range(1 << num_rounds)]


def synthetic_point(length, seed):
    return [((seed * 6364136223846793005) + 2 * i + 3) % (1 << 64) for i in range(length)]


# ---- multilinear polynomials -----------------------------------------------------------------------------------------------------
def bind_low_to_high(t, r):  # dense.rs:223-303: t[y] <- t[2y] + r (t[2y+1] - t[2y])
    return [(t[2 * y] + r * (t[2 * y + 1] - t[2 * y])) % R for y in range(len(t) // 2)]


def bind_high_to_low(t, r):  # dense.rs:188-220: t[i] <- t[i] + r (t[i + half] - t[i])
    h = len(t) // 2
    return [(t[i] + r * (t[i + h] - t[i])) % R for i in range(h)]


def eq_table(point):  # eq.rs:50-98, big-endian: point[0] pairs the most significant index bit
    out = [1]
    for x in point:
        out = [v for e in out for v in ((e * (1 - x)) % R, (e * x) % R)]
    return out


def evaluate(evals, point):  # dense.rs:340-366
    return sum(e * w for e, w in zip(evals, eq_table(point))) % R


# ---- G1, complete projective arithmetic --------------------------------------------------------------------------------------------
def padd(p, q):  # RCB16 Algorithm 7 (a = 0)
    X1, Y1, Z1 = p
    X2, Y2, Z2 = q
    t0, t1, t2 = X1 * X2 % Q, Y1 * Y2 % Q, Z1 * Z2 % Q
    t3 = ((X1 + Y1) * (X2 + Y2) - t0 - t1) % Q
    t4 = ((Y1 + Z1) * (Y2 + Z2) - t1 - t2) % Q
    Y3 = ((X1 + Z1) * (X2 + Z2) - t0 - t2) % Q
    X3 = 3 * t0 % Q
    t2 = B3 * t2 % Q
    Z3 = (t1 + t2) % Q
    t1 = (t1 - t2) % Q
    Y3 = B3 * Y3 % Q
    return ((t3 * t1 - t4 * Y3) % Q, (Y3 * X3 + t1 * Z3) % Q, (Z3 * t4 + X3 * t3) % Q)


def pneg(p):
    return (p[0], (-p[1]) % Q, p[2])


def pmul(p, k):  # least significant bit first
    k %= R
    acc, run = INF, p
    while k:
        if k & 1:
            acc = padd(acc, run)
        run = padd(run, run)
        k >>= 1
    return acc


def affine(p):
    if p[2] % Q == 0:
        return None
    zi = pow(p[2], -1, Q)
    return (p[0] * zi % Q, p[1] * zi % Q)


def on_curve(p):
    a = affine(p)
    return a is None or (a[1] * a[1] - a[0] ** 3 - 3) % Q == 0


def msm_naive(bases, scalars):
    acc = INF
    for b, s in zip(bases, scalars):
        acc = padd(acc, pmul(b, s))
    return acc


def msm_pippenger(bases, scalars, c=4):  # textbook unsigned windows, most significant window first
    windows = (254 + c - 1) // c
    total = INF
    for w in reversed(range(windows)):
        for _ in range(c):
            total = padd(total, total)
        buckets = [INF] * (1 << c)
        for b, s in zip(bases, scalars):
            d = ((s % R) >> (w * c)) & ((1 << c) - 1)
            if d:
                buckets[d] = padd(buckets[d], b)
        running, acc = INF, INF
        for d in range((1 << c) - 1, 0, -1):
            running = padd(running, buckets[d])
            acc = padd(acc, running)
        total = padd(total, acc)
    return total


# ---- KZG / HyperKZG --------------------------------------------------------------------------------------------------------------
def srs(beta, n):  # scheme.rs:54-73: g1_powers[i] = beta^i G
    out, cur = [], G
    for _ in range(n):
        out.append(cur)
        cur = pmul(cur, beta)
    return out


def kzg_commit(coeffs, powers):  # kzg.rs:15-27
    return msm_naive(powers[: len(coeffs)], coeffs)


def witness_polynomial(f, u):  # kzg.rs:34-46: h[i-1] = f[i] + h[i] u
    h, acc = [0] * (len(f) - 1), 0
    for i in range(len(f) - 1, 0, -1):
        acc = (f[i] + acc * u) % R
        h[i - 1] = acc
    return h


def eval_univariate(coeffs, u):  # kzg.rs:51-59
    return sum(c * pow(u, i, R) for i, c in enumerate(coeffs)) % R


def fold_polynomials(evals, point):  # scheme.rs:88-114: level i+1 = level i bound low-to-high by point[ell - 1 - i]
    polys = [list(evals)]
    for k in range(len(point) - 1, 0, -1):
        polys.append(bind_low_to_high(polys[-1], point[k]))
    return polys


def open_given_challenges(powers, evals, point, r, q):
    """HyperKZGScheme::open + kzg_open_batch with the two transcript challenges supplied: (level commitments, v[3][ell], witness commitments)"""
    polys = fold_polynomials(evals, point)
    coms = [kzg_commit(p, powers) for p in polys[1:]]
    u = [r % R, (-r) % R, r * r % R]
    v = [[eval_univariate(p, ui) for p in polys] for ui in u]
    b = [0] * len(polys[0])
    qj = 1
    for p in polys:
        for i, c in enumerate(p):
            b[i] = (b[i] + qj * c) % R
        qj = qj * q % R
    ws = [kzg_commit(witness_polynomial(b, ui), powers) for ui in u]
    return coms, v, ws
