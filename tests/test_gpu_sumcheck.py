"""GPU parity: sumcheck members and the batched round loop through the C ABI vs the CPU oracle, bit-exact.
Mirrors the reference's tier-parity lockstep (crates/jolt-kernels/src/optimized/parity.rs:79-118: equal
coefficients every round, equal output claims) with the oracle in the role of the reference tier."""
import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import ffi
from util import rand_challenge, rand_fr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = ffi.Context(0)
    yield c
    c.close()


ONE = None


def one():
    global ONE
    if ONE is None:
        ONE = O.to_mont([1])[0]
    return ONE


def lockstep(dev, orc, n_vars, seed, shifted=True):
    """run_lockstep: same claim and challenges into both members, coefficients compared every round."""
    claim = orc.input_claim()
    assert np.array_equal(dev.input_claim(), claim)
    bind = None
    for rnd in range(n_vars):
        want = orc.prove_round(bind, claim)
        got = dev.prove_round(bind, want_aux=True)
        evals, aux = got
        if dev.uniform:
            coeffs = ffi.host_gruen_poly_from_q(aux[0], aux[1], evals, claim)
        elif dev.split_eq:
            coeffs = ffi.host_gruen_poly_deg_3(aux[0], aux[1], evals[0], evals[1], claim)
        elif dev.skip_one:
            full = np.concatenate([evals[:1], O.fr_sub(claim.reshape(1, 4), evals[:1]), evals[1:]])
            coeffs = ffi.host_univariate_from_evals(full)
        else:
            assert np.array_equal(O.fr_add(evals[:1], evals[1:2])[0], claim), f"round {rnd}: s(0)+s(1) != claim"
            coeffs = ffi.host_univariate_from_evals(evals)
        assert np.array_equal(coeffs, want), f"round {rnd}"
        bind = rand_challenge(seed + rnd, shifted)
        claim = O.univariate_evaluate(want, bind)
    orc.finish_rounds(bind)
    dev.finish(bind)
    assert np.array_equal(dev.final_values(), orc.final_values())
    return claim


CATALOGUE = [
    # (name, degree, n_tables, terms as table-index lists) -- SURVEY.md 8 a13 summand shapes
    ("linear", 1, 1, [[0]]),
    ("claim_reduction eq*(o1+g o2+..)", 2, 4, [[0, 1], [0, 2], [0, 3]]),
    ("triple product", 3, 3, [[0, 1, 2]]),
    ("hamming booleanity eq*(H^2-H)", 3, 2, [[0, 1, 1], [0, 1]]),
    ("instruction_input", 3, 5, [[0, 1, 2], [0, 3, 4]]),
    ("ra virtualization deg5", 5, 5, [[0, 1, 2, 3, 4]]),
    ("with constant term", 2, 2, [[0, 1], []]),
]


@pytest.mark.parametrize("name,degree,n_tables,terms_idx", CATALOGUE)
@pytest.mark.parametrize("order", [ffi.ORDER_LOW_TO_HIGH, ffi.ORDER_HIGH_TO_LOW])
def test_expr_member_lockstep_with_oracle(ctx, name, degree, n_tables, terms_idx, order):
    n_vars = 7
    tabs = [rand_fr(1 << n_vars, 1000 + 17 * k + degree) for k in range(n_tables)]
    coeffs = rand_fr(len(terms_idx), 2000 + degree)
    coeffs[0] = one()  # exercise the coefficient-one fast path too
    terms = [(coeffs[i], f) for i, f in enumerate(terms_idx)]
    orc = O.Member.expr(tabs, terms, degree, order)
    dev = ctx.member_expr([ctx.upload(t) for t in tabs], terms, degree, order)
    lockstep(dev, orc, n_vars, 3000 + degree, shifted=(order == ffi.ORDER_LOW_TO_HIGH))


def test_lc_member_equals_flat_expr_member(ctx):
    """Descriptor rewrite (linear-leaf fusion, optimized/inc_claim_reduction.rs:68-89): eq1*s1 + eq2*s2 fused into one
    factor must emit the same polynomials as the flat four-term expression; skip-one recovers s(1) from the claim."""
    n_vars = 8
    eq1, eq2, a, b = [rand_fr(1 << n_vars, 4000 + k) for k in range(4)]
    s1, s2, g = rand_fr(3, 4100)
    # flat: s1*eq1*a + s2*eq2*a + g*s1... keep it simple: (s1 eq1 + s2 eq2) * (a + g b)
    flat_terms = [(s1, [0, 2]), (s2, [1, 2]), (O.fr_mul(s1.reshape(1, 4), g.reshape(1, 4))[0], [0, 3]),
                  (O.fr_mul(s2.reshape(1, 4), g.reshape(1, 4))[0], [1, 3])]
    orc = O.Member.expr([eq1, eq2, a, b], flat_terms, 2)
    groups = [[(None, [(s1, 0), (s2, 1)]), (None, [(one(), 2), (g, 3)])]]
    for skip in (False, True):
        orc = O.Member.expr([eq1, eq2, a, b], flat_terms, 2)
        dev = ctx.member_lc([ctx.upload(t) for t in (eq1, eq2, a, b)], groups, 2, skip_one=skip)
        lockstep(dev, orc, n_vars, 4200 + int(skip))


@pytest.mark.parametrize("n_vars", [1, 2, 3, 6, 9, 12])
def test_split_eq_product_member_lockstep(ctx, n_vars):
    a, b = rand_fr(1 << n_vars, 5000 + n_vars), rand_fr(1 << n_vars, 5100 + n_vars)
    w = rand_fr(n_vars, 5200 + n_vars)
    scale = rand_fr(1, 5300 + n_vars)[0] if n_vars % 2 else None
    orc = O.Member.gruen_product(a, b, w, scale)
    dev = ctx.member_split_eq_product(ctx.upload(a), ctx.upload(b), w, scale)
    lockstep(dev, orc, n_vars, 5400 + n_vars)


def test_member_error_behaviour(ctx):
    t = rand_fr(16, 1)
    with pytest.raises(ffi.JoltError):  # KernelError::TableSizeMismatch
        ctx.member_expr([ctx.upload(t), ctx.upload(t[:8])], [(one(), [0, 1])], 2)
    with pytest.raises(ffi.JoltError):  # factor index out of range
        ctx.member_expr([ctx.upload(t)], [(one(), [1])], 1)
    m = ctx.member_expr([ctx.upload(t)], [(one(), [0])], 1)
    with pytest.raises(ffi.JoltError) as e:  # SumcheckKernelError::NotFullyBound
        m.final_values()
    assert e.value.status == 7


def build_batch(ctx, seed):
    """Four members of different length / kind, tail-aligned (prover.rs:193-362 shapes)."""
    specs = [(6, 0, "linear"), (4, 2, "cubic"), (5, 1, "quad"), (6, 0, "spliteq")]
    orcs, devs = [], []
    for idx, (rounds, _, kind) in enumerate(specs):
        N = 1 << rounds
        if kind == "linear":
            evals = O.fr_from_u64(np.array([7 + 31 * i + 11 for i in range(N)], dtype=np.uint64))  # tests.rs:1129-1135
            orcs.append(O.Member.expr([evals], [(one(), [0])], 1))
            devs.append(ctx.member_expr([ctx.upload(evals)], [(one(), [0])], 1))
        elif kind == "spliteq":
            a, b, w = rand_fr(N, seed + 1), rand_fr(N, seed + 2), rand_fr(rounds, seed + 3)
            orcs.append(O.Member.gruen_product(a, b, w))
            devs.append(ctx.member_split_eq_product(ctx.upload(a), ctx.upload(b), w))
        else:
            d = 3 if kind == "cubic" else 2
            tabs = [rand_fr(N, seed + 10 * idx + k) for k in range(3)]
            terms = [(rand_fr(1, seed + 77 + idx)[0], [0, 1] if d == 2 else [0, 1, 2]), (one(), [0, 2])]
            orcs.append(O.Member.expr(tabs, terms, d))
            devs.append(ctx.member_expr([ctx.upload(t) for t in tabs], terms, d))
    claims = [o.input_claim() for o in orcs]
    coeffs = list(rand_fr(len(specs), seed + 99))
    offsets = [o for _, o, _ in specs]
    return orcs, devs, claims, coeffs, offsets


@pytest.mark.parametrize("grouped", [False, True])
@pytest.mark.parametrize("challenge_mode", [0, 1])
def test_prove_batch_bit_exact_with_oracle(ctx, grouped, challenge_mode):
    orcs, devs, claims, coeffs, offsets = build_batch(ctx, 6000)
    want = O.prove_batch(orcs, claims, coeffs, offsets, 6, 3, label=42, challenge_mode=challenge_mode)
    got = ctx.prove_batch(devs, claims, coeffs, offsets, 6, 3, label=42, challenge_mode=challenge_mode, use_round_group=grouped)
    assert np.array_equal(got["polys"], want["polys"])
    assert np.array_equal(got["challenges"], want["challenges"])
    assert np.array_equal(got["member_claims"], want["member_claims"])
    assert np.array_equal(got["final_claim"], want["final_claim"])
    for d, o in zip(devs, orcs):
        assert np.array_equal(d.final_values(), o.final_values())


def test_prove_batch_reports_round_check_failure(ctx):
    orcs, devs, claims, coeffs, offsets = build_batch(ctx, 6100)
    claims[1] = rand_fr(1, 1)[0]  # wrong input claim -> SumcheckError::RoundCheckFailed
    with pytest.raises(ffi.JoltError) as e:
        ctx.prove_batch(devs, claims, coeffs, offsets, 6, 3, label=1)
    assert e.value.status == 8


def test_full_size_sumcheck_properties(ctx):
    """T = 2^20 (BASELINE configs[1] scale): too large for the scalar oracle, so check the protocol's own invariants:
    s(0)+s(1) == claim every round (enforced inside prove_batch), and the final claim equals the summand evaluated
    at the bound point using independently computed multilinear evaluations (Polynomial::evaluate)."""
    n = 20
    tabs = [rand_fr(1 << n, 7000 + k) for k in range(3)]
    g = rand_fr(1, 7100)[0]
    terms = [(one(), [0, 1]), (g, [0, 2])]
    dev = ctx.member_expr([ctx.upload(t) for t in tabs], terms, 2)
    keep = [ctx.upload(t) for t in tabs]
    claim = dev.input_claim()
    out = ctx.prove_batch([dev], [claim], [one()], [0], n, 2, label=5)
    point = out["challenges"][::-1]  # LowToHigh binds consume the point back to front
    vals = [ctx.evaluate(t, point) for t in keep]
    want = O.fr_add(O.fr_mul(vals[0].reshape(1, 4), vals[1].reshape(1, 4)),
                    O.fr_mul(O.fr_mul(g.reshape(1, 4), vals[0].reshape(1, 4)), vals[2].reshape(1, 4)))[0]
    assert np.array_equal(out["final_claim"], want)
    assert np.array_equal(dev.final_values(), np.stack(vals))


def test_borrowed_members_leave_tables_intact_and_can_be_reset(ctx):
    """BORROW members bind into their own scratch: the shared resident tables stay bit-identical, two members can
    share one table, and a reset member re-proves to the same transcript."""
    n_vars = 8
    eq, a, b = [rand_fr(1 << n_vars, 8000 + k) for k in range(3)]
    g = rand_fr(1, 8100)[0]
    shared = [ctx.upload(t) for t in (eq, a, b)]
    groups1 = [[(None, [(one(), 0)]), (None, [(one(), 1), (g, 2)])]]
    groups2 = [[(None, [(one(), 0)]), (None, [(one(), 1)]), (None, [(one(), 2)])]]
    for order in (ffi.ORDER_LOW_TO_HIGH, ffi.ORDER_HIGH_TO_LOW):
        m1 = ctx.member_lc(shared, groups1, 2, order=order, borrow=True)
        m2 = ctx.member_lc(shared, groups2, 3, order=order, borrow=True)
        w = rand_fr(n_vars, 8200)
        m3 = ctx.member_split_eq_product(shared[1], shared[2], w, borrow=True)
        o1 = lambda: O.Member.expr([eq, a, b], [(one(), [0, 1]), (g, [0, 2])], 2, order)
        o2 = lambda: O.Member.expr([eq, a, b], [(one(), [0, 1, 2])], 3, order)
        o3 = lambda: O.Member.gruen_product(a, b, w)
        for rep in range(2):
            orcs = [o1(), o2(), o3()]
            claims = [o.input_claim() for o in orcs]
            coeffs = list(rand_fr(3, 8300))
            want = O.prove_batch(orcs, claims, coeffs, [0, 0, 0], n_vars, 3, label=7 + rep)
            got = ctx.prove_batch([m1, m2, m3], claims, coeffs, [0, 0, 0], n_vars, 3, label=7 + rep)
            assert np.array_equal(got["polys"], want["polys"]) and np.array_equal(got["final_claim"], want["final_claim"])
            for t, h in zip(shared, (eq, a, b)):
                assert np.array_equal(t.download(), h)
            for m in (m1, m2, m3):
                m.reset()
        for m in (m1, m2, m3):
            m.destroy()
    owned = ctx.member_expr([ctx.upload(a)], [(one(), [0])], 1)
    with pytest.raises(ffi.JoltError):
        owned.reset()


@pytest.mark.parametrize("V,F,n_vars", [(1, 2, 6), (1, 3, 7), (3, 3, 5), (1, 4, 8), (8, 4, 7), (2, 4, 1), (8, 4, 13)])
def test_split_eq_uniform_product_member_lockstep(ctx, V, F, n_vars):
    """eq * sum_v c_v prod_{i<F} T_{v,i} served from split tables with the product tree on evaluation points must emit the
    polynomials of the oracle's flat Expr member over the dense eq table (optimized tier vs reference tier)."""
    N = 1 << n_vars
    w = rand_fr(n_vars, 9000 + F)
    tabs = [rand_fr(N, 9100 + k) for k in range(V * F)]
    coeffs = rand_fr(V, 9200 + V)
    coeffs[0] = one()
    scale = rand_fr(1, 9300)[0] if V == 3 else None
    eq = O.eq_evals(w, scale)
    terms = [(coeffs[v], [0] + [1 + v * F + k for k in range(F)]) for v in range(V)]
    orc = O.Member.expr([eq] + tabs, terms, F + 1)
    dev = ctx.member_split_eq_uniform([ctx.upload(t) for t in tabs], V, F, coeffs, w, scale=scale)
    claim = orc.input_claim()
    assert np.array_equal(dev.input_claim(), claim)
    bind = None
    for rnd in range(n_vars):
        want = orc.prove_round(bind, claim)
        evals, aux = dev.prove_round(bind, want_aux=True)
        got = ffi.host_gruen_poly_from_q(aux[0], aux[1], evals, claim)
        assert np.array_equal(got, want), f"round {rnd}"
        bind = rand_challenge(9400 + rnd)
        claim = O.univariate_evaluate(want, bind)
    orc.finish_rounds(bind)
    dev.finish(bind)
    fv, ofv = dev.final_values(), orc.final_values()
    assert np.array_equal(fv[:-1], ofv[1:]) and np.array_equal(fv[-1], ofv[0])  # bound tables, then the bound eq scalar


@pytest.mark.parametrize("n_vars", [1, 5, 9, 14])
def test_eq_weighted_lc_member_matches_oracle(ctx, n_vars):
    """eq(w,j) * q(j) with the eq weight factored out (jolt_member_create_split_eq_lc): q = (f1*a + f2*b) + g*(f3*c) + const-factor
    group, inner degree 2 -> two sums per round, s = l*q on the host.  Lock-step against the oracle's flat Expr member over the
    dense eq table; borrowed tables, both challenge shapes, tail and grouped kernels (n_vars 14 crosses the tail threshold), and
    again inside prove_batch next to an ordinary member."""
    N = 1 << n_vars
    w = rand_fr(n_vars, 5100)
    scale = rand_fr(1, 5101)[0]
    tabs = [rand_fr(N, 5110 + k) for k in range(6)]
    g, c0 = rand_fr(1, 5120)[0], rand_fr(1, 5121)[0]
    o = one()
    dev_tabs = [ctx.upload(t) for t in tabs]
    inner = [[(None, [(o, 0)]), (None, [(o, 1)])],          # f1 * a
             [(None, [(o, 2)]), (None, [(o, 3)])],          # f2 * b
             [(None, [(g, 4)]), (c0, [(o, 5)])]]            # g*f3 * (c0 + c)
    dev = ctx.member_lc(dev_tabs, inner, 2, borrow=True, eq_point=w, eq_scale=scale)
    eq = O.eq_evals(w, scale)
    mul = lambda a, b: O.fr_mul(np.asarray(a).reshape(1, 4), np.asarray(b).reshape(1, 4))[0]
    terms = [(o, [0, 1, 2]), (o, [0, 3, 4]), (g, [0, 5, 6]), (mul(g, c0), [0, 5])]
    def fresh():
        return O.Member.expr([eq] + tabs, terms, 3)
    for rep in range(2):
        orc = fresh()
        claim = orc.input_claim()
        if rep == 0:
            assert np.array_equal(dev.input_claim(), claim)
        bind = None
        for rnd in range(n_vars):
            want = orc.prove_round(bind, claim)
            evals, aux = dev.prove_round(bind, want_aux=True)
            got = ffi.host_gruen_poly_from_q(aux[0], aux[1], evals, claim)
            assert np.array_equal(got, want), f"rep {rep} round {rnd}"
            bind = rand_challenge(5130 + rnd) if rnd % 2 == 0 else rand_fr(1, 5130 + rnd)[0]
            claim = O.univariate_evaluate(want, bind)
        orc.finish_rounds(bind)
        dev.finish(bind)
        fv, ofv = dev.final_values(), orc.final_values()
        assert np.array_equal(fv[:-1], ofv[1:]) and np.array_equal(fv[-1], ofv[0])
        dev.reset()
    a, b = rand_fr(N, 5140), rand_fr(N, 5141)
    flat = ctx.member_expr([ctx.upload(a), ctx.upload(b)], [(o, [0, 1])], 2)
    orcs = [fresh(), O.Member.expr([a, b], [(o, [0, 1])], 2)]
    claims = [m.input_claim() for m in orcs]
    cf = list(rand_fr(2, 5150))
    want = O.prove_batch(orcs, claims, cf, [0, 0], n_vars, 3, label=6)
    got = ctx.prove_batch([dev, flat], claims, cf, [0, 0], n_vars, 3, label=6)
    for k in ("polys", "challenges", "member_claims", "final_claim"):
        assert np.array_equal(got[k], want[k]), k
