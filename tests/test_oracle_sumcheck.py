"""Pin the oracle's sumcheck members and batched round loop with the reference's own checks
(/root/reference/crates/jolt-sumcheck/src/prover.rs:316-324 round check, naive.rs:298-309,
optimized/parity.rs:79-118 lockstep equality, jolt-sumcheck/src/tests.rs:1123-1180 DenseMember)."""
import random

import numpy as np

import oracle_lib as O

R = O.R_MOD


def model_expr_sum(tables, terms):
    n = len(tables[0])
    tot = 0
    for x in range(n):
        for c, f in terms:
            v = c
            for k in f:
                v = v * tables[k][x] % R
            tot += v
    return tot % R


def build(rng, n_vars, n_tables, terms_idx):
    tabs = [[rng.randrange(R) for _ in range(1 << n_vars)] for _ in range(n_tables)]
    terms = [(rng.randrange(R), f) for f in terms_idx]
    return tabs, terms


def run_single(member, claim_limbs, n_vars, challenges):
    claim = claim_limbs
    bind = None
    polys = []
    for rnd in range(n_vars):
        coeffs = member.prove_round(bind, claim)
        polys.append(coeffs)
        bind = challenges[rnd]
        claim = O.univariate_evaluate(coeffs, bind)
    member.finish_rounds(bind)
    return polys, claim


def test_expr_member_rounds_and_final_claim():
    rng = random.Random(11)
    n_vars = 5
    for order in (O.ORDER_LOW_TO_HIGH, O.ORDER_HIGH_TO_LOW):
        # eq-like * (a + g b) shape (degree 2) and a triple product (degree 3)
        for degree, terms_idx, n_tables in ((2, [[0, 1], [0, 2]], 3), (3, [[0, 1, 2]], 3), (3, [[0, 1, 1], [0, 1]], 2)):
            tabs, terms = build(rng, n_vars, n_tables, terms_idx)
            tm = [O.to_mont(t) for t in tabs]
            mterms = [(O.to_mont([c])[0], f) for c, f in terms]
            m = O.Member.expr(tm, mterms, degree, order)
            claim = O.to_mont([model_expr_sum(tabs, terms)])[0]
            assert np.array_equal(m.input_claim(), claim)
            ch = [rng.randrange(R) for _ in range(n_vars)]
            chm = O.to_mont(ch)
            polys, final_claim = run_single(m, claim, n_vars, chm)
            # final claim == summand at the bound point (tables evaluated at the point)
            point = ch if order == O.ORDER_HIGH_TO_LOW else list(reversed(ch))
            vals = [O.from_mont(O.poly_evaluate(t, O.to_mont(point)))[0] for t in tm]
            want = 0
            for c, f in terms:
                v = c
                for k in f:
                    v = v * vals[k] % R
                want = (want + v) % R
            assert O.from_mont(final_claim) == [want]
            assert O.from_mont(m.final_values()) == vals
            # skipped-evals assembly (support.rs:450-459) yields identical coefficient vectors
            m2 = O.Member.expr(tm, mterms, degree, order, skip_one=True)
            polys2, final2 = run_single(m2, claim, n_vars, chm)
            for p, q in zip(polys, polys2):
                assert np.array_equal(p, q)


def test_round_check_failure_is_reported():
    rng = random.Random(12)
    tabs, terms = build(rng, 3, 2, [[0, 1]])
    m = O.Member.expr([O.to_mont(t) for t in tabs], [(O.to_mont([c])[0], f) for c, f in terms], 2)
    bad = O.to_mont([12345])[0]
    try:
        m.prove_round(None, bad)
        assert False, "expected RoundCheckFailed"
    except RuntimeError:
        pass


def test_gruen_product_member_equals_dense_eq_member():
    # split_eq.rs tests + optimized-vs-reference lockstep: eq(w,.)*a*b served from split tables must emit the
    # same cubic as the dense three-table product member
    rng = random.Random(13)
    for n_vars in (1, 2, 3, 6, 7):
        N = 1 << n_vars
        a = [rng.randrange(R) for _ in range(N)]
        b = [rng.randrange(R) for _ in range(N)]
        w = [rng.randrange(R) for _ in range(n_vars)]
        am, bm, wm = O.to_mont(a), O.to_mont(b), O.to_mont(w)
        eq = O.eq_evals(wm)
        one = O.to_mont([1])[0]
        dense = O.Member.expr([eq, am, bm], [(one, [0, 1, 2])], 3, O.ORDER_LOW_TO_HIGH)
        gruen = O.Member.gruen_product(am, bm, wm)
        claim = dense.input_claim()
        assert np.array_equal(claim, gruen.input_claim())
        ch = O.to_mont([rng.randrange(R) for _ in range(n_vars)])
        p1, f1 = run_single(dense, claim, n_vars, ch)
        p2, f2 = run_single(gruen, claim, n_vars, ch)
        for x, y in zip(p1, p2):
            assert np.array_equal(x, y)
        assert np.array_equal(f1, f2)
        fv_d, fv_g = dense.final_values(), gruen.final_values()
        assert np.array_equal(fv_d[1:], fv_g[:2])
        assert np.array_equal(fv_d[0], fv_g[2])  # bound eq scalar == dense eq table's final value


def test_prove_batch_mixed_rounds_and_offsets():
    # prover.rs:193-362: members of different length, tail-aligned and head-aligned; DenseMember::with_sum recipe
    rng = random.Random(14)
    one = O.to_mont([1])[0]
    # (rounds, offset): shorter members are tail-aligned (offset = max - rounds); a head-aligned short member
    # would have to emit polynomials at the padded 2^(max-rounds) scale itself (batch.rs:48-55)
    specs = [(5, 0), (3, 2), (4, 1), (5, 0)]
    members, claims = [], []
    for idx, (rounds, _) in enumerate(specs):
        N = 1 << rounds
        if idx == 0:  # tests.rs:1129-1135 with_sum fixture values (linear member, degree 1 inside degree-3 batch)
            seed = 7
            evals = [(seed + 31 * i + 11) % R for i in range(N)]
            m = O.Member.expr([O.to_mont(evals)], [(one, [0])], 1, O.ORDER_HIGH_TO_LOW)
        elif idx == 3:
            a = O.to_mont([rng.randrange(R) for _ in range(N)])
            b = O.to_mont([rng.randrange(R) for _ in range(N)])
            m = O.Member.gruen_product(a, b, O.to_mont([rng.randrange(R) for _ in range(rounds)]))
        else:
            tabs = [O.to_mont([rng.randrange(R) for _ in range(N)]) for _ in range(3)]
            m = O.Member.expr(tabs, [(O.to_mont([rng.randrange(R)])[0], [0, 1]), (one, [0, 2, 2])], 3)
        members.append(m)
        claims.append(m.input_claim())
    coeffs = [O.to_mont([rng.randrange(R)])[0] for _ in specs]
    for mode in (0,):
        out = O.prove_batch(members, claims, coeffs, [o for _, o in specs], 5, 3, label=99, challenge_mode=mode)
        # 125-bit challenge shape: two low Montgomery limbs are zero (mod.rs:254)
        assert all(int(c[0]) == 0 and int(c[1]) == 0 for c in out["challenges"])
        # final claim == sum coeff_i * member final claim
        tot = 0
        for c, mc in zip(coeffs, out["member_claims"]):
            tot = (tot + O.from_mont(c)[0] * O.from_mont(mc)[0]) % R
        assert O.from_mont(out["final_claim"]) == [tot]
        # every member's final claim equals its summand at its own opening point
        for m, mc in zip(members, out["member_claims"]):
            fv = O.from_mont(m.final_values())
            if m.gruen:
                assert O.from_mont(mc) == [fv[0] * fv[1] * fv[2] % R]


def test_triple_product_round_evals_matches_member():
    rng = random.Random(15)
    N = 64
    a, b, c = [O.to_mont([rng.randrange(R) for _ in range(N)]) for _ in range(3)]
    one = O.to_mont([1])[0]
    m = O.Member.expr([a, b, c], [(one, [0, 1, 2])], 3)
    coeffs = m.prove_round(None, m.input_claim())
    s = [O.univariate_evaluate(coeffs, O.to_mont([t])[0]) for t in (0, 2, 3)]
    got = O.triple_product_round_evals(a, b, c)
    for x, y in zip(s, got):
        assert np.array_equal(x, y)
