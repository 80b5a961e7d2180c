"""GPU parity: dense-table kernels through the C ABI vs the CPU oracle, bit-exact (integer field arithmetic).
Mirrors /root/reference/crates/jolt-poly/src/dense.rs:578-1119 and eq.rs:475-756 test shapes."""
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


@pytest.mark.parametrize("n_vars", [1, 2, 5, 10, 13])
@pytest.mark.parametrize("shifted", [True, False])
def test_bind_both_orders_match_oracle(ctx, n_vars, shifted):
    t = rand_fr(1 << n_vars, 100 + n_vars)
    r = rand_challenge(7 + n_vars, shifted)
    for order, ref in ((ffi.ORDER_LOW_TO_HIGH, O.bind_low_to_high), (ffi.ORDER_HIGH_TO_LOW, O.bind_high_to_low)):
        tab = ctx.upload(t)
        ctx.bind([tab], r, order)
        assert len(tab) == (1 << n_vars) // 2
        assert np.array_equal(tab.download(), ref(t, r))


def test_bind_many_tables_one_launch_and_repeated_rounds(ctx):
    n_vars = 9
    tabs_h = [rand_fr(1 << n_vars, 200 + k) for k in range(45)]  # > one launch batch (40)
    tabs = [ctx.upload(t) for t in tabs_h]
    for rnd in range(n_vars):
        r = rand_challenge(300 + rnd, shifted=(rnd % 2 == 0))
        ctx.bind(tabs, r, ffi.ORDER_LOW_TO_HIGH)
        tabs_h = [O.bind_low_to_high(t, r) for t in tabs_h]
    for t, th in zip(tabs, tabs_h):
        assert len(t) == 1 and np.array_equal(t.download(), th)


def test_bind_edge_values_and_errors(ctx):
    # 0, 1, r-1 and a challenge of r-1 / 0 / 1 (conditional-subtraction corners)
    vals = O.to_mont([0, 1, O.R_MOD - 1, O.R_MOD - 1, 0, O.R_MOD - 2, 5, O.R_MOD - 5])
    for c in O.to_mont([0, 1, O.R_MOD - 1]):
        tab = ctx.upload(vals)
        ctx.bind([tab], c, ffi.ORDER_LOW_TO_HIGH)
        assert np.array_equal(tab.download(), O.bind_low_to_high(vals, c))
    one = ctx.upload(vals[:1])
    with pytest.raises(ffi.JoltError):  # dense.rs:225 "cannot bind a zero-variable polynomial"
        ctx.bind([one], vals[1], ffi.ORDER_LOW_TO_HIGH)
    a, b = ctx.upload(vals), ctx.upload(vals[:4])
    with pytest.raises(ffi.JoltError):
        ctx.bind([a, b], vals[1], ffi.ORDER_LOW_TO_HIGH)
    bad = np.array([2**64 - 1] * 4, dtype=np.uint64)  # not canonical
    with pytest.raises(ffi.JoltError):
        ctx.bind([a], bad, ffi.ORDER_LOW_TO_HIGH)


@pytest.mark.parametrize("n", [0, 1, 3, 8, 9, 12, 17])
def test_eq_evals_match_oracle(ctx, n):
    r = rand_fr(n, 400 + n) if n else np.zeros((0, 4), dtype=np.uint64)
    assert np.array_equal(ctx.eq_evals(r).download(), O.eq_evals(r))
    s = rand_fr(1, 500 + n)[0]
    assert np.array_equal(ctx.eq_evals(r, s).download(), O.eq_evals(r, s))


def test_eq_aligned_block_is_a_slice_of_the_full_table(ctx):
    n = 11
    r = rand_fr(n, 600)
    full = O.eq_evals(r)
    for block, start in ((1, 5), (4, 8), (256, 1024), (1024, 1024), (2048, 0)):
        got = ctx.eq_evals_aligned_block(r, start, block).download()
        assert np.array_equal(got, full[start:start + block])
    with pytest.raises(ffi.JoltError):
        ctx.eq_evals_aligned_block(r, 3, 4)  # eq.rs:245 misaligned start


@pytest.mark.parametrize("n", [1, 4, 8, 11, 14])
def test_lt_and_eq_plus_one_match_oracle(ctx, n):
    r = rand_fr(n, 700 + n)
    assert np.array_equal(ctx.lt_evals(r).download(), O.lt_evals(r))
    eq, eq1 = ctx.eq_plus_one_evals(r)
    oeq, oeq1 = O.eq_plus_one_evals(r)
    assert np.array_equal(eq.download(), oeq)
    assert np.array_equal(eq1.download(), oeq1)
    s = rand_fr(1, 800 + n)[0]
    eq, eq1 = ctx.eq_plus_one_evals(r, s)
    oeq, oeq1 = O.eq_plus_one_evals(r, s)
    assert np.array_equal(eq.download(), oeq) and np.array_equal(eq1.download(), oeq1)


def test_small_scalar_promotion_and_sums(ctx):
    rng = np.random.default_rng(9)
    u = rng.integers(0, 2**64, size=1000, dtype=np.uint64)
    u[:4] = [0, 1, (1 << 14) - 1, 1 << 14]
    assert np.array_equal(ctx.from_u64(u).download(), O.fr_from_u64(u))
    i = rng.integers(-2**63, 2**63, size=1000, dtype=np.int64)
    i[:3] = [0, -1, -2**63]
    assert np.array_equal(ctx.from_i64(i).download(), O.fr_from_i64(i))
    t = rand_fr(1 << 10, 10)
    tab = ctx.upload(t)
    want = O.fr_array(1)
    acc = np.zeros((1, 4), dtype=np.uint64)
    for row in t:
        acc = O.fr_add(acc, row.reshape(1, 4))
    assert np.array_equal(ctx.table_sum(tab), acc[0])
    pt = rand_fr(10, 11)
    assert np.array_equal(ctx.evaluate(tab, pt), O.poly_evaluate(t, pt))


def test_full_size_bind_chain_equals_evaluate(ctx):
    """Size-independent property at the BASELINE scale (T = 2^20): binding every variable low-to-high with
    r_{n-1}..r_0 yields f(r) = <f, eq(r)>, computed two different ways on the device and anchored to the oracle on
    a 2^12 prefix slice."""
    n = 20
    t = rand_fr(1 << n, 12)
    pt = np.stack([rand_challenge(900 + k) for k in range(n)])
    tab = ctx.upload(t)
    want = ctx.evaluate(tab, pt)
    for k in reversed(range(n)):
        ctx.bind([tab], pt[k], ffi.ORDER_LOW_TO_HIGH)
    assert np.array_equal(tab.download()[0], want)
    # oracle anchor: the first bind of the leading 2^12 entries
    tab2 = ctx.upload(t)
    ctx.bind([tab2], pt[n - 1], ffi.ORDER_LOW_TO_HIGH)
    assert np.array_equal(tab2.download(0, 1 << 11), O.bind_low_to_high(t[: 1 << 12], pt[n - 1]))


def test_views_and_rlc_match_reference_definitions(ctx):
    """address_fold / cycle_fold / tile / replicate_stream_lsb (reference/views.rs:35-138) and the RLC joint polynomial
    (multilinear.rs:358-464), checked against their defining sums computed with the oracle's field ops."""
    log_k, log_t = 3, 6
    K, T = 1 << log_k, 1 << log_t
    grid = rand_fr(K * T, 900)
    pk, pt = rand_fr(log_k, 901), rand_fr(log_t, 902)
    eqk, eqt = O.eq_evals(pk), O.eq_evals(pt)
    g = grid.reshape(K, T, 4)
    want_a = np.zeros((T, 4), dtype=np.uint64)
    for k in range(K):
        want_a = O.fr_add(want_a, O.fr_mul(g[k], np.repeat(eqk[k:k + 1], T, axis=0)))
    want_c = np.zeros((K, 4), dtype=np.uint64)
    for k in range(K):
        prod = O.fr_mul(g[k], eqt)
        acc = np.zeros((1, 4), dtype=np.uint64)
        for row in prod:
            acc = O.fr_add(acc, row.reshape(1, 4))
        want_c[k] = acc[0]
    dg = ctx.upload(grid)
    assert np.array_equal(ctx.address_fold(dg, ctx.eq_evals(pk)).download(), want_a)
    assert np.array_equal(ctx.cycle_fold(dg, ctx.eq_evals(pt)).download(), want_c)
    with pytest.raises(ffi.JoltError):
        ctx.address_fold(dg, ctx.upload(grid[:7]))
    base = rand_fr(16, 903)
    db = ctx.upload(base)
    assert np.array_equal(ctx.tile(db, 5).download(), np.tile(base, (5, 1)))
    assert np.array_equal(ctx.replicate_stream_lsb(db).download(), np.repeat(base, 2, axis=0))
    tabs = [rand_fr(256, 910 + i) for i in range(5)]
    gamma = rand_fr(1, 920)
    sc = [O.to_mont([1])]
    for _ in range(4):
        sc.append(O.fr_mul(sc[-1], gamma))
    sc = np.concatenate(sc, axis=0)
    want = np.zeros((256, 4), dtype=np.uint64)
    for t, s in zip(tabs, sc):
        want = O.fr_add(want, O.fr_mul(t, np.repeat(s.reshape(1, 4), 256, axis=0)))
    assert np.array_equal(ctx.rlc([ctx.upload(t) for t in tabs], sc).download(), want)
    with pytest.raises(ffi.JoltError):
        ctx.rlc([ctx.upload(tabs[0]), ctx.upload(tabs[1][:128])], sc[:2])


@pytest.mark.parametrize("n", [0, 1, 2, 5, 8, 11])
def test_split_lt_equals_dense_lt_plus_constant_bound_identically(ctx, n):
    """SplitLt (optimized/support.rs:640-760): at every stage -- split state, the fold of the lo scalar into the hi table, dense
    state -- the served evaluations equal LtPolynomial::evaluations(r) + constant bound low-to-high with the same challenges."""
    r = rand_fr(n, 1300 + n)
    c = rand_fr(1, 1310)[0]
    for constant in (None, c):
        want = O.lt_evals(r) if n else np.zeros((1, 4), dtype=np.uint64)
        if constant is not None:
            want = O.fr_add(want, np.repeat(constant.reshape(1, 4), want.shape[0], axis=0))
        s = ffi.SplitLt(ctx, r, constant)
        assert len(s) == 1 << n
        assert np.array_equal(s.to_dense().download(), want)
        if n:
            with pytest.raises(ffi.JoltError) as e:
                s.final_value()
            assert e.value.status == 7  # NotFullyBound
        for k in range(n):
            ch = rand_challenge(1320 + k) if k % 2 == 0 else rand_fr(1, 1320 + k)[0]
            s.bind(ch)
            want = O.bind_low_to_high(want, ch)
            assert len(s) == want.shape[0]
            assert np.array_equal(s.to_dense().download(), want), (n, k)
        assert np.array_equal(s.final_value(), want[0])
        s.free()


def test_deferred_reduction_accumulator_on_the_device():
    """WideAccumulator on the device (field.hip.h WideAcc: unreduced 512-bit sums of products, one REDC per 20 products): equal to
    the plain field dot product and to the oracle's restatement, including the headroom worst case (every operand r - 1) and lengths
    that leave partial blocks of products (crates/jolt-field/src/bn254/mont.rs:630-735 tests the same contract)."""
    c = ffi.Context(0)
    for n in (1, 19, 20, 21, 1000, 65537):
        a, b = rand_fr(n, 900 + n), rand_fr(n, 901 + n)
        ta, tb = c.upload(a), c.upload(b)
        want = O.fr_mul(a, b)
        acc = np.zeros((1, 4), dtype=np.uint64)
        for k in range(0, n, 4096):
            chunk = want[k:k + 4096]
            while chunk.shape[0] > 1:
                if chunk.shape[0] % 2:
                    chunk = np.concatenate([chunk, np.zeros((1, 4), dtype=np.uint64)])
                chunk = O.fr_add(chunk[0::2], chunk[1::2])
            acc = O.fr_add(acc, chunk)
        assert np.array_equal(c.table_dot(ta, tb, deferred=True), acc[0]), n
        assert np.array_equal(c.table_dot(ta, tb, deferred=False), acc[0]), n
    top = O.to_mont([O.R_MOD - 1] * 4096)  # the largest products: the accumulator's documented headroom
    t = c.upload(top)
    one = O.to_mont([1])[0]
    want = O.to_mont([4096 % O.R_MOD])[0]  # (r-1)^2 = 1 mod r, summed 4096 times
    assert np.array_equal(c.table_dot(t, t, deferred=True), want) and np.array_equal(one, O.fr_mul(top[:1], top[:1])[0])
    c.close()
