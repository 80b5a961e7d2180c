"""Small-scalar (integer witness column) forms of the Spartan outer sums through the C ABI (SURVEY.md 8 a2 + 8f row 3):
FrSmallScalarAccumulator-style deferred reduction on the device (small_scalar.hip.h) and the exact integer arithmetic of the uni-skip
extension.  Pinned three ways: against the oracle's field-arithmetic restatement of the reference's row loops on the PROMOTED columns
(exact algebra: integer weights' field images), against the device's own field-arithmetic operators, and -- for the accumulator
itself -- against the oracle's restatement of FrSmallScalarAccumulator (tests/test_oracle_field.py pins that to plain sums)."""
import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import ffi
from util import rand_fr

pytestmark = pytest.mark.gpu
I64_MIN, I64_MAX = -(2**63), 2**63 - 1


@pytest.fixture(scope="module")
def ctx():
    c = ffi.Context(0)
    yield c
    c.close()


def make_columns(T, n_inputs, seed, wide_only_for=()):
    """integer columns of every kind with their corner values; returns (device Ints, promoted field columns, python ints)"""
    rng = np.random.default_rng(seed)
    cols = []
    for v in range(n_inputs):
        kind = ("flag", "u64", "i64", "i128")[v % 4]
        if kind == "flag":
            vals = [int(x) for x in rng.integers(0, 2, size=T)]
            arr, k = np.array(vals, dtype=np.uint64), "u64"
        elif kind == "u64":
            vals = [int(x) for x in rng.integers(0, 2**64, size=T, dtype=np.uint64)]
            vals[: min(T, 3)] = [0, 2**64 - 1, 1][: min(T, 3)]
            arr, k = np.array(vals, dtype=np.uint64), "u64"
        elif kind == "i64":
            vals = [int(x) for x in rng.integers(I64_MIN, I64_MAX, size=T, dtype=np.int64)]
            vals[: min(T, 4)] = [I64_MIN, I64_MAX, -1, 0][: min(T, 4)]
            arr, k = np.array(vals, dtype=np.int64), "i64"
        else:
            vals = [int(rng.integers(0, 2**62)) * int(rng.integers(0, 2**62)) * (1 if rng.random() < 0.5 else -1) for _ in range(T)]
            vals[: min(T, 4)] = [-(2**127), 2**127 - 1, -1, 0][: min(T, 4)]
            arr, k = vals, "i128"
        cols.append((k, arr, vals))
    return cols


def promote(vals):
    return O.to_mont([v % O.R_MOD for v in vals])


@pytest.mark.parametrize("log_n,k", [(0, 1), (3, 4), (10, 9), (13, 35)])
def test_ints_evaluate_matches_oracle(ctx, log_n, k):
    n = 1 << log_n
    cols = make_columns(n, k, 100 + log_n)
    dev = [ctx.ints(arr, kind) for kind, arr, _ in cols]
    point = rand_fr(log_n, 101)
    got = ctx.ints_evaluate(dev, point)
    eq = O.eq_evals(point) if log_n else O.to_mont([1])
    for v, (kind, arr, vals) in enumerate(cols):
        assert np.array_equal(got[v], O.poly_evaluate(promote(vals), point) if log_n else promote(vals)[0]), (v, kind)
        if kind == "i64" and n <= 4:  # the accumulator restated (its five limbs hold only a handful of full-range terms)
            assert np.array_equal(got[v], O.small_scalar_accumulate(eq, arr)), v
    # equal to the field-arithmetic operator on the promoted tables
    tabs = [ctx.upload(promote(vals)) for _, _, vals in cols]
    assert np.array_equal(got, ctx.tables_evaluate(tabs, point))
    with pytest.raises(ffi.JoltError):
        ctx.ints_evaluate(dev, rand_fr(log_n + 1, 102))  # length is not 2^n


@pytest.mark.parametrize("n_inputs,log_t,n_nodes", [(35, 6, 9), (3, 2, 2), (35, 11, 9), (1, 0, 1)])
def test_uniskip_sums_small_equal_the_field_form(ctx, n_inputs, log_t, n_nodes):
    T = 1 << log_t
    rng = np.random.default_rng(200 + n_inputs + log_t)
    cols = make_columns(T, n_inputs, 201 + log_t)
    # integer column weights in the range of the Lagrange extension coefficients times small row coefficients; most are zero.
    # A side: only the flag / i64 columns carry weight (|Az| must stay a 128-bit integer), B side: every kind
    wa = np.zeros((n_nodes, 2, 1 + n_inputs), dtype=np.int64)
    wb = np.zeros((n_nodes, 2, 1 + n_inputs), dtype=np.int64)
    for node in range(n_nodes):
        for s in range(2):
            wa[node, s, 0] = rng.integers(-2**20, 2**20)
            wb[node, s, 0] = rng.integers(-2**40, 2**40)
            for v in range(n_inputs):
                if rng.random() < 0.4 and v % 4 in (0, 2):
                    wa[node, s, 1 + v] = rng.integers(-2**20, 2**20)
                if rng.random() < 0.4:
                    wb[node, s, 1 + v] = rng.integers(-2**40, 2**40)
    wa[0, 0, 0], wb[0, 1, 0] = 0, 0
    dev = [ctx.ints(arr, kind) for kind, arr, _ in cols]
    eq = O.eq_evals(rand_fr(log_t + 1, 202))
    dev_eq = ctx.upload(eq)
    got = ctx.r1cs_uniskip_sums_small(dev, dev_eq, wa, wb)
    fields = [promote(vals) for _, _, vals in cols]
    fwa = O.fr_from_i64(wa.reshape(-1)).reshape(n_nodes, 2, 1 + n_inputs, 4)
    fwb = O.fr_from_i64(wb.reshape(-1)).reshape(n_nodes, 2, 1 + n_inputs, 4)
    assert np.array_equal(got, O.r1cs_uniskip_sums(fields, eq, fwa, fwb))
    assert np.array_equal(got, ctx.r1cs_uniskip_sums([ctx.upload(f) for f in fields], dev_eq, fwa, fwb))
    bad = wa.copy()
    bad[0, 0, 0] = I64_MIN
    with pytest.raises(ffi.JoltError):
        ctx.r1cs_uniskip_sums_small(dev, dev_eq, bad, wb)


@pytest.mark.parametrize("n_inputs,log_t", [(35, 7), (4, 1), (35, 12)])
def test_materialize_small_equals_the_field_form(ctx, n_inputs, log_t):
    T = 1 << log_t
    rng = np.random.default_rng(300 + log_t)
    cols = make_columns(T, n_inputs, 301 + log_t)
    wa, wb = rand_fr(2 * (1 + n_inputs), 302).reshape(2, 1 + n_inputs, 4), rand_fr(2 * (1 + n_inputs), 303).reshape(2, 1 + n_inputs, 4)
    for w in (wa, wb):  # sparse weights, as the folded row weights are
        for s in range(2):
            for v in range(1 + n_inputs):
                if rng.random() < 0.5:
                    w[s, v] = 0
    wa[0, 1] = O.to_mont([1])[0]
    wb[1, 1] = O.to_mont([O.R_MOD - 1])[0]
    dev = [ctx.ints(arr, kind) for kind, arr, _ in cols]
    az, bz = ctx.r1cs_materialize_small(dev, wa, wb)
    fields = [promote(vals) for _, _, vals in cols]
    want_az, want_bz = O.r1cs_materialize(fields, wa, wb)
    assert np.array_equal(az.download(), want_az)
    assert np.array_equal(bz.download(), want_bz)


def test_device_accumulator_equals_the_reference_accumulator_on_register_sized_columns(ctx):
    """sum_t eq[t] * z[t] for 40-bit signed columns of 2^10 cycles: device (13-limb sums, REDC) == the oracle's restatement of
    FrSmallScalarAccumulator (5-limb sums, Barrett), value for value"""
    log_n = 10
    rng = np.random.default_rng(400)
    arrs = [rng.integers(-2**40, 2**40, size=1 << log_n, dtype=np.int64) for _ in range(5)]
    point = rand_fr(log_n, 401)
    got = ctx.ints_evaluate([ctx.ints(a) for a in arrs], point)
    eq = O.eq_evals(point)
    for v, a in enumerate(arrs):
        assert np.array_equal(got[v], O.small_scalar_accumulate(eq, a)), v
