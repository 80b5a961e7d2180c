"""Spartan product virtualization (stage 2) on the device through the one-stream integer operators:
uni-skip extended-node values off the typed lanes, the remainder's left / right tables, the remainder rounds through the split-eq
product member, and the claimed-input evaluations -- against oracle/r1cs.c's restatement of spartan_product.rs."""
import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import ffi
from product_fixture import device_columns, field_column_weights, integer_column_weights, make_rows
from util import rand_fr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = ffi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("log_t", [0, 3, 9, 14])
def test_uniskip_values_and_remainder_tables_match_oracle(ctx, log_t):
    T = 1 << log_t
    rows = make_rows(T, 20 + log_t)
    cols = device_columns(ctx, rows)
    tau_low = rand_fr(log_t, 21)
    eq = O.eq_evals(tau_low) if log_t else O.to_mont([1])
    a, b = integer_column_weights(O.spartan_product_extension_coefficients())
    got = ctx.r1cs_uniskip_sums_small(cols, ctx.upload(eq), a, b, streams=1)
    assert np.array_equal(got, O.spartan_product_t1(rows, eq))
    w = rand_fr(3, 22)  # centered_lagrange_evals(3, r0) in the reference: any three field weights here
    fa, fb = field_column_weights(w, O)
    left, right = ctx.r1cs_materialize_small(cols, fa, fb, streams=1)
    want_left, want_right = O.spartan_product_tables(rows, w)
    assert np.array_equal(left.download(), want_left) and np.array_equal(right.download(), want_right)
    if log_t:  # claimed inputs: every lane at r_cycle from one eq table
        point = rand_fr(log_t, 23)
        vals = ctx.ints_evaluate(cols, point)
        assert np.array_equal(vals[0], O.poly_evaluate(O.fr_from_u64(rows["left_input"]), point))
        assert np.array_equal(vals[3], O.poly_evaluate(O.to_mont([v % O.R_MOD for v in rows["_right_python"]]), point))
        assert np.array_equal(vals[5], O.poly_evaluate(O.fr_from_u64(rows["next_is_noop"].astype(np.uint64)), point))


def test_product_remainder_rounds_through_the_split_eq_member(ctx):
    """ProductRemainderKernel (:297-523): eq(tau_low, .) * kernel * left * right over log_t rounds, lock step with the oracle's member"""
    log_t = 8
    rows = make_rows(1 << log_t, 30)
    cols = device_columns(ctx, rows)
    w, tau_low, kernel = rand_fr(3, 31), rand_fr(log_t, 32), rand_fr(1, 33)[0]
    fa, fb = field_column_weights(w, O)
    left, right = ctx.r1cs_materialize_small(cols, fa, fb, streams=1)
    member = ctx.member_split_eq_product(left, right, tau_low, scale=kernel)
    oleft, oright = O.spartan_product_tables(rows, w)
    orc = O.Member.gruen_product(oleft, oright, tau_low, scale=kernel)
    claim = orc.input_claim()
    one = O.to_mont([1])[0]
    got = ctx.prove_batch([member], [claim], [one], [0], log_t, 3, label=5)
    want = O.prove_batch([orc], [claim], [one], [0], log_t, 3, label=5)
    for k in ("polys", "challenges", "final_claim"):
        assert np.array_equal(got[k], want[k]), k


def test_stream_count_is_checked(ctx):
    cols = device_columns(ctx, make_rows(8, 40))
    a, b = integer_column_weights(O.spartan_product_extension_coefficients())
    with pytest.raises(ffi.JoltError):
        ctx.r1cs_uniskip_sums_small(cols, ctx.upload(rand_fr(16, 41)), a, b, streams=1)  # eq must have streams * cycles entries
    with pytest.raises(ffi.JoltError) as e:
        ffi._ck(ffi.lib().jolt_r1cs_uniskip_sums_small(ctx.h, None, 0, None, 3, None, None, 0, None), "x", ctx)
    assert e.value.status == 1
