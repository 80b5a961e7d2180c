"""GPU parity of the Spartan outer (stage 1) T-scale sums through the C ABI: the uni-skip extended-node sums,
the materialised Az / Bz of the remainder member, the remainder rounds through the split-eq product member, and the post-hoc
evaluation of all inputs -- against the oracle's restatement of reference/spartan_outer.rs (tests/test_oracle_r1cs.py pins that
restatement to the reference's row loops and dense member)."""
import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import ffi
from r1cs_fixture import column_weights, make_system
from util import rand_challenge, rand_fr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = ffi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("n_rows,n_inputs,log_t,n_nodes", [(19, 35, 6, 9), (5, 3, 3, 2), (19, 35, 12, 9), (3, 1, 1, 1)])
def test_uniskip_sums_and_materialised_tables_match_oracle(ctx, n_rows, n_inputs, log_t, n_nodes):
    T = 1 << log_t
    a_rows, b_rows = make_system(n_rows, n_inputs, 60 + n_rows)
    inputs = [O.fr_from_u64(np.random.default_rng(70 + v).integers(0, 2**64, size=T, dtype=np.uint64)) if v % 2 else rand_fr(T, 70 + v) for v in range(n_inputs)]
    dev_inputs = [ctx.upload(z) for z in inputs]
    eq = O.eq_evals(rand_fr(log_t + 1, 80))
    row_w = rand_fr(n_nodes * 2 * n_rows, 81).reshape(n_nodes, 2, n_rows, 4)
    wa, wb = column_weights(a_rows, row_w, n_inputs, O), column_weights(b_rows, row_w, n_inputs, O)
    got = ctx.r1cs_uniskip_sums(dev_inputs, ctx.upload(eq), wa, wb)
    assert np.array_equal(got, O.r1cs_uniskip_sums(inputs, eq, wa, wb))
    if log_t <= 6:  # the reference's row loop itself
        assert np.array_equal(got, O.r1cs_uniskip_sums_rows(O.r1cs_row_values(inputs, a_rows), O.r1cs_row_values(inputs, b_rows), eq, row_w))
    az, bz = ctx.r1cs_materialize(dev_inputs, wa[0], wb[0])
    want_az, want_bz = O.r1cs_materialize(inputs, wa[0], wb[0])
    assert np.array_equal(az.download(), want_az) and np.array_equal(bz.download(), want_bz)
    point = rand_fr(log_t, 82)
    vals = ctx.tables_evaluate(dev_inputs, point)
    for v in range(0, n_inputs, max(1, n_inputs // 5)):
        assert np.array_equal(vals[v], O.poly_evaluate(inputs[v], point))


def test_outer_remainder_rounds_through_the_split_eq_member(ctx):
    """The whole remainder member on the device: materialise Az / Bz, then all log_t + 1 rounds of eq(tau_low, .) * kernel * Az * Bz
    through jolt_member_create_split_eq_product, lock step with the oracle's dense member over the same tables."""
    log_t, n_inputs = 7, 6
    T = 1 << log_t
    inputs = [rand_fr(T, 90 + v) for v in range(n_inputs)]
    tau_low, kernel = rand_fr(log_t + 1, 91), rand_fr(1, 92)[0]
    wa, wb = rand_fr(2 * (1 + n_inputs), 93).reshape(2, 1 + n_inputs, 4), rand_fr(2 * (1 + n_inputs), 94).reshape(2, 1 + n_inputs, 4)
    az, bz = ctx.r1cs_materialize([ctx.upload(z) for z in inputs], wa, wb)
    member = ctx.member_split_eq_product(az, bz, tau_low, scale=kernel)
    oaz, obz = O.r1cs_materialize(inputs, wa, wb)
    orc = O.Member.gruen_product(oaz, obz, tau_low, scale=kernel)
    claim = orc.input_claim()
    one = O.to_mont([1])[0]
    got = ctx.prove_batch([member], [claim], [one], [0], log_t + 1, 3, label=4)
    want = O.prove_batch([orc], [claim], [one], [0], log_t + 1, 3, label=4)
    for k in ("polys", "challenges", "final_claim"):
        assert np.array_equal(got[k], want[k]), k


def test_r1cs_argument_checks(ctx):
    z = [ctx.upload(rand_fr(8, 1)), ctx.upload(rand_fr(16, 2))]
    w = rand_fr(2 * 3, 3)
    with pytest.raises(ffi.JoltError) as e:
        ctx.r1cs_materialize(z, w, w)
    assert e.value.status == 5  # inputs of different length
    with pytest.raises(ffi.JoltError) as e:
        ctx.r1cs_uniskip_sums(z[:1], ctx.upload(rand_fr(8, 4)), rand_fr(4, 5), rand_fr(4, 6))  # eq must have 2 * cycles entries
    assert e.value.status == 5
