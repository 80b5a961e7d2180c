"""Spartan outer (stage 1) T-scale sums in the oracle: the reference's row-weight loops (reference/spartan_outer.rs:172-221,318-349)
against the column-weight form the device kernels take, and the materialised Az / Bz against the reference's dense remainder member
(:236-300: TauKernel * (AzConst + sum_v AzWeight_v z_v) * (BzConst + sum_u BzWeight_u z_u) over the joint (cycle || stream) domain)."""
import numpy as np
import pytest

import oracle_lib as O
from r1cs_fixture import column_weights, make_system
from util import rand_challenge, rand_fr


@pytest.mark.parametrize("n_rows,n_inputs,log_t", [(5, 3, 3), (19, 8, 4), (3, 1, 1)])
def test_column_weight_form_equals_the_row_loops(n_rows, n_inputs, log_t):
    T = 1 << log_t
    a_rows, b_rows = make_system(n_rows, n_inputs, 10 + n_rows)
    inputs = [rand_fr(T, 20 + v) for v in range(n_inputs)]
    eq = O.eq_evals(rand_fr(log_t + 1, 30))
    n_nodes = 4
    row_w = rand_fr(n_nodes * 2 * n_rows, 31).reshape(n_nodes, 2, n_rows, 4)
    az_rows, bz_rows = O.r1cs_row_values(inputs, a_rows), O.r1cs_row_values(inputs, b_rows)
    want = O.r1cs_uniskip_sums_rows(az_rows, bz_rows, eq, row_w)
    wa, wb = column_weights(a_rows, row_w, n_inputs, O), column_weights(b_rows, row_w, n_inputs, O)
    assert np.array_equal(O.r1cs_uniskip_sums(inputs, eq, wa, wb), want)
    # materialised linear forms at (node 0): Az[(t << 1) | s] = sum_r w[s][r] * az_rows[r][t]
    az, bz = O.r1cs_materialize(inputs, wa[0], wb[0])
    for s in range(2):
        acc_a = np.zeros((T, 4), dtype=np.uint64)
        for r in range(n_rows):
            acc_a = O.fr_add(acc_a, O.fr_mul(az_rows[r], np.repeat(row_w[0, s, r].reshape(1, 4), T, axis=0)))
        assert np.array_equal(az[s::2], acc_a)


def test_materialised_tables_reproduce_the_dense_remainder_member():
    """eq * Az * Bz over the materialised tables (what the device hands to its split-eq product member) proves the same rounds as the
    reference's flat member over leaf tables: TauKernel, replicated inputs, stream-paired weight tables."""
    log_t, n_inputs = 3, 2
    T = 1 << log_t
    inputs = [rand_fr(T, 40 + v) for v in range(n_inputs)]
    tau_low = rand_fr(log_t + 1, 41)
    kernel = rand_fr(1, 42)[0]
    wa, wb = rand_fr(2 * (1 + n_inputs), 43).reshape(2, 1 + n_inputs, 4), rand_fr(2 * (1 + n_inputs), 44).reshape(2, 1 + n_inputs, 4)
    az, bz = O.r1cs_materialize(inputs, wa, wb)
    tau_kernel = O.eq_evals(tau_low, kernel)
    one = O.to_mont([1])[0]
    product = O.Member.expr([tau_kernel, az, bz], [(one, [0, 1, 2])], 3)
    # reference leaves: TauKernel (0), z_v replicated over the stream LSB (1..), AzWeight_v / BzWeight_v and the constants as stream pairs
    rep = lambda t: np.repeat(t, 2, axis=0)                     # replicate_stream_lsb: out[(t << 1) | s] = base[t]
    pair = lambda w0, w1: np.tile(np.stack([w0, w1]), (T, 1))   # stream_pair_lsb
    tables = [tau_kernel] + [rep(z) for z in inputs]
    aw = [len(tables) + v for v in range(n_inputs)]
    tables += [pair(wa[0, 1 + v], wa[1, 1 + v]) for v in range(n_inputs)]
    bw = [len(tables) + v for v in range(n_inputs)]
    tables += [pair(wb[0, 1 + v], wb[1, 1 + v]) for v in range(n_inputs)]
    ac, bc = len(tables), len(tables) + 1
    tables += [pair(wa[0, 0], wa[1, 0]), pair(wb[0, 0], wb[1, 0])]
    terms = [(one, [0, ac, bc])]
    for v in range(n_inputs):
        terms.append((one, [0, aw[v], 1 + v, bc]))
        terms.append((one, [0, ac, bw[v], 1 + v]))
        for u in range(n_inputs):
            terms.append((one, [0, aw[v], 1 + v, bw[u], 1 + u]))
    flat = O.Member.expr(tables, terms, 3)
    claim = product.input_claim()
    assert np.array_equal(claim, flat.input_claim())
    bind = None
    for rnd in range(log_t + 1):
        a, b = product.prove_round(bind, claim), flat.prove_round(bind, claim)
        assert np.array_equal(a, b), rnd
        bind = rand_challenge(50 + rnd)
        claim = O.univariate_evaluate(a, bind)
