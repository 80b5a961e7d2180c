"""oracle/registers_rw.c (the restatement of the optimized registers read/write-checking kernel) against the oracle's dense naive member
over the materialised (K x T) grids -- the reference's own pin for this kernel (optimized-vs-reference lock step, parity.rs:79-118):
every round polynomial of the log T cycle rounds and the log K address rounds, the final claim, and the two operand claims."""
import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import stages as S
from registers_fixture import dense_grids, inc_table
from util import rand_challenge, rand_fr


def _m(a, b): return O.fr_mul(np.asarray(a).reshape(1, 4), np.asarray(b).reshape(1, 4))[0]
def _a(a, b): return O.fr_add(np.asarray(a).reshape(1, 4), np.asarray(b).reshape(1, 4))[0]


@pytest.mark.parametrize("log_k,log_t,hot,probs", [(3, 4, None, (0.8, 0.6, 0.7)), (7, 6, None, (0.8, 0.6, 0.7)), (2, 5, 2, (1.0, 1.0, 1.0)), (4, 3, None, (0.2, 0.1, 0.3)),
                                                    (5, 7, 3, (0.9, 0.9, 0.9)), (1, 1, None, (1.0, 1.0, 1.0)), (3, 5, None, (0.0, 0.0, 0.0))])
def test_sparse_registers_kernel_equals_the_dense_member(log_k, log_t, hot, probs):
    rng = np.random.default_rng(100 + 7 * log_k + log_t)
    tr = S.consistent_register_trace(log_k, log_t, rng, *probs, hot=hot)
    K, T = 1 << log_k, 1 << log_t
    r_cycle, gamma = rand_fr(log_t, 300 + log_t), rand_fr(1, 301)[0]
    inc = inc_table(tr, O)
    rs1g, rs2g, wag, valg = dense_grids(tr, O)
    one = O.to_mont([1])[0]
    g2 = _m(gamma, gamma)
    eq_t = np.tile(O.eq_evals(r_cycle), (K, 1))
    inc_t = np.tile(inc, (K, 1))
    # tables: 0 eq, 1 rd_wa, 2 rd_inc, 3 val, 4 rs1_ra, 5 rs2_ra;   eq * (wa * (inc + val) + g * rs1 * val + g^2 * rs2 * val)
    dense = O.Member.expr([eq_t, wag, inc_t, valg, rs1g, rs2g], [(one, [0, 1, 2]), (one, [0, 1, 3]), (gamma, [0, 4, 3]), (g2, [0, 5, 3])], 3)
    claim = dense.input_claim()
    m = O.RegMatrix(tr["rs1"], tr["rs1_val"], tr["rs2"], tr["rs2_val"], tr["rd"], tr["rd_pre"], tr["rd_post"], gamma)
    eq_state = O.SplitEqState(r_cycle)
    inc_cur = inc.copy()
    bind, dense_state, chal = None, None, []
    for rnd in range(log_t + log_k):
        want = dense.prove_round(bind, claim)
        if bind is not None:
            if rnd - 1 < log_t:
                m.cycle_bind(bind)
                eq_state.bind(bind)
                inc_cur = O.bind_low_to_high(inc_cur, bind)
                if rnd == log_t:
                    dense_state = list(m.into_dense(K))
            else:
                dense_state = [O.bind_low_to_high(t, bind) for t in dense_state]
        if rnd < log_t:
            e_out, e_in, _ = eq_state.tables()
            q = m.cycle_round(e_out, e_in, inc_cur)
            got = O.gruen_poly_deg_3(eq_state.scalar, eq_state.point(), q[0], q[1], claim)
        else:
            evals = O.regrw_address_round(*dense_state, inc_cur[0], eq_state.scalar)
            assert np.array_equal(_a(evals[0], evals[1]), claim), rnd  # the round check the reference keeps (mod.rs:241-248)
            got = O.univariate_from_evals(evals)
        assert np.array_equal(got, want), f"round {rnd}"
        bind = rand_challenge(400 + rnd, shifted=(rnd % 3 != 1))
        chal.append(bind)
        claim = O.univariate_evaluate(want, bind)
    # final bind and the output claims (mod.rs:386-402)
    if log_k:
        dense_state = [O.bind_low_to_high(t, bind) for t in dense_state]
    else:
        m.cycle_bind(bind); eq_state.bind(bind); inc_cur = O.bind_low_to_high(inc_cur, bind); dense_state = list(m.into_dense(1))
    ra_f, wa_f, val_f = (t[0] for t in dense_state)
    assert np.array_equal(claim, _m(eq_state.scalar, _a(_m(wa_f, _a(inc_cur[0], val_f)), _m(ra_f, val_f))))
    # the bound point, big-endian: (r_address, r_cycle) = the reversed challenge halves (mod.rs:279-293)
    r_cyc_pt = np.stack(chal[:log_t][::-1])
    r_adr_pt = np.stack(chal[log_t:][::-1]) if log_k else np.zeros((0, 4), dtype=np.uint64)
    point = np.concatenate([r_adr_pt, r_cyc_pt])
    eq_adr, eq_cyc = (O.eq_evals(r_adr_pt) if log_k else O.to_mont([1])), O.eq_evals(r_cyc_pt)
    assert np.array_equal(O.regrw_operand_claim(tr["rs1"], eq_adr, eq_cyc), O.poly_evaluate(rs1g, point))
    assert np.array_equal(O.regrw_operand_claim(tr["rs2"], eq_adr, eq_cyc), O.poly_evaluate(rs2g, point))
    assert np.array_equal(wa_f, O.poly_evaluate(wag, point)) and np.array_equal(val_f, O.poly_evaluate(valg, point))
    assert np.array_equal(_a(_m(gamma, O.poly_evaluate(rs1g, point)), _m(g2, O.poly_evaluate(rs2g, point))), ra_f)
    m.close()


def test_cells_of_a_cycle_fold_and_sort():
    """rs2 == rs1 folds into one cell (ra = g + g^2), rd folds into a read's cell (wa = 1, next = post), cells sorted by register"""
    g = rand_fr(1, 5)[0]
    u8, u64 = (lambda *v: np.array(v, dtype=np.uint8)), (lambda *v: np.array(v, dtype=np.uint64))
    m = O.RegMatrix(u8(5, 9, 0xFF), u64(11, 12, 0), u8(5, 3, 7), u64(11, 13, 14), u8(5, 0xFF, 2), u64(11, 0, 15), u64(99, 0, 16), g)
    e = m.export()
    assert list(e["rows"]) == [0, 1, 1, 2, 2] and list(e["cols"]) == [5, 3, 9, 2, 7]
    g2 = _m(g, g)
    one, zero = O.to_mont([1])[0], np.zeros(4, dtype=np.uint64)
    assert np.array_equal(e["ra"][0], _a(g, g2)) and np.array_equal(e["wa"][0], one) and e["prev"][0] == 11 and e["next"][0] == 99
    assert np.array_equal(e["ra"][1], g2) and np.array_equal(e["ra"][2], g) and np.array_equal(e["wa"][1], zero)
    assert np.array_equal(e["ra"][3], zero) and np.array_equal(e["wa"][3], one) and e["prev"][3] == 15 and e["next"][3] == 16
    assert np.array_equal(e["val"][3], O.to_mont([15])[0]) and np.array_equal(e["ra"][4], g2)
    m.close()
