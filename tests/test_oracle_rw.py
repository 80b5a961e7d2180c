"""The oracle's sparse read-write matrix (oracle/rw_matrix.c) against the oracle's dense naive member over the materialised K x T
grids: the reference pins its optimized RAM read/write-checking kernel exactly this way (lock-step equality with the reference
member, crates/jolt-kernels/src/optimized/parity.rs:79-118) -- it holds no vectors for this path."""
import numpy as np
import pytest

import oracle_lib as O
from rw_fixture import dense_grids, make_trace
from util import rand_challenge, rand_fr


def run_sparse_vs_dense(tr, seed, sparse_factory):
    """Drives `sparse_factory(tr, inc_table)` (an object with the RwMatrix / device interface) in lock step with the dense member."""
    log_k, log_t = tr["log_k"], tr["log_t"]
    K, T = 1 << log_k, 1 << log_t
    ra, val, inc, val_init = dense_grids(tr, O)
    tau = rand_fr(log_t, seed)
    gamma = rand_fr(1, seed + 1)[0]
    one = O.to_mont([1])[0]
    eq_t = O.eq_evals(tau)
    tile = lambda t: np.tile(t, (K, 1))
    g1 = O.fr_add(one.reshape(1, 4), gamma.reshape(1, 4))[0]
    # eq * ra * (val + gamma (val + inc)) = (1 + gamma) eq ra val + gamma eq ra inc, index = k * T + j, LowToHigh
    dense = O.Member.expr([tile(eq_t), ra, val, tile(inc)], [(g1, [0, 1, 2]), (gamma, [0, 1, 3])], 3)
    claim = dense.input_claim()
    sparse = sparse_factory(tr, inc)
    eq_state = O.SplitEqState(tau)
    inc_cur, vi_cur = inc.copy(), val_init.copy()
    bind = None
    for rnd in range(log_t + log_k):
        want = dense.prove_round(bind, claim)
        if bind is not None:  # ingest(bind, rnd - 1): ram_read_write.rs:108-143
            if rnd - 1 < log_t:
                sparse.cycle_bind(bind)
                eq_state.bind(bind)
                inc_cur = O.bind_low_to_high(inc_cur, bind)
                if rnd - 1 == log_t - 1:
                    sparse.into_address_major()
            else:
                vi_cur = sparse.address_bind(bind, vi_cur)
        if rnd < log_t:
            e_out, e_in, in_bits = eq_state.tables()
            q = sparse.cycle_round(e_out, e_in, in_bits, inc_cur, gamma)
            got = O.gruen_poly_deg_3(eq_state.scalar, eq_state.point(), q[0], q[1], claim)
        else:
            s = sparse.address_round(vi_cur, inc_cur, eq_state.scalar.reshape(1, 4), gamma)
            s1 = O.fr_sub(claim.reshape(1, 4), s[0].reshape(1, 4))[0]  # UnivariatePoly::from_evals_and_hint: s(1) from the claim
            got = np.zeros((4, 4), dtype=np.uint64)
            got[:3] = O.univariate_from_evals(np.stack([s[0], s1, s[1]]))
        assert np.array_equal(got, want), f"round {rnd}"
        bind = rand_challenge(seed + 10 + rnd)
        claim = O.univariate_evaluate(want, bind)
    dense.finish_rounds(bind)
    vi_cur = sparse.address_bind(bind, vi_cur) if log_k else vi_cur
    if log_k == 0:
        sparse.cycle_bind(bind)
        sparse.into_address_major()
    fin = dense.final_values()  # eq, ra, val, inc
    ra_f, val_f = sparse.final_values(vi_cur)
    assert np.array_equal(ra_f, fin[1]) and np.array_equal(val_f, fin[2])
    return sparse


@pytest.mark.parametrize("log_k,log_t,access,hot", [(3, 4, 0.7, None), (4, 6, 1.0, None), (2, 5, 0.3, None), (5, 5, 0.9, 3), (3, 3, 0.0, None), (1, 1, 1.0, None)])
def test_sparse_matrix_matches_dense_member(log_k, log_t, access, hot):
    tr = make_trace(log_k, log_t, 100 + log_k * 7 + log_t, access=access, hot=hot)
    run_sparse_vs_dense(tr, 900 + log_t, lambda t, inc: O.RwMatrix(t["addresses"], t["pre"], t["post"]))
