"""Oracle pin for the one-hot (Twist/Shout) restatement (oracle/onehot.c).  The reference holds no vectors for LazyFoldedRa;
its parity statement (crates/jolt-kernels/src/optimized/lazy_ra.rs:26-32) is that every gathered value equals the iterated
dense `lo + r*(hi - lo)` bind of the materialised selector column -- checked here against the oracle's own bind, for every
branch width, with cold cycles, plus the pushforward identity sum_k G[k]*table[k] == <w, column>."""
import numpy as np

import oracle_lib as O
from util import rand_challenge, rand_fr


def test_lazy_gather_equals_dense_bind_chain_and_pushforward_identity():
    rng = np.random.default_rng(7)
    for K, n_vars in ((16, 7), (5, 4), (255, 9)):
        T = 1 << n_vars
        col = rng.integers(0, K, size=T, dtype=np.uint8)
        col[rng.random(T) < 0.3] = 0xFF  # cold cycles (RAM columns)
        table = rand_fr(K, 100 + K)
        dense = np.zeros((T, 4), dtype=np.uint64)
        hot = col != 0xFF
        dense[hot] = table[col[hot]]
        assert np.array_equal(O.onehot_values(table, 1, K, col, T), dense)
        branch, width = table, 1
        for b in range(min(4, n_vars)):
            c = rand_challenge(200 + b) if b % 2 == 0 else rand_fr(1, 300 + b)[0]  # both challenge shapes
            branch = O.onehot_double_branches(branch, c)
            width *= 2
            dense = O.bind_low_to_high(dense, c)
            assert np.array_equal(O.onehot_values(branch, width, K, col, T), dense), (K, b)
        w = rand_fr(T, 400)
        G = O.onehot_pushforward(col, K, w)
        col_dense = np.zeros((T, 4), dtype=np.uint64)
        col_dense[hot] = table[col[hot]]
        def total(x):
            acc = np.zeros((1, 4), dtype=np.uint64)
            for row in x:
                acc = O.fr_add(acc, row.reshape(1, 4))
            return acc
        assert np.array_equal(total(O.fr_mul(G, table)), total(O.fr_mul(w, col_dense)))


def test_booleanity_address_phase_equals_the_dense_joint_member():
    """stage 6a (optimized/booleanity.rs:283-427): the K-domain round loop over the pushforward masses, against the oracle's dense naive member over the
    joint (cycle || address) tables -- index j * K + k, bound low-to-high, so the log K address variables come first -- with the summand
    eq(r_addr, k) eq(r_cycle, j) sum_i g^(2i) (ra_i^2 - ra_i); the one-hot columns make the squared-weight bind exact"""
    import oracle_lib as O
    from util import rand_challenge, rand_fr
    log_k, log_t, n_polys = 3, 5, 4
    K, T = 1 << log_k, 1 << log_t
    rng = np.random.default_rng(41)
    cols = rng.integers(0, K, size=(n_polys, T)).astype(np.uint8)
    cols[1, rng.random(T) < 0.4] = 0xFF
    r_cyc, r_adr, gamma = rand_fr(log_t, 42), rand_fr(log_k, 43), rand_fr(1, 44)[0]
    eq_cyc, eq_adr = O.eq_evals(r_cyc), O.eq_evals(r_adr)
    masses = np.stack([O.onehot_pushforward(cols[i], K, eq_cyc) for i in range(n_polys)])
    ker = O.BooleanityAddress(masses, gamma, r_adr)
    mul = lambda a, b: O.fr_mul(np.asarray(a).reshape(1, 4), np.asarray(b).reshape(1, 4))[0]
    one = O.to_mont([1])[0]
    # dense joint tables, index j * K + k
    eq_joint = O.fr_mul(np.repeat(eq_cyc, K, axis=0), np.tile(eq_adr, (T, 1)))
    ras = []
    for i in range(n_polys):
        g = np.zeros((T, K), dtype=np.uint64)
        hot = cols[i] != 0xFF
        g[np.nonzero(hot)[0], cols[i][hot]] = 1
        ras.append(O.fr_from_u64(g.reshape(-1)))
    terms, w = [], one
    g2 = mul(gamma, gamma)
    for i in range(n_polys):
        terms += [(w, [0, 1 + i, 1 + i]), (O.fr_neg(w.reshape(1, 4))[0], [0, 1 + i])]
        w = mul(w, g2)
    dense = O.Member.expr([eq_joint] + ras, terms, 3)
    claim = dense.input_claim()
    assert not np.any(claim)  # ra^2 = ra on 0/1 grids: the input claim is exactly zero
    bind = None
    for rnd in range(log_k):
        want = dense.prove_round(bind, claim)
        if bind is not None:
            ker.bind(bind)
        evals = ker.round()
        assert np.array_equal(O.fr_add(evals[0].reshape(1, 4), evals[1].reshape(1, 4))[0], claim)
        assert np.array_equal(O.univariate_from_evals(evals), want), rnd
        bind = rand_challenge(50 + rnd, shifted=(rnd != 1))
        claim = O.univariate_evaluate(want, bind)
    ker.bind(bind)
    assert np.array_equal(ker.intermediate(), claim)
