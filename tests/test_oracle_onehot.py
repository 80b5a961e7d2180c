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
