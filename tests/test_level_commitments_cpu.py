"""The identity behind the opening's first level commitments (jolt_host_hyperkzg_open_grid over a jolt_grid_hint), on the oracle alone:
for the joint polynomial J = sum_p s_p [hot_p(j) = k] + [k = 0] sum_d c_d f_d[j] on the grid index k T + j, the commitment of its s-th LowToHigh fold is

    com(P_s) = sum_p s_p sum_c w_c S_p^(s,c) + com(the dense part folded s times),      w_c = prod_b (x_b if bit b of c else 1 - x_b),  x_b = point[ell - 1 - b]

with S_p^(s,c) the sum of the SRS bases at (hot_p(j) T + j) >> s over the cycles j = c mod 2^s.  Everything here is the oracle's: the folds are
hyperkzg_fold_polynomials, the commitments kzg_commit, the group operations g1_*; no device, no product code -- this pins the weights, the bit order of the classes and
the base index of the class sums that the device path (jolt_grid_hint_begin / jolt_grid_commit_onehot_classes + jolt_host_hyperkzg_open_grid) relies on."""
import numpy as np
import pytest

import oracle_lib as O
from util import rand_challenge, rand_fr


@pytest.mark.parametrize("log_t,log_k,levels,cold", [(3, 2, 2, 0.0), (4, 2, 3, 0.3), (3, 1, 3, 0.5)])
def test_level_commitments_are_linear_in_the_class_sums(log_t, log_k, levels, cold):
    T, K = 1 << log_t, 1 << log_k
    ell = log_t + log_k
    rng = np.random.default_rng(7 + log_t + levels)
    srs = O.srs_setup_from_secret(rand_fr(1, 3)[0], K * T)
    n_cols = 3
    hot = rng.integers(0, K, size=(n_cols, T))
    hot[rng.random((n_cols, T)) < cold] = -1  # cold cycles
    s_oh, c_d = rand_fr(n_cols, 11), rand_fr(2, 12)
    dense = [rand_fr(T, 13), O.fr_from_u64(rng.integers(0, 2**64, size=T, dtype=np.uint64))]
    one = O.to_mont([1])[0]
    mul = lambda a, b: O.fr_mul(np.asarray(a).reshape(1, 4), np.asarray(b).reshape(1, 4))[0]
    # the joint polynomial, from the definition
    joint = np.zeros((K * T, 4), dtype=np.uint64)
    for p in range(n_cols):
        for j in range(T):
            if hot[p, j] >= 0:
                i = hot[p, j] * T + j
                joint[i] = O.fr_add(joint[i:i + 1], s_oh[p].reshape(1, 4))[0]
    dense_row = np.zeros((T, 4), dtype=np.uint64)
    for d in range(2):
        dense_row = O.fr_add(dense_row, O.fr_mul(dense[d], np.repeat(c_d[d].reshape(1, 4), T, axis=0)))
    joint[:T] = O.fr_add(joint[:T], dense_row)
    point = np.stack([rand_challenge(20 + k) for k in range(ell)])
    folds = O.hyperkzg_fold_polynomials(joint, point)  # folds[s] = P_s
    for s in range(1, levels + 1):
        want = O.kzg_commit(folds[s], srs[: len(folds[s])])
        xs = [point[ell - 1 - b] for b in range(s)]
        acc = O.g1_identity()
        for c in range(1 << s):
            w = one
            for b in range(s):
                w = mul(w, xs[b] if (c >> b) & 1 else O.fr_sub(one.reshape(1, 4), xs[b].reshape(1, 4))[0])
            for p in range(n_cols):
                S = O.g1_identity()  # the class sum of column p: bases at (hot T + j) >> s over the cycles of class c
                for j in range(c, T, 1 << s):
                    if hot[p, j] >= 0:
                        S = O.g1_add(S, srs[(hot[p, j] * T + j) >> s])
                acc = O.g1_add(acc, O.g1_scalar_mul(S, mul(s_oh[p], w)))
        d = dense_row.copy()
        for b in range(s):  # LowToHigh bind: new[y] = lo + x (hi - lo)
            lo, hi = d[0::2], d[1::2]
            d = O.fr_add(lo, O.fr_mul(O.fr_sub(hi, lo), np.repeat(xs[b].reshape(1, 4), lo.shape[0], axis=0)))
        acc = O.g1_add(acc, O.kzg_commit(d, srs[: d.shape[0]]))
        assert O.g1_eq(acc, want), s
