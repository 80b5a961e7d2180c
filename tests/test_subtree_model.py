"""CPU check of the subtree sharding's specification (tests/subtree_model.py): every sharded step equals the same step on the global
polynomial, for 1 to 8 ranks and polynomials down to the size of the crown."""
import random

import pytest

import subtree_model as M


@pytest.mark.parametrize("gamma,ell", [(0, 5), (1, 4), (2, 3), (2, 6), (3, 4), (3, 7), (1, 1), (2, 2)])
def test_ownership_is_a_partition_with_nested_prefixes(gamma, ell):
    G, n = 1 << gamma, 1 << ell
    seen = {}
    for g in range(G):
        m = M.owned(n, g, gamma)
        idx = [M.insert(c, g, gamma) for c in range(m)]
        assert idx == sorted(idx) and all(i < n for i in idx)
        assert M.insert(m, g, gamma) >= n
        for c, i in enumerate(idx):
            assert i not in seen
            seen[i] = (g, c)
            assert M.slot_of(i, gamma) == (g, c)
        for k in range(n + 1):  # every prefix of the indices is a prefix of the compact order
            assert M.owned(k, g, gamma) == sum(1 for i in idx if i < k)
    assert sorted(seen) == list(range(n))
    if ell > gamma:
        assert all(M.owned(n, g, gamma) == n >> gamma for g in range(G))


@pytest.mark.parametrize("gamma,ell", [(1, 4), (2, 5), (3, 6), (2, 3), (3, 4), (0, 4)])
def test_sharded_fold_evaluation_and_witness_equal_the_global_ones(gamma, ell):
    rng = random.Random(1000 * gamma + ell)
    G, n = 1 << gamma, 1 << ell
    poly = [rng.randrange(M.R) for _ in range(n)]
    point = [rng.randrange(M.R) for _ in range(ell)]
    want_levels = M.fold_levels_global(poly, point)
    got = M.fold_levels_sharded([M.to_compact(poly, g, gamma) for g in range(G)], point, gamma)
    for k, lvl in enumerate(want_levels):
        for g in range(G):
            assert got[g][k] == M.to_compact(lvl, g, gamma), (k, g)
    u = rng.randrange(M.R)
    for k, lvl in enumerate(want_levels):
        want = sum(c * pow(u, i, M.R) for i, c in enumerate(lvl)) % M.R
        assert M.evaluate_sharded([got[g][k] for g in range(G)], u, gamma) == want, k
    q = rng.randrange(M.R)
    b = [sum(pow(q, k, M.R) * lvl[i] for k, lvl in enumerate(want_levels) if i < len(lvl)) % M.R for i in range(n)]
    b_compacts = []
    for g in range(G):  # the RLC is slot-wise on the compact arrays
        m = len(got[g][0])
        b_compacts.append([sum(pow(q, k, M.R) * got[g][k][c] for k in range(ell) if c < len(got[g][k])) % M.R for c in range(m)])
        assert b_compacts[g] == M.to_compact(b, g, gamma)
    want_h = M.witness_global(b, u)
    got_h = M.witness_sharded(b_compacts, u, gamma, n)
    for g in range(G):
        assert got_h[g] == M.to_compact(want_h, g, gamma), g
