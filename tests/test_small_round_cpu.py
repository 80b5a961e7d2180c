"""The per-pair evaluation of the integer round kernel (jolt_amd/csrc/small_round.hip.h: small_pair_eval + the descriptor analysis of jolt_member_create_lc_small)
compiled for the HOST, against a Python big-integer model of the summand: sum_g prod_{f in g} (const_f + sum_k c_k * table_k(t)) with table_k(t) = lo + t (hi - lo).
Same code as the device kernel except for the field multiply (host_mul64 instead of the 29-bit-limb product; both are pinned to the oracle in test_abi_cpu.py).
Covers: integer groups of one and two columns (exact integer products, negative values at t >= 2, coefficients on either factor), integer entries inside field-valued
linear combinations (deferred reduction), constants, mixed groups, the corner values 0 / 1 / 2^64 - 1, every evaluation-point layout the members use."""
import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import ffi
from util import rand_fr

R_MOD = O.R_MOD


def mont(v):
    return O.to_mont([v % R_MOD])[0]


def model(groups_int, is_int, int_pairs, fr_vals, points):
    """groups_int: [[(const, [(coeff, table)])]] over Python ints; table values: integer columns from int_pairs, field tables from fr_vals (canonical ints)"""
    out = []
    for t in points:
        total = 0
        for g in groups_int:
            prod = 1
            for const, entries in g:
                v = const
                for c, ti in entries:
                    lo, hi = (int(int_pairs[ti][0]), int(int_pairs[ti][1])) if is_int[ti] else fr_vals[ti]
                    v += c * (lo + t * (hi - lo))
                prod = prod * v % R_MOD
            total = (total + prod) % R_MOD
        out.append(total)
    return out


def run_case(rng, n_tables, groups_int, is_int, int_pairs, n_evals, skip):
    fr_tabs = rand_fr(2 * n_tables, int(rng.integers(1 << 30))).reshape(n_tables, 2, 4)
    fr_canon = O.from_mont(fr_tabs.reshape(-1, 4))
    fr_vals = [(int(fr_canon[2 * i]), int(fr_canon[2 * i + 1])) for i in range(n_tables)]
    groups = [[(None if const == 0 else mont(const), [(mont(c), ti) for c, ti in entries]) for const, entries in g] for g in groups_int]
    got = ffi.host_small_round_pair(n_tables, groups, is_int, int_pairs, fr_tabs, n_evals, skip)
    points = [0] + [(s + 1 if skip else s) for s in range(1, n_evals)]
    want = model(groups_int, is_int, int_pairs, fr_vals, points)
    assert [int(x) for x in O.from_mont(got)] == want


CORNERS = [0, 1, 2, 2**32 - 1, 2**32, 2**63, 2**64 - 1]


@pytest.mark.parametrize("n_evals,skip", [(1, False), (1, True), (2, True), (3, True), (2, False), (3, False), (4, False), (4, True)])
def test_integer_groups_and_mixed_linear_combinations(n_evals, skip):
    rng = np.random.default_rng(100 * n_evals + int(skip))
    for trial in range(40):
        n_tables = 8
        is_int = np.array([1, 1, 1, 1, 1, 0, 0, 1], dtype=np.uint8)
        if trial % 5 == 0:
            ints = np.array(CORNERS, dtype=np.uint64)[rng.integers(0, len(CORNERS), size=(n_tables, 2))]
        else:
            ints = rng.integers(0, 2**64, size=(n_tables, 2), dtype=np.uint64)
        big = lambda: int(rng.integers(1, 2**62)) * int(rng.integers(1, 2**62)) * int(rng.integers(1, 2**62)) * int(rng.integers(1, 2**60)) % R_MOD
        groups = [
            [(0, [(1, 0)]), (0, [(1, 1)])],                       # integer group, two columns, unit coefficients (flag x value)
            [(0, [(big(), 2)]), (0, [(1, 3)])],                   # coefficient on the first factor
            [(0, [(1, 4)]), (0, [(big(), 7)])],                   # ... on the second
            [(0, [(big(), 0)])],                                  # one column, one coefficient
            [(0, [(1, 5)]), (0, [(1, 0), (big(), 1), (big(), 2), (1, 3)])],  # field table x integer linear combination (deferred reduction)
            [(0, [(big(), 6)]), (1, [(R_MOD - 1, 4)])],           # gamma * eq-like table x (1 - column): a constant and a negative coefficient
            [(5, [(1, 5), (big(), 7)]), (0, [(1, 6)])],           # field and integer entries in one factor
        ]
        k = int(rng.integers(1, len(groups) + 1))
        chosen = [groups[i] for i in rng.permutation(len(groups))[:k]]
        run_case(rng, n_tables, chosen, is_int, ints, n_evals, skip)


def test_three_factor_groups_stay_field_valued():
    """a product of three integer columns is not an integer group (two factors at most): every factor goes through the accumulator + REDC path"""
    rng = np.random.default_rng(7)
    is_int = np.ones(3, dtype=np.uint8)
    for _ in range(20):
        ints = rng.integers(0, 2**64, size=(3, 2), dtype=np.uint64)
        run_case(rng, 3, [[(0, [(1, 0)]), (0, [(1, 1)]), (0, [(1, 2)])]], is_int, ints, 4, False)


def test_bind_to_field_formula():
    """what k_bind_ints_to_field computes per output -- REDC((1 - r) R^2 * lo + r R^2 * hi) -- is the oracle's bind_to_field (dense.rs:129-142): checked through the
    integer linear combination path with coefficients (1 - r) and r at the point t = 0 over the columns (lo, hi)"""
    rng = np.random.default_rng(9)
    for _ in range(20):
        lo, hi = int(rng.integers(0, 2**64, dtype=np.uint64)), int(rng.integers(0, 2**64, dtype=np.uint64))
        r = int(rng.integers(1, 2**62)) * int(rng.integers(1, 2**62)) * int(rng.integers(1, 2**62)) % R_MOD
        ints = np.array([[lo, 0], [hi, 0]], dtype=np.uint64)
        groups = [[(None, [(mont(1 - r), 0), (mont(r), 1)])]]
        got = ffi.host_small_round_pair(2, groups, np.ones(2, dtype=np.uint8), ints, np.zeros((2, 2, 4), dtype=np.uint64), 1, False)
        want = O.bind_to_field_u64(np.array([lo, hi], dtype=np.uint64), mont(r))
        assert np.array_equal(got[0], np.asarray(want).reshape(-1, 4)[0])
