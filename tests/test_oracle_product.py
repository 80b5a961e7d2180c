"""Pins of the oracle's restatement of the Spartan product-virtualization formulas (oracle/r1cs.c, spartan_product.rs): the integer
extension coefficients are the Lagrange basis of the window {-1, 0, 1} at the nodes {-2 .. 2} (the reference's own unit test
extension_coefficients_match_field_lagrange, :941-956), in-window nodes select one lane pair, and the remainder tables at a window node
reproduce that lane."""
import numpy as np

import oracle_lib as O
from product_fixture import make_rows
from util import rand_fr


def lagrange_at(node, i, domain=(-1, 0, 1)):
    num, den = 1, 1
    for j, xj in enumerate(domain):
        if j != i:
            num *= node - xj
            den *= domain[i] - xj
    assert num % den == 0
    return num // den


def test_extension_coefficients_are_the_integer_lagrange_basis():
    c = O.spartan_product_extension_coefficients()
    for p, node in enumerate(range(-2, 3)):
        assert [int(x) for x in c[p]] == [lagrange_at(node, i) for i in range(3)]
    assert [list(map(int, c[p])) for p in (1, 2, 3)] == [[1, 0, 0], [0, 1, 0], [0, 0, 1]]  # 0/1 selectors inside the window


def test_t1_and_tables_against_a_python_big_integer_model():
    T = 16
    rows = make_rows(T, 5)
    eq = O.eq_evals(rand_fr(4, 6))
    eq_int = O.from_mont(eq)
    left_lanes = [[int(x) for x in rows["left_input"]], [int(x) for x in rows["lookup_output"]], [int(x) for x in rows["jump"]]]
    right_lanes = [rows["_right_python"], [int(x) for x in rows["branch"]], [1 - int(x) for x in rows["next_is_noop"]]]
    got = O.from_mont(O.spartan_product_t1(rows, eq))
    for p, node in enumerate(range(-2, 3)):
        c = [lagrange_at(node, i) for i in range(3)]
        want = 0
        for j in range(T):
            left = sum(c[i] * left_lanes[i][j] for i in range(3))   # exact integers, as the reference computes them (S128 x S192 -> S256)
            right = sum(c[i] * right_lanes[i][j] for i in range(3))
            assert abs(left) < 2**67 and abs(right) < 2**130 and abs(left * right) < 2**197
            want = (want + eq_int[j] * left * right) % O.R_MOD
        assert got[p] == want, node
    w = rand_fr(3, 7)
    w_int = O.from_mont(w)
    left, right = O.spartan_product_tables(rows, w)
    for j in range(T):
        assert O.from_mont(left[j: j + 1])[0] == sum(w_int[i] * left_lanes[i][j] for i in range(3)) % O.R_MOD
        assert O.from_mont(right[j: j + 1])[0] == sum(w_int[i] * right_lanes[i][j] for i in range(3)) % O.R_MOD
