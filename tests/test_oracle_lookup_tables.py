"""Pins of oracle/lookup_tables.c (CPU only).

* materialize_entry of all 42 tables against the Python model of what each table means (tests/lookup_table_fixture.py);
* evaluate_mle on Boolean points == materialize_entry -- the reference's mle_random_test (tables/test_utils.rs:64-84), with the reference's index shapes;
* the address rounds computed from the definition are a sumcheck of the first-principles input claim (instruction_read_raf.rs:1545-1575):
  s_0(0) + s_0(1) = claim, s_{i+1}(0) + s_{i+1}(1) = s_i(r_i), quadratic in every round."""
import numpy as np
import pytest

import oracle_lib as O
from lookup_table_fixture import TABLES, all_table_rows, challenge, fixture_rows, random_index, shaped_index, table_model
from util import rand_fr

R = O.R_MOD


def boolean_point(index):
    bits = [(index >> (127 - i)) & 1 for i in range(128)]
    return O.to_mont(bits)


def test_table_ids_follow_the_enum():
    assert O.table_count() == len(TABLES) == 42
    assert O.TABLE_KINDS == TABLES


@pytest.mark.parametrize("kind", range(42))
def test_materialize_entry_and_boolean_mle(kind):
    name = TABLES[kind]
    rng = np.random.default_rng(100 + kind)
    edge = [0, (1 << 128) - 1, 1, 2, 3, 4, 1 << 127, 1 << 126, 1 << 64, (1 << 64) - 1, ((1 << 64) - 1) << 64, 5 << 61]
    cases = [shaped_index(name, rng) for _ in range(40)] + [random_index(name, rng) for _ in range(40)]
    if name not in ("VirtualSRL", "VirtualSRA", "VirtualROTR", "VirtualROTRW"):
        cases += edge
    for index in cases:
        entry = O.table_materialize_entry(kind, index)
        zero_mask = name == "VirtualSRA" and index & 0x5555555555555555_5555555555555555 == 0
        if not zero_mask:  # a shift by 64 is outside the instruction's range; the recurrence gives 2^64 - 2 there
            assert entry == table_model(name, index), (name, hex(index))
        got = O.from_mont(O.table_evaluate_mle(kind, boolean_point(index)))[0]
        assert got == entry, (name, hex(index))


def interpolate_quadratic(e0, e1, e2, x):
    """value at x of the degree-2 polynomial through (0, e0), (1, e1), (2, e2)"""
    inv2 = pow(2, -1, R)
    c = e0
    a = (e2 - 2 * e1 + e0) * inv2 % R
    b = (e1 - e0 - a) % R
    return (a * x * x + b * x + c) % R


def run_rounds(idx, tab, raf, log_t, gamma_int, canonical):
    u = O.eq_evals(O.to_mont([1000 + 37 * i for i in range(log_t)]))
    gamma = O.to_mont([gamma_int])[0]
    ch = O.to_mont([challenge(i) for i in range(128)])
    evals, tv, ops = O.read_raf_address_rounds(idx, tab, raf, u, gamma, ch, canonical=canonical)
    claim = O.from_mont(O.read_raf_input_claim(idx, tab, raf, u, gamma, canonical=canonical))[0]
    return evals, tv, ops, claim, u


@pytest.mark.parametrize("canonical", [False, True])
def test_direct_address_rounds_are_a_sumcheck_of_the_input_claim(canonical):
    log_t = 4
    idx, tab, raf = fixture_rows(log_t, 12345)  # parity_default_geometry's rows
    evals, tv, ops, claim, u = run_rounds(idx, tab, raf, log_t, 0xACE157EF, canonical)
    for i in range(128):
        e0, e1, e2 = O.from_mont(evals[i])
        assert (e0 + e1) % R == claim, i
        claim = interpolate_quadratic(e0, e1, e2, challenge(i))
    # what is left after the address rounds is sum_j eq(r, j) * eq(r_address, k_j) * F_j(r_address): check it from the table values
    u_int = O.from_mont(u)
    r_int = [challenge(i) for i in range(128)]
    tv_int, (left, right, identity, upper) = O.from_mont(tv), O.from_mont(ops)
    g = 0xACE157EF
    total = 0
    for j in range(1 << log_t):
        k = int(idx[j, 0]) | (int(idx[j, 1]) << 64)
        w = u_int[j]
        for i in range(128):
            w = w * (r_int[i] if (k >> (127 - i)) & 1 else (1 - r_int[i])) % R
        f = tv_int[tab[j]] if tab[j] != 0xFF else 0
        f += (g * g * identity + (g * g * g * upper if canonical else 0)) if raf[j] else (g * left + g * g * right)
        total += w * f
    assert total % R == claim


def test_direct_address_rounds_all_tables():
    log_t = 7
    idx, tab, raf = all_table_rows(log_t, 7)
    assert set(int(t) for t in tab if t != 0xFF) == set(range(42))
    evals, tv, ops, claim, _ = run_rounds(idx, tab, raf, log_t, 0xBEEF, False)
    for i in range(128):
        e0, e1, e2 = O.from_mont(evals[i])
        assert (e0 + e1) % R == claim, i
        claim = interpolate_quadratic(e0, e1, e2, challenge(i))


def test_mle_is_multilinear_in_each_variable():
    """evaluate_mle at a random point is the multilinear interpolation of its values at the two Boolean settings of any one variable"""
    rng = np.random.default_rng(9)
    for kind in range(42):
        point = rand_fr(128, 300 + kind)
        p_int = O.from_mont(point)
        for var in rng.choice(128, size=6, replace=False):
            lo, hi = point.copy(), point.copy()
            lo[var] = O.to_mont([0])[0]
            hi[var] = O.to_mont([1])[0]
            v = O.from_mont(O.table_evaluate_mle(kind, point))[0]
            v0 = O.from_mont(O.table_evaluate_mle(kind, lo))[0]
            v1 = O.from_mont(O.table_evaluate_mle(kind, hi))[0]
            assert v == (v0 + p_int[var] * (v1 - v0)) % R, (TABLES[kind], int(var))


def test_reference_known_answers_of_the_bit_layer():
    """The concrete values the reference's own unit tests hold for this path (crates/jolt-lookup-tables/src/interleave.rs:66-112, lookup_bits.rs:181-246):
    interleave / uninterleave on every pair and value they list, LookupBits masking, split, trailing zeros / leading ones -- against the oracle's bit layer, the
    product's (suffix_mle.hip.h built for the host) and the fixture's Python integers."""
    import ctypes as C
    from jolt_amd import ffi
    from lookup_table_fixture import interleave, operands

    def oracle_operands(v):
        x, y = C.c_uint64(), C.c_uint64()
        O.lib().orc_uninterleave(C.c_uint64(v & (2**64 - 1)), C.c_uint64(v >> 64), C.byref(x), C.byref(y))
        return x.value, y.value

    M = 2**64 - 1
    assert interleave(0b01, 0b10) == 0b0110 and interleave(1, 0) == 0b10 and interleave(0, 1) == 0b01  # roundtrip_small, single_bit_positions
    for x, y in [(0, 0), (M, M), (M, 0), (0, M), (0xDEADBEEFCAFEBABE, 0x123456789ABCDEF0), (1, 1), (1 << 63, 1 << 63), (0xDEAD, 0xBEEF)]:
        v = interleave(x, y)
        assert oracle_operands(v) == (x, y) == operands(v)
        assert O.suffix_mle(5, v, 128) == y and ffi.host_suffix_mle(5, v, 128) == y                      # Suffixes::RightOperand reads y off the index
        assert O.suffix_mle(1, v, 128) == x & y and ffi.host_suffix_mle(1, v, 128) == x & y
    for v in [0, 1, 2**128 - 1, 0xAAAABBBBCCCCDDDD1111222233334444]:                                     # uninterleave_interleave_roundtrip
        assert interleave(*oracle_operands(v)) == v
    # LookupBits::new(0xFF, 4) keeps 0x0F (new_masks_excess_bits): LowerWord of the 4-bit suffix
    assert O.suffix_mle(10, 0xFF, 4) == 0x0F and ffi.host_suffix_mle(10, 0xFF, 4) == 0x0F
    # split_roundtrip: 0b1101_0110 splits into 0b1101 | 0b0110 -- the chunk / suffix cut every phase makes
    assert (0b11010110 >> 4, O.suffix_mle(10, 0b11010110, 4)) == (0b1101, 0b0110)
    # trailing_zeros_and_leading_ones: 0b1110_1000 of length 8 has 3 of each.  As operands: y = 0b1110_1000 spread onto the even positions of a 16-bit suffix:
    v = interleave(0b10110101, 0b11101000)
    for lib_suffix in (O.suffix_mle, ffi.host_suffix_mle):
        assert lib_suffix(24, v, 16) == 1 << 3              # RightShiftHelper = 2^leading_ones(y)
        assert lib_suffix(23, v, 16) == 0b10110101 >> 3     # RightShift = x >> trailing_zeros(y)
