"""Pins of oracle/read_raf.c: the 48 suffix polynomials against the Python big-integer model of the same Rust source
(tests/read_raf_fixture.py), the 0/1-valued classification, and the phase scans against brute-force sums over the model."""
import numpy as np
import pytest

import oracle_lib as O
from read_raf_fixture import KINDS, ZERO_ONE, interesting_bits, make_rows, suffix_lists, suffix_model, uninterleave
from util import rand_fr


@pytest.mark.parametrize("length", [0, 8, 16, 56, 64, 72, 120])
def test_suffix_polynomials_match_the_big_integer_model(length):
    assert O.NUM_SUFFIX_KINDS == len(KINDS)
    rng = np.random.default_rng(length)
    for bits in interesting_bits(rng, length):
        for kind in range(len(KINDS)):
            want = suffix_model(kind, bits, length)
            assert O.suffix_mle(kind, bits, length) == want, (KINDS[kind], hex(bits), length)
            if KINDS[kind] in ZERO_ONE:
                assert want in (0, 1)
            assert O.suffix_is_01_valued(kind) == (KINDS[kind] in ZERO_ONE)


def test_phase_scan_equals_brute_force_sums():
    T, n_tables, address_bits = 96, 5, 128
    idx, table, raf = make_rows(T, n_tables, 3)
    lists = suffix_lists(n_tables, 4)
    u = rand_fr(T, 5)
    u_int = O.from_mont(u)
    offs = np.concatenate([[0], np.cumsum([len(l) for l in lists])])
    for suffix_len in (120, 64, 56, 0):
        got_raf, got_suf = O.read_raf_phase_scan(idx, table, raf, n_tables, u, suffix_len, address_bits, lists, canonical=True)
        want_raf = [[0] * 256 for _ in range(6)]
        want_suf = [[0] * 256 for _ in range(int(offs[-1]))]
        upper = max(suffix_len - address_bits // 2, 0)
        for j in range(T):
            index = int(idx[j, 0]) | (int(idx[j, 1]) << 64)
            chunk = (index >> suffix_len) & 255
            bits = index % (1 << suffix_len)
            if raf[j]:
                want_raf[4][chunk] += u_int[j]
                want_raf[2][chunk] += u_int[j] * bits
                if upper == 0 or (bits >> (suffix_len - upper)) == (1 << upper) - 1:
                    want_raf[5][chunk] += u_int[j]
            else:
                x, _, y, _ = uninterleave(bits, suffix_len)
                want_raf[3][chunk] += u_int[j]
                want_raf[0][chunk] += u_int[j] * x
                want_raf[1][chunk] += u_int[j] * y
            if table[j] != 0xFF:
                for s, kind in enumerate(lists[table[j]]):
                    want_suf[int(offs[table[j]]) + s][chunk] += u_int[j] * suffix_model(kind, bits, suffix_len)
        for q in range(6):
            assert O.from_mont(got_raf[q]) == [v % O.R_MOD for v in want_raf[q]], (suffix_len, q)
        for s in range(int(offs[-1])):
            assert O.from_mont(got_suf[s]) == [v % O.R_MOD for v in want_suf[s]], (suffix_len, s)


def test_condense_and_cycle_tables_equal_their_definitions():
    T, n_tables, address_bits, phases, ra_count = 40, 4, 128, 16, 4
    idx, table, raf = make_rows(T, n_tables, 6)
    u, v = rand_fr(T, 7), rand_fr(256, 8)
    got = O.from_mont(O.read_raf_condense(idx, u, v, 48))
    u_int, v_int = O.from_mont(u), O.from_mont(v)
    for j in range(T):
        index = int(idx[j, 0]) | (int(idx[j, 1]) << 64)
        assert got[j] == u_int[j] * v_int[(index >> 48) & 255] % O.R_MOD
    tv, ri, rid, vt = rand_fr(n_tables, 9), rand_fr(1, 10)[0], rand_fr(1, 11)[0], rand_fr(phases * 256, 12).reshape(phases, 256, 4)
    combined, ra = O.read_raf_cycle_tables(idx, table, raf, tv, ri, rid, vt, address_bits, ra_count)
    tv_i, vt_i = O.from_mont(tv), [O.from_mont(vt[p]) for p in range(phases)]
    ri_i, rid_i = O.from_mont(ri.reshape(1, 4))[0], O.from_mont(rid.reshape(1, 4))[0]
    c_i = O.from_mont(combined)
    for j in range(T):
        index = int(idx[j, 0]) | (int(idx[j, 1]) << 64)
        assert c_i[j] == ((0 if table[j] == 0xFF else tv_i[table[j]]) + (rid_i if raf[j] else ri_i)) % O.R_MOD
        for i in range(ra_count):
            want = 1
            for p in range(i * 4, i * 4 + 4):
                want = want * vt_i[p][(index >> (address_bits - 8 * (p + 1))) & 255] % O.R_MOD
            assert O.from_mont(ra[i][j: j + 1])[0] == want


@pytest.mark.parametrize("length", [0, 8, 56, 64, 72, 120])
def test_device_suffix_code_built_for_the_host_matches_oracle_and_model(length):
    """suffix_mle.hip.h (what the phase-scan kernel evaluates per row) compiled for the host: all 48 kinds against the C oracle and the
    Python big-integer model on the corner / random suffixes"""
    from jolt_amd import ffi
    rng = np.random.default_rng(100 + length)
    for bits in interesting_bits(rng, length):
        for kind in range(len(KINDS)):
            got = ffi.host_suffix_mle(kind, bits, length)
            assert got == O.suffix_mle(kind, bits, length) == suffix_model(kind, bits, length), (KINDS[kind], hex(bits), length)
