"""Instruction read+RAF checking (stage 5) scans on the device through the C ABI: every phase's condensation, RAF
sums and per-table suffix accumulators (all 48 suffix kinds), the cycle-phase columns, and the cycle rounds through the split-eq
uniform member -- against oracle/read_raf.c (tests/test_oracle_read_raf.py pins that restatement to a big-integer model)."""
import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import ffi
from read_raf_fixture import KINDS, make_rows, suffix_lists
from util import rand_challenge, rand_fr

pytestmark = pytest.mark.gpu
ADDRESS_BITS, PHASES = 128, 16


@pytest.fixture(scope="module")
def ctx():
    c = ffi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("log_t,n_tables,canonical", [(12, 8, False), (9, 3, True), (0, 2, False)])
def test_all_sixteen_phases_match_oracle(ctx, log_t, n_tables, canonical):
    """the loop of OptimizedInstructionReadRafKernel::init_phase over the 16 phases: condense with the previous phase's eq table, then
    scan; the per-cycle mass u, the RAF sums and every (table, suffix) accumulator equal the oracle's after every phase"""
    T = 1 << log_t
    idx, table, raf = make_rows(T, n_tables, 50 + log_t)
    lists = suffix_lists(n_tables, 51)
    assert sorted({k for l in lists for k in l}) == list(range(len(KINDS)))  # every suffix kind is exercised
    rr = ctx.read_raf(idx, table, raf, n_tables)
    u_host = rand_fr(T, 52)
    u = ctx.upload(u_host)
    for phase in range(PHASES):
        suffix_len = ADDRESS_BITS - 8 * (phase + 1)
        if phase:
            v = rand_fr(256, 60 + phase)  # the bound-challenge eq table of the previous phase (any 256 field elements here)
            rr.condense(u, v, suffix_len + 8)
            u_host = O.read_raf_condense(idx, u_host, v, suffix_len + 8)
            assert np.array_equal(u.download(), u_host), phase
        got_raf, got_suf = rr.phase_scan(u, suffix_len, ADDRESS_BITS, lists, canonical=canonical)
        want_raf, want_suf = O.read_raf_phase_scan(idx, table, raf, n_tables, u_host, suffix_len, ADDRESS_BITS, lists, canonical=canonical)
        assert np.array_equal(got_raf, want_raf), phase
        for t in range(n_tables):
            base = sum(len(l) for l in lists[:t])
            for s, kind in enumerate(lists[t]):
                assert np.array_equal(got_suf[base + s], want_suf[base + s]), (phase, t, KINDS[kind])
    rr.free()


def test_skewed_rows_one_table_one_chunk(ctx):
    """every row in the same (table, chunk) bin -- one wavefront sums 2^11 rows -- and a table nobody uses"""
    T, n_tables = 1 << 11, 3
    rng = np.random.default_rng(70)
    idx = np.zeros((T, 2), dtype=np.uint64)
    idx[:, 0] = rng.integers(0, 2**48, size=T, dtype=np.uint64)
    idx[:, 1] = np.uint64(0xAB) << np.uint64(56)  # the top chunk of every row is 0xAB
    table = np.full(T, 1, dtype=np.uint8)
    raf = (rng.random(T) < 0.5).astype(np.uint8)
    lists = [[0, 1], [0, 5, 12, 23, 44], [3]]
    u_host = rand_fr(T, 71)
    rr = ctx.read_raf(idx, table, raf, n_tables)
    got_raf, got_suf = rr.phase_scan(ctx.upload(u_host), 120, ADDRESS_BITS, lists)
    want_raf, want_suf = O.read_raf_phase_scan(idx, table, raf, n_tables, u_host, 120, ADDRESS_BITS, lists)
    assert np.array_equal(got_raf, want_raf) and np.array_equal(got_suf, want_suf)
    assert not np.any(got_suf[0]) and not np.any(got_suf[7])  # tables 0 and 2 are absent
    rr.free()


def test_cycle_columns_and_cycle_rounds(ctx):
    """pending_combined_base / pending_ra_base for every cycle, then eq(r_reduction, j) * combined(j) * prod_i ra_i(j) over the log T
    cycle rounds through the eq-weighted product member, lock step with the oracle's dense member"""
    log_t, n_tables, ra_count = 9, 6, 4
    T = 1 << log_t
    idx, table, raf = make_rows(T, n_tables, 80)
    rr = ctx.read_raf(idx, table, raf, n_tables)
    tv, ri, rid = rand_fr(n_tables, 81), rand_fr(1, 82)[0], rand_fr(1, 83)[0]
    vt = rand_fr(PHASES * 256, 84).reshape(PHASES, 256, 4)
    combined, ra = rr.cycle_tables(tv, ri, rid, vt, ADDRESS_BITS, ra_count)
    want_combined, want_ra = O.read_raf_cycle_tables(idx, table, raf, tv, ri, rid, vt, ADDRESS_BITS, ra_count)
    assert np.array_equal(combined.download(), want_combined)
    for i in range(ra_count):
        assert np.array_equal(ra[i].download(), want_ra[i]), i
    r_reduction = rand_fr(log_t, 85)
    one = O.to_mont([1])[0]
    # one product group of 1 + ra_count linear factors with the eq weight factored out (degree 2 + ra_count = 6 round messages)
    member = ctx.member_lc([combined] + ra, [[(None, [(one, i)]) for i in range(1 + ra_count)]], 1 + ra_count, eq_point=r_reduction)
    orc = O.Member.expr([O.eq_evals(r_reduction), want_combined] + [want_ra[i] for i in range(ra_count)], [(one, list(range(2 + ra_count)))], 2 + ra_count)
    claim = orc.input_claim()
    bind = None
    for rnd in range(log_t):
        want = orc.prove_round(bind, claim)
        evals, aux = member.prove_round(bind, want_aux=True)
        assert np.array_equal(ffi.host_gruen_poly_from_q(aux[0], aux[1], evals, claim), want), rnd
        bind = rand_challenge(90 + rnd)
        claim = O.univariate_evaluate(want, bind)
    rr.free()


def test_argument_checks(ctx):
    idx, table, raf = make_rows(16, 2, 95)
    with pytest.raises(ffi.JoltError):
        ctx.read_raf(idx, np.full(16, 2, dtype=np.uint8), raf, 2)  # table index out of range
    rr = ctx.read_raf(idx, table, raf, 2)
    u = ctx.upload(rand_fr(16, 96))
    with pytest.raises(ffi.JoltError) as e:
        rr.phase_scan(u, 120, ADDRESS_BITS, [[0], [200]])  # unknown suffix kind
    assert e.value.status == 6  # JOLT_ERR_UNSUPPORTED
    with pytest.raises(ffi.JoltError):
        rr.phase_scan(u, 124, ADDRESS_BITS, [[0], [1]])  # the chunk above the suffix must fit the address
    with pytest.raises(ffi.JoltError) as e:
        rr.phase_scan(ctx.upload(rand_fr(8, 97)), 120, ADDRESS_BITS, [[0], [1]])
    assert e.value.status == 5
    rr.free()


@pytest.mark.parametrize("log_t,seed,canonical,all_tables", [(4, 12345, False, False), (3, 67890, True, False), (10, 31, False, True)])
def test_address_rounds_with_device_scans_match_the_definition(ctx, log_t, seed, canonical, all_tables):
    """The kernel end to end on its address side: device condensation + scans over the 42 REAL tables feed the host state machine (jolt_host_read_raf_address_*),
    and every one of the 128 round polynomials equals the oracle's from-the-definition round (oracle/lookup_tables.c).  Rows, reduction point, gamma and
    challenges are the reference kernel's own parity recipe (instruction_read_raf.rs:1477-1522, 1579-1646); the third case has every table present.  Then the
    cycle columns built from the state machine's table values equal the oracle's built from evaluate_mle at r_address, and the running claim is their sum."""
    from lookup_table_fixture import all_table_rows, challenge, fixture_rows
    idx, tab, raf = all_table_rows(log_t, seed) if all_tables else fixture_rows(log_t, seed)
    T = 1 << log_t
    lists = ffi.lookup_suffix_lists()
    r_reduction = O.to_mont([1000 + 37 * i for i in range(log_t)])
    u_host = O.eq_evals(r_reduction)
    gamma = O.to_mont([0xACE157EF])[0]
    challenges = O.to_mont([challenge(i) for i in range(128)])
    present = np.zeros(42, dtype=np.uint8)
    present[[int(t) for t in set(tab.tolist()) if t != 0xFF]] = 1
    rr = ctx.read_raf(idx, tab, raf, 42)
    u = ctx.upload(u_host)
    state = ffi.HostReadRafAddress(gamma, present, canonical)
    direct = O.ReadRafAddressDirect(idx, tab, raf, u_host, gamma, canonical=canonical)
    claim = O.read_raf_input_claim(idx, tab, raf, u_host, gamma, canonical=canonical)
    v_tables = []
    for phase in range(PHASES):
        suffix_len = ADDRESS_BITS - 8 * (phase + 1)
        if phase:
            rr.condense(u, v_tables[-1], suffix_len + 8)
        raf_sums, suffix_sums = rr.phase_scan(u, suffix_len, ADDRESS_BITS, lists, canonical=canonical)
        state.init_phase(phase, raf_sums, suffix_sums)
        if phase == 0:
            first = state.message()  # s(1) summed from the tables: s(0) + s(1) is the first-principles input claim
            assert np.array_equal(ffi.host_fr_add(first[0], first[1]), claim)
        for rnd in range(8):
            i = 8 * phase + rnd
            got = state.message(claim)
            assert np.array_equal(got, direct.round()), i
            coeffs = O.univariate_from_evals(got)
            claim = O.univariate_evaluate(coeffs, challenges[i])
            direct.bind(challenges[i])
            assert state.bind(challenges[i]) == (rnd == 7)
        v_tables.append(state.v_table(phase))
        assert np.array_equal(v_tables[-1], O.eq_evals(challenges[8 * phase: 8 * phase + 8]))
    table_values, raf_interleaved, raf_identity = state.finish()
    want_values, (left, right, identity, upper) = direct.values()
    used = np.flatnonzero(present)
    assert np.array_equal(table_values[used], want_values[used])
    g2 = O.fr_mul(gamma.reshape(1, 4), gamma.reshape(1, 4))
    assert np.array_equal(raf_interleaved, O.fr_add(O.fr_mul(gamma.reshape(1, 4), left.reshape(1, 4)), O.fr_mul(g2, right.reshape(1, 4)))[0])
    want_identity = O.fr_mul(g2, identity.reshape(1, 4))
    if canonical:
        want_identity = O.fr_add(want_identity, O.fr_mul(O.fr_mul(g2, gamma.reshape(1, 4)), upper.reshape(1, 4)))
    assert np.array_equal(raf_identity, want_identity[0])
    ra_count = 8 if log_t != 3 else 4
    vt = np.stack(v_tables)
    combined, ra = rr.cycle_tables(table_values, raf_interleaved, raf_identity, vt, ADDRESS_BITS, ra_count)
    want_combined, want_ra = O.read_raf_cycle_tables(idx, tab, raf, want_values, raf_interleaved, raf_identity, vt, ADDRESS_BITS, ra_count)
    assert np.array_equal(combined.download(), want_combined)
    orc = O.Member.expr([O.eq_evals(r_reduction), want_combined] + [want_ra[i] for i in range(ra_count)], [(O.to_mont([1])[0], list(range(2 + ra_count)))], 2 + ra_count)
    assert np.array_equal(orc.input_claim(), claim)  # what 128 address rounds leave is what the log T cycle rounds sum
    for i in range(ra_count):
        assert np.array_equal(ra[i].download(), want_ra[i])
        ra[i].free()
    combined.free()
    u.free()
    state.close()
    rr.free()
