"""GPU parity for the one-hot (Twist/Shout) path through the C ABI: address-folded materialisation,
pushforward G tables, and the lazily bound RA-virtualization member (LazyFoldedRa) -- bit-exact against the oracle and against
the dense split-eq uniform member over the materialised columns (the reference's own parity statement, lazy_ra.rs:26-32)."""
import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import ffi
from util import rand_challenge, rand_fr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = ffi.Context(0)
    yield c
    c.close()


def make_columns(n_polys, T, K, seed, cold=0.0):
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, K, size=(n_polys, T), dtype=np.uint8)
    if cold:
        idx[rng.random((n_polys, T)) < cold] = 0xFF
    return idx


def dense_column(idx_row, table):
    out = np.zeros((idx_row.shape[0], 4), dtype=np.uint64)
    hot = idx_row != 0xFF
    out[hot] = table[idx_row[hot]]
    return out


@pytest.mark.parametrize("K,n_vars,cold", [(16, 10, 0.0), (16, 9, 0.4), (255, 12, 0.1), (3, 4, 0.5)])
def test_materialize_and_pushforward_match_oracle(ctx, K, n_vars, cold):
    T, N = 1 << n_vars, 5
    idx = make_columns(N, T, K, 10 + K, cold)
    src = ctx.onehot(idx, K)
    tables = rand_fr(N * K, 20 + K).reshape(N, K, 4)
    for p in (0, N - 1):
        got = src.materialize(p, ctx.upload(tables[p])).download()
        assert np.array_equal(got, dense_column(idx[p], tables[p]))
        assert np.array_equal(got, O.onehot_values(tables[p], 1, K, idx[p], T))
    w = rand_fr(T, 30 + K)
    G = src.pushforward(ctx.upload(w)).download().reshape(N, K, 4)
    for p in range(N):
        assert np.array_equal(G[p], O.onehot_pushforward(idx[p], K, w)), p
    if K < 255:
        with pytest.raises(ffi.JoltError):
            ctx.onehot(np.full((1, 16), K, dtype=np.uint8), K)  # hot index outside the scale table
    src.free()


@pytest.mark.parametrize("V,F,K,n_vars,cold", [(1, 2, 16, 6, 0.0), (2, 3, 16, 7, 0.3), (8, 4, 16, 9, 0.0), (1, 4, 255, 13, 0.2), (3, 2, 5, 4, 0.5)])
def test_lazy_ra_member_equals_dense_uniform_member_and_oracle(ctx, V, F, K, n_vars, cold):
    """Every round message, bind and final value of the lazily bound member equals (a) the oracle's flat Expr member over the
    dense eq table and the materialised selector columns and (b) the device's dense split-eq uniform member -- through the
    index-encoded rounds (widths 1, 2, 4, 8), the materialising fourth bind and the dense rounds after it; then again after
    a reset."""
    T, N = 1 << n_vars, V * F
    idx = make_columns(N, T, K, 50 + K + F, cold)
    tables = rand_fr(N * K, 60 + F).reshape(N, K, 4)
    w = rand_fr(n_vars, 70 + F)
    coeffs = rand_fr(V, 80 + V)
    coeffs[0] = O.to_mont([1])[0]
    scale = rand_fr(1, 90)[0] if V == 2 else None
    dense = [dense_column(idx[p], tables[p]) for p in range(N)]
    src = ctx.onehot(idx, K)
    lazy = ctx.member_lazy_ra_uniform(src, tables, V, F, coeffs, w, scale=scale)
    twin = ctx.member_split_eq_uniform([ctx.upload(t) for t in dense], V, F, coeffs, w, scale=scale)
    eq = O.eq_evals(w, scale)
    terms = [(coeffs[v], [0] + [1 + v * F + k for k in range(F)]) for v in range(V)]
    for rep in range(2):
        orc = O.Member.expr([eq] + dense, terms, F + 1)
        claim = orc.input_claim()
        bind = None
        for rnd in range(n_vars):
            want = orc.prove_round(bind, claim)
            evals, aux = lazy.prove_round(bind, want_aux=True)
            if rep == 0:
                t_evals, _ = twin.prove_round(bind, want_aux=True)
                assert np.array_equal(evals, t_evals), f"round {rnd} vs dense member"
            got = ffi.host_gruen_poly_from_q(aux[0], aux[1], evals, claim)
            assert np.array_equal(got, want), f"rep {rep} round {rnd}"
            bind = rand_challenge(100 + rnd) if rnd % 3 else rand_fr(1, 100 + rnd)[0]  # 125-bit and full-width challenges
            claim = O.univariate_evaluate(want, bind)
        orc.finish_rounds(bind)
        lazy.finish(bind)
        fv, ofv = lazy.final_values(), orc.final_values()
        assert np.array_equal(fv[:-1], ofv[1:]) and np.array_equal(fv[-1], ofv[0])
        if rep == 0:
            twin.finish(bind)
            assert np.array_equal(fv, twin.final_values())
        lazy.reset()


def test_lazy_ra_member_in_a_batch_with_dense_members(ctx):
    """prove_batch (grouped launches) over a lazy member next to an ordinary expression member: transcript equals the oracle's."""
    n_vars, V, F, K = 8, 2, 4, 16
    T, N = 1 << n_vars, V * F
    idx = make_columns(N, T, K, 501, 0.1)
    tables = rand_fr(N * K, 502).reshape(N, K, 4)
    w = rand_fr(n_vars, 503)
    coeffs = rand_fr(V, 504)
    dense = [dense_column(idx[p], tables[p]) for p in range(N)]
    a, b = rand_fr(T, 505), rand_fr(T, 506)
    one = O.to_mont([1])[0]
    src = ctx.onehot(idx, K)
    lazy = ctx.member_lazy_ra_uniform(src, tables, V, F, coeffs, w)
    flat = ctx.member_expr([ctx.upload(a), ctx.upload(b)], [(one, [0, 1])], 2)
    terms = [(coeffs[v], [0] + [1 + v * F + k for k in range(F)]) for v in range(V)]
    orcs = [O.Member.expr([O.eq_evals(w)] + dense, terms, F + 1), O.Member.expr([a, b], [(one, [0, 1])], 2)]
    claims = [o.input_claim() for o in orcs]
    cf = list(rand_fr(2, 507))
    want = O.prove_batch(orcs, claims, cf, [0, 0], n_vars, F + 1, label=9)
    got = ctx.prove_batch([lazy, flat], claims, cf, [0, 0], n_vars, F + 1, label=9)
    for k in ("polys", "challenges", "member_claims", "final_claim"):
        assert np.array_equal(got[k], want[k]), k


def test_row_major_uniform_kernels_match_oracle(monkeypatch):
    """The big-round form of the uniform members (one work item per pair, eq weight applied to the sum over the products,
    coefficients pre-scaled into the lazy member's scale tables) is selected by size; JOLT_UNIFORM_ROWS_PAIRS=2 forces it at
    test sizes for both the dense and the lazily bound member, each round checked against the oracle."""
    monkeypatch.setenv("JOLT_UNIFORM_ROWS_PAIRS", "2")
    c = ffi.Context(0)
    try:
        for V, F, n_vars in ((8, 4, 9), (3, 3, 6), (2, 2, 5)):
            T, N, K = 1 << n_vars, V * F, 16
            idx = make_columns(N, T, K, 900 + F, 0.2)
            tables = rand_fr(N * K, 901 + F).reshape(N, K, 4)
            w = rand_fr(n_vars, 902 + F)
            coeffs = rand_fr(V, 903 + V)
            dense = [dense_column(idx[p], tables[p]) for p in range(N)]
            src = c.onehot(idx, K)
            lazy = c.member_lazy_ra_uniform(src, tables, V, F, coeffs, w)
            twin = c.member_split_eq_uniform([c.upload(t) for t in dense], V, F, coeffs, w)
            terms = [(coeffs[v], [0] + [1 + v * F + k for k in range(F)]) for v in range(V)]
            orc = O.Member.expr([O.eq_evals(w)] + dense, terms, F + 1)
            claim = orc.input_claim()
            assert np.array_equal(lazy.input_claim(), claim) and np.array_equal(twin.input_claim(), claim)
            bind = None
            for rnd in range(n_vars):
                want = orc.prove_round(bind, claim)
                for dev in (lazy, twin):
                    evals, aux = dev.prove_round(bind, want_aux=True)
                    assert np.array_equal(ffi.host_gruen_poly_from_q(aux[0], aux[1], evals, claim), want), (V, F, rnd)
                bind = rand_challenge(950 + rnd)
                claim = O.univariate_evaluate(want, bind)
            orc.finish_rounds(bind)
            for dev in (lazy, twin):
                dev.finish(bind)
                fv, ofv = dev.final_values(), orc.final_values()
                assert np.array_equal(fv[:-1], ofv[1:]) and np.array_equal(fv[-1], ofv[0])
    finally:
        c.close()


@pytest.mark.parametrize("N,K,n_vars,cold", [(3, 16, 6, 0.0), (8, 16, 9, 0.3), (40, 16, 7, 0.1), (2, 255, 12, 0.2)])
def test_lazy_booleanity_member_matches_oracle(ctx, N, K, n_vars, cold):
    """Booleanity cycle phase eq(w,j) * sum_i (H_i^2 - rho_i H_i) over gamma-pre-scaled, lazily bound selector columns: each round's
    cubic (gruen_poly_deg_3 of the two device sums) equals the oracle's flat Expr member over the dense eq table and the
    materialised columns; final values are the bound H_i.  Also inside prove_batch, and again after a reset."""
    T = 1 << n_vars
    idx = make_columns(N, T, K, 700 + N, cold)
    gamma = rand_fr(1, 701)[0]
    one = O.to_mont([1])[0]
    rho = [one]
    for _ in range(N - 1):
        rho.append(O.fr_mul(rho[-1].reshape(1, 4), gamma.reshape(1, 4))[0])
    eq_address = rand_fr(K, 702)
    tables = np.stack([O.fr_mul(eq_address, np.repeat(r.reshape(1, 4), K, axis=0)) for r in rho])  # rho_i * eq_address
    w = rand_fr(n_vars, 703)
    scale = rand_fr(1, 704)[0]
    dense = [dense_column(idx[p], tables[p]) for p in range(N)]
    neg = lambda x: O.fr_neg(np.asarray(x).reshape(1, 4))[0]
    # flat terms: eq*H_i*H_i - rho_i * eq*H_i
    terms = []
    for i in range(N):
        terms.append((one, [0, 1 + i, 1 + i]))
        terms.append((neg(rho[i]), [0, 1 + i]))
    src = ctx.onehot(idx, K)
    dev = ctx.member_lazy_booleanity(src, tables, rho, w, scale=scale)
    eq = O.eq_evals(w, scale)
    for rep in range(2):
        orc = O.Member.expr([eq] + dense, terms, 3)
        claim = orc.input_claim()
        bind = None
        for rnd in range(n_vars):
            want = orc.prove_round(bind, claim)
            evals, aux = dev.prove_round(bind, want_aux=True)
            got = ffi.host_gruen_poly_deg_3(aux[0], aux[1], evals[0], evals[1], claim)
            assert np.array_equal(got, want), f"rep {rep} round {rnd}"
            bind = rand_challenge(710 + rnd) if rnd % 2 else rand_fr(1, 710 + rnd)[0]
            claim = O.univariate_evaluate(want, bind)
        orc.finish_rounds(bind)
        dev.finish(bind)
        fv, ofv = dev.final_values(), orc.final_values()
        assert np.array_equal(fv[:-1], ofv[1:]) and np.array_equal(fv[-1], ofv[0])
        dev.reset()
    # in a batch (grouped launches, message assembly on the host mirror)
    orc = O.Member.expr([eq] + dense, terms, 3)
    claim = orc.input_claim()
    cf = [rand_fr(1, 720)[0]]
    want = O.prove_batch([orc], [claim], cf, [0], n_vars, 3, label=4)
    got = ctx.prove_batch([dev], [claim], cf, [0], n_vars, 3, label=4)
    for k in ("polys", "challenges", "member_claims", "final_claim"):
        assert np.array_equal(got[k], want[k]), k


def test_packed_witness_rows_expand_to_tables_and_hot_indices(ctx):
    """One upload of packed per-cycle records (the CommittedColumnsWitness shape: rd_inc i64, ram_inc i64, lookup_index u128,
    bytecode_pc u32, ram_address Option<u32>) expanded on the device: integer fields promote to the same Fr tables as
    jolt_table_from_u64 / _i64, the address fields become the hot-index columns of RaChunkSelector (chunk i of `chunks` 4-bit
    chunks = (value >> (chunks-1-i)*4) & 15), cold where the Option is None; a lazy member over them matches one over explicit
    indices."""
    T = 1 << 10
    rng = np.random.default_rng(77)
    dt = np.dtype([("rd_inc", "<i8"), ("ram_inc", "<i8"), ("lookup_lo", "<u8"), ("lookup_hi", "<u8"), ("pc", "<u4"), ("ram_addr", "<u4"),
                   ("ram_valid", "u1"), ("pad", "u1", 7)])
    rows = np.zeros(T, dtype=dt)
    rows["rd_inc"] = rng.integers(-2**62, 2**62, size=T)
    rows["ram_inc"] = rng.integers(-2**40, 2**40, size=T)
    rows["lookup_lo"] = rng.integers(0, 2**64, size=T, dtype=np.uint64)
    rows["lookup_hi"] = rng.integers(0, 2**64, size=T, dtype=np.uint64)
    rows["pc"] = rng.integers(0, 2**20, size=T)
    rows["ram_addr"] = rng.integers(0, 2**16, size=T)
    rows["ram_valid"] = rng.random(T) < 0.6
    R = ffi.Rows(ctx, rows)
    assert R.row_bytes == dt.itemsize
    assert np.array_equal(R.table(dt.fields["rd_inc"][1], 8, signed=True).download(), O.fr_from_i64(rows["rd_inc"]))
    assert np.array_equal(R.table(dt.fields["ram_inc"][1], 8, signed=True).download(), O.fr_from_i64(rows["ram_inc"]))
    assert np.array_equal(R.table(dt.fields["pc"][1], 4).download(), O.fr_from_u64(rows["pc"].astype(np.uint64)))
    assert np.array_equal(R.table(dt.fields["ram_valid"][1], 1).download(), O.fr_from_u64(rows["ram_valid"].astype(np.uint64)))
    # every integer field in ONE pass over the rows (jolt_ints_from_rows_many: whole rows staged in LDS) -- the same columns as field by field, on a ragged row count too
    fields = [(dt.fields["rd_inc"][1], 8, True), (dt.fields["ram_inc"][1], 8, True), (dt.fields["pc"][1], 4, False), (dt.fields["ram_valid"][1], 1, False),
              (dt.fields["lookup_hi"][1], 8, False), (dt.fields["ram_addr"][1], 2, False)]
    for n_rows in (T, 777):
        Rn = R if n_rows == T else ffi.Rows(ctx, rows[:n_rows])
        many = Rn.ints_many(fields)
        for (off, width, signed), col in zip(fields, many):
            one = Rn.ints(off, width, signed=signed)
            assert col.count == n_rows and np.array_equal(ctx.table_from_ints(col).download(), ctx.table_from_ints(one).download()), (n_rows, off, width)
            assert np.array_equal(ctx.table_from_ints(col).download(), Rn.table(off, width, signed=signed).download())
            one.free()
            col.free()
    with pytest.raises(ffi.JoltError):
        R.ints_many([(dt.itemsize - 4, 8, False)])  # field beyond the row
    # instruction RA: 32 chunks of 4 bits of the 128-bit lookup index, most significant chunk first
    chunks, bits = 32, 4    # instruction RA: 32 chunks of 4 bits of the 128-bit lookup index, most significant chunk first
    chunks, bits = 32, 4
    shifts = [(chunks - 1 - i) * bits for i in range(chunks)]
    src = R.onehot(dt.fields["lookup_lo"][1], 16, shifts, bits)
    got = src.download()
    value = [int(lo) | (int(hi) << 64) for lo, hi in zip(rows["lookup_lo"], rows["lookup_hi"])]
    want = np.array([[(v >> s) & 15 for v in value] for s in shifts], dtype=np.uint8)
    assert np.array_equal(got, want)
    # RAM RA: 4 chunks of the 16-bit remapped address, cold when the cycle has no RAM access
    ram_shifts = [(4 - 1 - i) * bits for i in range(4)]
    ram = R.onehot(dt.fields["ram_addr"][1], 4, ram_shifts, bits, valid_offset=dt.fields["ram_valid"][1])
    want_ram = np.array([[(int(a) >> s) & 15 for a in rows["ram_addr"]] for s in ram_shifts], dtype=np.uint8)
    want_ram[:, rows["ram_valid"] == 0] = 0xFF
    assert np.array_equal(ram.download(), want_ram)
    # the lazily bound member is indifferent to where its indices came from
    V, F, K, n_vars = 8, 4, 16, 10
    tables = rand_fr(V * F * K, 78).reshape(V * F, K, 4)
    w, coeffs = rand_fr(n_vars, 79), rand_fr(V, 80)
    a = ctx.member_lazy_ra_uniform(src, tables, V, F, coeffs, w)
    b = ctx.member_lazy_ra_uniform(ctx.onehot(want, K), tables, V, F, coeffs, w)
    bind = None
    for rnd in range(n_vars):
        ea, _ = a.prove_round(bind, want_aux=True)
        eb, _ = b.prove_round(bind, want_aux=True)
        assert np.array_equal(ea, eb), rnd
        bind = rand_challenge(81 + rnd)
    with pytest.raises(ffi.JoltError):
        R.onehot(0, 4, [30], 4)  # chunk beyond the field


def test_lookahead_windows_and_sentinel_rows(ctx):
    """The extractors' window over a physical trace shorter than the padded domain (RandomAccessRows::window, optimized/rows.rs:58-66:
    current row or the NEXT row, padding rows beyond the trace, no next row at the last cycle) and InstructionCycleRow's sentinel
    packing (pc_plus_one / ram_address_plus_one: 0 = None, optimized/instruction_read_raf.rs:82-123)."""
    n_rows, cycles = 700, 1024
    rng = np.random.default_rng(90)
    dt = np.dtype([("lookup_lo", "<u8"), ("lookup_hi", "<u8"), ("pc_plus_one", "<u8"), ("ram_plus_one", "<u8"), ("next_pc", "<u8"), ("imm", "<i4"), ("flag", "u1"), ("pad", "u1", 3)])
    rows = np.zeros(n_rows, dtype=dt)
    rows["next_pc"] = rng.integers(0, 2**64, size=n_rows, dtype=np.uint64)
    rows["imm"] = rng.integers(-2**31, 2**31, size=n_rows)
    rows["flag"] = rng.integers(0, 2, size=n_rows)
    rows["pc_plus_one"] = rng.integers(0, 2**20, size=n_rows)
    rows["pc_plus_one"][rng.random(n_rows) < 0.1] = 0
    rows["ram_plus_one"] = rng.integers(1, 2**16 + 1, size=n_rows)
    rows["ram_plus_one"][rng.random(n_rows) < 0.5] = 0
    rows["ram_plus_one"][:3] = [1, 2**16, 0]  # address 0, the largest address, no access
    R = ffi.Rows(ctx, rows)

    def want(field, lookahead, padding_value, none_value, signed):
        vals = []
        for j in range(cycles):
            src = j + lookahead
            if lookahead and src >= cycles:
                vals.append(none_value)
            elif src >= n_rows:
                vals.append(padding_value)
            else:
                vals.append(int(rows[field][src]))
        return O.to_mont([v % O.R_MOD for v in vals])

    for field, width, signed in (("next_pc", 8, False), ("imm", 4, True), ("flag", 1, False)):
        off = dt.fields[field][1]
        for lookahead, pad, none in ((0, 0, 0), (1, 0, 0), (1, 1, -5), (0, 7, 0)):
            got = R.window_table(off, width, signed=signed, lookahead=lookahead, cycles=cycles, padding_value=pad, none_value=none).download()
            assert np.array_equal(got, want(field, lookahead, pad, none, signed)), (field, lookahead, pad, none)
    with pytest.raises(ffi.JoltError) as e:
        R.window_table(0, 8, cycles=512)  # the physical trace does not fit the cycle domain
    assert e.value.status == 5
    # sentinel-packed addresses -> hot-index columns: bytecode pc (5 chunks of 4 bits), RAM address (4 chunks)
    for field, chunks in (("pc_plus_one", 5), ("ram_plus_one", 4)):
        shifts = [(chunks - 1 - i) * 4 for i in range(chunks)]
        got = R.onehot_sentinel(dt.fields[field][1], 8, shifts, 4, cycles=cycles).download()
        exp = np.full((chunks, cycles), 0xFF, dtype=np.uint8)
        for j in range(n_rows):
            v = int(rows[field][j])
            if v:
                for i, s in enumerate(shifts):
                    exp[i, j] = ((v - 1) >> s) & 15
        assert np.array_equal(got, exp), field
    # a 16-byte field whose decrement borrows across the 64-bit halves
    wide = np.zeros(4, dtype=np.dtype([("lo", "<u8"), ("hi", "<u8")]))
    wide["lo"], wide["hi"] = [0, 1, 0, 5], [1, 0, 0, 2]
    got = ffi.Rows(ctx, wide).onehot_sentinel(0, 16, [60, 64, 0], 4).download()
    vals = [(int(lo) | (int(hi) << 64)) for lo, hi in zip(wide["lo"], wide["hi"])]
    exp = np.array([[0xFF if v == 0 else ((v - 1) >> s) & 15 for v in vals] for s in (60, 64, 0)], dtype=np.uint8)
    assert np.array_equal(got, exp)
