"""K = 256 one-hot sources (16-bit hot indices, 0xFFFF = cold): the chunk width the reference's OneHotConfig picks for long traces
(crates/jolt-witness/src/one_hot.rs: log_k_chunk = 8 above 2^25 cycles; RaChunkParams in crates/jolt-claims).  Every consumer of a
source -- materialise, pushforward, the lazily bound RA and booleanity members, the K x T grid commitment and joint polynomial,
the Dory one-hot tier-1 rows -- is checked (a) wide against narrow on the same K = 16 data, bit for bit, and (b) at K = 256 with
index 255 hot against the oracle over the materialised dense columns."""
import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import ffi
from kzg_check import same_point
from util import rand_challenge, rand_fr

pytestmark = pytest.mark.gpu
COLD16 = 0xFFFF


@pytest.fixture(scope="module")
def ctx():
    c = ffi.Context(0)
    yield c
    c.close()


def widen(idx8):
    out = idx8.astype(np.uint16)
    out[idx8 == 0xFF] = COLD16
    return out


def wide_columns(n_polys, T, K, seed, cold):
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, K, size=(n_polys, T)).astype(np.uint16)
    idx[:, 0] = K - 1  # the top entry is always exercised
    if cold:
        idx[rng.random((n_polys, T)) < cold] = COLD16
    return idx


def dense_column(idx_row, table):
    out = np.zeros((idx_row.shape[0], 4), dtype=np.uint64)
    hot = idx_row != COLD16
    out[hot] = table[idx_row[hot]]
    return out


def halves(col16):
    """a 16-bit column over K = 256 as two 8-bit columns over K = 128 (what the byte-indexed oracle entry points take)"""
    lo = np.where((col16 != COLD16) & (col16 < 128), col16, 0xFF).astype(np.uint8)
    hi = np.where((col16 != COLD16) & (col16 >= 128), col16 - 128, 0xFF).astype(np.uint8)
    return lo, hi


def run_member(dev, orc, n_vars, degree_from_aux, seed):
    claim = orc.input_claim()
    bind = None
    polys = []
    for rnd in range(n_vars):
        want = orc.prove_round(bind, claim)
        evals, aux = dev.prove_round(bind, want_aux=True)
        got = degree_from_aux(aux, evals, claim)
        assert np.array_equal(got, want), f"round {rnd}"
        polys.append(evals)
        bind = rand_challenge(seed + rnd) if rnd % 3 else rand_fr(1, seed + rnd)[0]
        claim = O.univariate_evaluate(want, bind)
    orc.finish_rounds(bind)
    dev.finish(bind)
    fv, ofv = dev.final_values(), orc.final_values()
    assert np.array_equal(fv[:-1], ofv[1:]) and np.array_equal(fv[-1], ofv[0])
    return polys


def test_upload_round_trip_and_range_check(ctx):
    idx = wide_columns(3, 64, 256, 1, 0.3)
    src = ctx.onehot(idx, 256)
    assert np.array_equal(src.download(), idx)
    with pytest.raises(ffi.JoltError):
        ctx.onehot(np.full((1, 16), 256, dtype=np.uint16), 256)  # hot index outside the scale table
    with pytest.raises(ffi.JoltError):
        ctx.onehot(np.zeros((1, 16), dtype=np.uint16), 1 << 16)  # 0xFFFF must stay free for "cold"
    src.free()


@pytest.mark.parametrize("n_vars,cold", [(9, 0.0), (11, 0.3)])
def test_wide_source_is_bit_identical_to_narrow(ctx, n_vars, cold):
    T, N, K, V, F = 1 << n_vars, 4, 16, 2, 2
    rng = np.random.default_rng(n_vars)
    idx8 = rng.integers(0, K, size=(N, T), dtype=np.uint8)
    if cold:
        idx8[rng.random((N, T)) < cold] = 0xFF
    a, b = ctx.onehot(idx8, K), ctx.onehot(widen(idx8), K)
    tables = rand_fr(N * K, 2).reshape(N, K, 4)
    for p in range(N):
        t = ctx.upload(tables[p])
        assert np.array_equal(a.materialize(p, t).download(), b.materialize(p, t).download())
    w = ctx.upload(rand_fr(T, 3))
    assert np.array_equal(a.pushforward(w).download(), b.pushforward(w).download())
    point, coeffs, scale = rand_fr(n_vars, 4), rand_fr(V, 5), rand_fr(1, 6)[0]
    ma = ctx.member_lazy_ra_uniform(a, tables, V, F, coeffs, point, scale=scale)
    mb = ctx.member_lazy_ra_uniform(b, tables, V, F, coeffs, point, scale=scale)
    rho = rand_fr(N, 7)
    ba = ctx.member_lazy_booleanity(a, tables, rho, point, scale=scale)
    bb = ctx.member_lazy_booleanity(b, tables, rho, point, scale=scale)
    for x, y in ((ma, mb), (ba, bb)):
        bind = None
        for rnd in range(n_vars):
            ex, ax = x.prove_round(bind, want_aux=True)
            ey, ay = y.prove_round(bind, want_aux=True)
            assert np.array_equal(ex, ey) and np.array_equal(ax, ay), rnd
            bind = rand_challenge(20 + rnd)
        x.finish(bind), y.finish(bind)
        assert np.array_equal(x.final_values(), y.final_values())
    # commitment grid and Dory tier 1
    beta = rand_fr(1, 8)[0]
    srs = ctx.srs_upload(O.srs_setup_from_secret(beta, K * T))
    ga, gb = ctx.grid_commit_onehot(srs, a), ctx.grid_commit_onehot(srs, b)
    assert all(same_point(ga[p], gb[p]) for p in range(N))
    s = rand_fr(N, 9)
    assert np.array_equal(ctx.grid_joint_polynomial([a], s, [], [], 4).download(), ctx.grid_joint_polynomial([b], s, [], [], 4).download())
    da, db = ctx.dory_commit_onehot(srs, a, 1, 64), ctx.dory_commit_onehot(srs, b, 1, 64)
    assert da.shape == db.shape and all(O.g1_eq(da[c, r], db[c, r]) for c in range(da.shape[0]) for r in range(K))


@pytest.mark.parametrize("n_vars,cold", [(8, 0.0), (10, 0.25)])
def test_k256_materialize_pushforward_match_oracle(ctx, n_vars, cold):
    T, N, K = 1 << n_vars, 3, 256
    idx = wide_columns(N, T, K, 30 + n_vars, cold)
    src = ctx.onehot(idx, K)
    tables = rand_fr(N * K, 31).reshape(N, K, 4)
    for p in range(N):
        assert np.array_equal(src.materialize(p, ctx.upload(tables[p])).download(), dense_column(idx[p], tables[p]))
    w = rand_fr(T, 32)
    G = src.pushforward(ctx.upload(w)).download().reshape(N, K, 4)
    for p in range(N):
        lo, hi = halves(idx[p])
        want = np.concatenate([O.onehot_pushforward(lo, 128, w), O.onehot_pushforward(hi, 128, w)])
        assert np.array_equal(G[p], want), p


@pytest.mark.parametrize("V,F,n_vars,cold", [(1, 2, 7, 0.0), (2, 2, 10, 0.3)])
def test_k256_lazy_ra_member_matches_oracle(ctx, V, F, n_vars, cold):
    T, N, K = 1 << n_vars, V * F, 256
    idx = wide_columns(N, T, K, 40 + n_vars, cold)
    tables = rand_fr(N * K, 41).reshape(N, K, 4)
    w, coeffs = rand_fr(n_vars, 42), rand_fr(V, 43)
    dense = [dense_column(idx[p], tables[p]) for p in range(N)]
    lazy = ctx.member_lazy_ra_uniform(ctx.onehot(idx, K), tables, V, F, coeffs, w)
    eq = O.eq_evals(w, None)
    terms = [(coeffs[v], [0] + [1 + v * F + k for k in range(F)]) for v in range(V)]
    for rep in range(2):
        orc = O.Member.expr([eq] + dense, terms, F + 1)
        run_member(lazy, orc, n_vars, lambda aux, evals, claim: ffi.host_gruen_poly_from_q(aux[0], aux[1], evals, claim), 50)
        lazy.reset()


def test_k256_lazy_booleanity_member_matches_oracle(ctx):
    n_vars, N, K = 9, 3, 256
    T = 1 << n_vars
    idx = wide_columns(N, T, K, 60, 0.2)
    one = O.to_mont([1])[0]
    rho = list(rand_fr(N, 61))
    rho[0] = one
    eq_address = rand_fr(K, 62)
    tables = np.stack([O.fr_mul(eq_address, np.repeat(r.reshape(1, 4), K, axis=0)) for r in rho])
    w, scale = rand_fr(n_vars, 63), rand_fr(1, 64)[0]
    dense = [dense_column(idx[p], tables[p]) for p in range(N)]
    neg = lambda x: O.fr_neg(np.asarray(x).reshape(1, 4))[0]
    terms = []
    for i in range(N):
        terms.append((one, [0, 1 + i, 1 + i]))
        terms.append((neg(rho[i]), [0, 1 + i]))
    dev = ctx.member_lazy_booleanity(ctx.onehot(idx, K), tables, rho, w, scale=scale)
    orc = O.Member.expr([O.eq_evals(w, scale)] + dense, terms, 3)
    run_member(dev, orc, n_vars, lambda aux, evals, claim: ffi.host_gruen_poly_deg_3(aux[0], aux[1], evals[0], evals[1], claim), 70)


def test_k256_grid_commitments_and_joint_polynomial(ctx):
    log_t, log_k = 4, 8
    T, K = 1 << log_t, 1 << log_k
    beta = rand_fr(1, 80)[0]
    host_srs = O.srs_setup_from_secret(beta, K * T)
    srs = ctx.srs_upload(host_srs)
    idx = wide_columns(3, T, K, 81, 0.3)
    src = ctx.onehot(idx, K)
    one = O.to_mont([1])[0]

    def embed(col):
        out = np.zeros((K * T, 4), dtype=np.uint64)
        j = np.nonzero(col != COLD16)[0]
        out[col[j].astype(np.int64) * T + j] = one
        return out

    got = ctx.grid_commit_onehot(srs, src)
    for p in range(3):
        assert same_point(got[p], O.kzg_commit(embed(idx[p]), host_srs)), p
    s = rand_fr(3, 82)
    dense_vals = np.random.default_rng(83).integers(0, 2**64, size=T, dtype=np.uint64)
    d_scalar = rand_fr(1, 84)
    joint = ctx.grid_joint_polynomial([src], s, [ctx.from_u64(dense_vals)], d_scalar, log_k).download()
    want = np.zeros((K * T, 4), dtype=np.uint64)
    for p in range(3):
        want = O.fr_add(want, O.fr_mul(embed(idx[p]), np.repeat(s[p].reshape(1, 4), K * T, axis=0)))
    want[:T] = O.fr_add(want[:T], O.fr_mul(O.fr_from_u64(dense_vals), np.repeat(d_scalar[0].reshape(1, 4), T, axis=0)))
    assert np.array_equal(joint, want)


def test_k256_dory_onehot_rows(ctx):
    """commitment[chunk][row] = (sum of beta^col over the chunk's columns on that row) G, rows 0, 200 and 255"""
    beta = rand_fr(1, 90)[0]
    width, cycles, K = 256, 2048, 256
    host = O.srs_setup_from_secret(beta, width)
    dev = ctx.srs_upload(host)
    idx = wide_columns(2, cycles, K, 91, 0.3)
    idx[1, width:2 * width] = 255  # one chunk entirely on the last row
    got = ctx.dory_commit_onehot(dev, ctx.onehot(idx, K), 1, width)
    assert got.shape[:2] == (cycles // width, K)
    for ch, row in ((0, 255), (1, 255), (1, 254), (3, 200), (7, 0)):
        col = idx[1, ch * width:(ch + 1) * width]
        lo, hi = halves(np.where(col == row, col, COLD16).astype(np.uint16))
        want = O.dory_onehot_chunk(host, lo if row < 128 else hi, 128)[row % 128]
        assert O.g1_eq(got[ch, row], want), (ch, row)


def test_k256_from_packed_rows(ctx):
    """8-bit chunks of a packed address field land in a 16-bit source (log_k = 8)"""
    rng = np.random.default_rng(95)
    n = 300
    rows = np.zeros((n, 16), dtype=np.uint8)
    addr = rng.integers(0, 2**32, size=n, dtype=np.uint64)
    rows[:, :8] = addr.view(np.uint8).reshape(n, 8)
    valid = rng.random(n) < 0.8
    rows[:, 8] = valid
    dev_rows = ffi.Rows(ctx, rows)
    src = dev_rows.onehot(0, 8, [24, 16, 8, 0], 8, valid_offset=8)
    got = src.download()
    assert got.dtype == np.uint16
    for p in range(4):
        shift = [24, 16, 8, 0][p]
        want = np.where(valid, (addr >> np.uint64(shift)) & np.uint64(0xFF), COLD16).astype(np.uint16)
        assert np.array_equal(got[p], want), p
