"""GPU parity: G1 MSM and the HyperKZG prover pieces through the C ABI vs the CPU oracle.
Points are compared as group elements (the reference's PartialEq on Projective and its compressed-affine wire form):
the Jacobian representative is free.  Mirrors crates/jolt-crypto/tests/group_laws.rs:69-78,135-146 and
crates/jolt-hyperkzg/tests/commit_open_verify.rs."""
import random

import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import ffi
from util import rand_challenge, rand_fr

pytestmark = pytest.mark.gpu
R = O.R_MOD


@pytest.fixture(scope="module")
def ctx():
    c = ffi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def srs_small(ctx):
    beta = rand_fr(1, 77)[0]
    host = O.srs_setup_from_secret(beta, 300)
    return beta, host, ctx.srs_upload(host)


def same_point(a, b):
    return O.g1_eq(a, b) and O.g1_serialize_compressed(a) == O.g1_serialize_compressed(b)


def test_srs_roundtrip_and_device_setup(ctx, srs_small):
    beta, host, dev = srs_small
    back = dev.download()
    assert all(same_point(back[i], host[i]) for i in range(len(host)))
    # device-side setup_from_secret == the reference's repeated scalar_mul (scheme.rs:54-73)
    dev2 = ctx.srs_setup_from_secret(beta, 300, O.g1_generator())
    back2 = dev2.download()
    assert all(same_point(back2[i], host[i]) for i in range(300))
    # identity among the bases survives the affine round trip
    mixed = np.stack([host[0], O.g1_identity(), host[1]])
    d3 = ctx.srs_upload(mixed).download()
    assert O.g1_is_identity(d3[1]) and same_point(d3[2], host[1])


@pytest.mark.parametrize("n", [0, 1, 2, 3, 31, 32, 33, 257, 300])
def test_msm_matches_oracle(ctx, srs_small, n):
    _, host, dev = srs_small
    scalars = rand_fr(n, 100 + n) if n else O.fr_array(0)
    got = ctx.msm(dev, scalars)
    want = O.g1_msm_pippenger(host[:n], scalars) if n else O.g1_identity()
    assert same_point(got, want)
    assert O.g1_on_curve(got)


def test_msm_corner_scalars_and_repeated_bases(ctx):
    rng = random.Random(5)
    g = O.g1_generator()
    b = O.g1_scalar_mul(g, O.to_mont([5])[0])
    # repeated / negated / identity bases hit P+P, P+(-P) and infinity inside buckets
    bases = np.stack([b, b, O.g1_neg(b), b, O.g1_identity(), b, O.g1_neg(b)] * 9)
    vals = [7, 7, 7, 3, 11, 0, 1, R - 1, R - 2, 2**64 - 1, 2**128 + 5, 1 << 253] * 6
    vals = vals[: len(bases)]
    scalars = O.to_mont(vals)
    dev = ctx.srs_upload(bases)
    got = ctx.msm(dev, scalars)
    assert same_point(got, O.g1_msm_naive(bases, scalars))
    # all-equal digits: every point lands in the same bucket of each window (heavy-bucket path)
    n = 2000
    srs = ctx.srs_setup_from_secret(rand_fr(1, 9)[0], n, g)
    host = srs.download()
    ones = O.to_mont([1] * n)
    assert same_point(ctx.msm(srs, ones), O.g1_msm_pippenger(host, ones))
    same = np.repeat(rand_fr(1, 10), n, axis=0)
    assert same_point(ctx.msm(srs, same), O.g1_msm_pippenger(host, same))
    with pytest.raises(ffi.JoltError) as e:  # more scalars than bases: SrsTooSmall / the reference's length-mismatch panic
        ctx.msm(srs, rand_fr(n + 1, 3))
    assert e.value.status == 9


@pytest.mark.parametrize("n", [5000, 70001])
def test_msm_skewed_scalars_multi_segment_buckets(ctx, n):
    """Boolean / tiny / mostly-equal scalars (witness-like): buckets far above the per-lane cap, split into several
    wavefront segments and recombined; the histogram and scatter see the same-address atomics path."""
    srs = ctx.srs_setup_from_secret(rand_fr(1, 11)[0], n, O.g1_generator())
    host = srs.download()
    rng = np.random.default_rng(n)
    bits = O.fr_from_u64(rng.integers(0, 2, size=n, dtype=np.uint64))
    assert same_point(ctx.msm(srs, bits), O.g1_msm_pippenger(host, bits))
    tiny = O.fr_from_u64(rng.integers(0, 4, size=n, dtype=np.uint64) * np.uint64(0x10001))
    assert same_point(ctx.msm(srs, tiny), O.g1_msm_pippenger(host, tiny))
    mixed = rand_fr(n, 12)
    mixed[rng.random(n) < 0.7] = rand_fr(1, 13)[0]  # 70 % of the scalars equal, the rest uniform
    assert same_point(ctx.msm(srs, mixed), O.g1_msm_pippenger(host, mixed))
    neg_one = np.repeat(O.to_mont([R - 1]), n, axis=0)  # every window all-ones -> carries ripple to the top window
    assert same_point(ctx.msm(srs, neg_one), O.g1_msm_pippenger(host, neg_one))


def test_msm_small_scalars_and_device_table(ctx, srs_small):
    _, host, dev = srs_small
    small = O.fr_from_u64(np.arange(300, dtype=np.uint64) * 977 % 65536)  # witness-like <= 16-bit scalars
    tab = ctx.upload(small)
    assert same_point(ctx.msm(dev, tab), O.g1_msm_pippenger(host, small))
    assert same_point(ctx.msm(dev, tab, n=100), O.g1_msm_pippenger(host[:100], small[:100]))


def test_hyperkzg_pieces_match_oracle(ctx):
    ell = 7
    n = 1 << ell
    evals, point = rand_fr(n, 200), np.stack([rand_challenge(210 + k) for k in range(ell)])
    tab = ctx.upload(evals)
    levels = ctx.hyperkzg_fold(tab, point)
    want_levels = O.hyperkzg_fold_polynomials(evals, point)
    assert [len(t) for t in levels] == [len(w) for w in want_levels]
    for t, w in zip(levels, want_levels):
        assert np.array_equal(t.download(), w)
    u = rand_fr(3, 220)
    v = ctx.hyperkzg_eval3(levels, u)
    for t in range(3):
        for j in range(ell):
            assert np.array_equal(v[t, j], O.kzg_eval_univariate(want_levels[j], u[t]))
    q = rand_fr(1, 230)[0]
    b = ctx.hyperkzg_rlc(levels, q).download()
    want_b = np.zeros((n, 4), dtype=np.uint64)
    qj = O.to_mont([1])
    for w in want_levels:
        want_b[: len(w)] = O.fr_add(want_b[: len(w)], O.fr_mul(w, np.repeat(qj, len(w), axis=0)))
        qj = O.fr_mul(qj, q.reshape(1, 4))
    assert np.array_equal(b, want_b)
    # chunk-boundary lengths of the suffix scan (chunks of 64 coefficients; 4097 and 2^17 + 5 need a second level of chunk values)
    for m in (1, 2, 3, 64, 65, 128, 4096, 4097, 5000, (1 << 17) + 5):
        f = rand_fr(m, 240 + m)
        h = ctx.hyperkzg_witness_poly(ctx.upload(f), u[0])
        assert len(h) == max(m - 1, 0)
        if m > 1:
            assert np.array_equal(h.download(), O.kzg_witness_polynomial(f, u[0]))
    with pytest.raises(ffi.JoltError) as e:
        ctx.hyperkzg_fold(tab, point[:0])
    assert e.value.status == 10  # HyperKZGError::EmptyPoint


@pytest.mark.parametrize("ell", [1, 2, 5, 8])
def test_hyperkzg_commit_open_bit_exact_with_oracle(ctx, ell):
    n = 1 << ell
    beta = rand_fr(1, 300 + ell)[0]
    host_srs = O.srs_setup_from_secret(beta, n + 1)
    srs = ctx.srs_upload(host_srs)
    evals, point = rand_fr(n, 310 + ell), np.stack([rand_challenge(320 + k) for k in range(ell)])
    tab = ctx.upload(evals)
    assert same_point(ctx.hyperkzg_commit(srs, tab), O.kzg_commit(evals, host_srs))
    got = ctx.hyperkzg_open(srs, tab, point, label=9)
    want = O.hyperkzg_open(host_srs, evals, point, label=9)
    assert np.array_equal(got["challenges"], want["challenges"])  # transcripts agree byte for byte (compressed points + v)
    assert np.array_equal(got["v"], want["v"])
    for i in range(ell - 1):
        assert same_point(got["com"][i], want["com"][i])
    for t in range(3):
        assert same_point(got["w"][t], want["w"][t])
    small = ctx.srs_upload(host_srs[: n // 2 if n > 1 else 0])
    with pytest.raises(ffi.JoltError) as e:
        ctx.hyperkzg_commit(small, tab)
    assert e.value.status == 9


@pytest.mark.parametrize("ell,window_bits", [(2, 10), (3, 10), (7, 10), (10, 13)])
def test_hyperkzg_open_with_window_tables_is_the_oracle_proof(ctx, ell, window_bits):
    """with window tables over the SRS the opening takes the fixed-base method and commits the witness polynomials at r and -r as commit(X q) +- r commit(q) + alpha G_0
    from ONE digit sort (hyperkzg.hip): the proof must be the one kzg.rs:108-116 computes from three independent commitments -- the oracle's, point for point"""
    n = 1 << ell
    beta = rand_fr(1, 700 + ell)[0]
    host_srs = O.srs_setup_from_secret(beta, n + 1)
    srs = ctx.srs_upload(host_srs)
    ctx.srs_precompute_windows(srs, window_bits, 1)  # min_terms = 1: every MSM of the opening, the pair included, on the tables
    for seed, small in ((710, False), (720, True)):
        evals = rand_fr(n, seed + ell)
        if small:  # 64-bit coefficients with repeats: the heavy-bucket path under both passes of the pair
            evals = O.fr_from_u64(np.random.default_rng(seed).integers(0, 5, size=n).astype(np.uint64) * np.uint64(0x0123456789ABCDEF))
        point = np.stack([rand_challenge(730 + k) for k in range(ell)])
        tab = ctx.upload(evals)
        got = ctx.hyperkzg_open(srs, tab, point, label=11)
        want = O.hyperkzg_open(host_srs, evals, point, label=11)
        assert np.array_equal(got["challenges"], want["challenges"])
        assert np.array_equal(got["v"], want["v"])
        for i in range(ell - 1):
            assert same_point(got["com"][i], want["com"][i])
        for t in range(3):
            assert same_point(got["w"][t], want["w"][t]), (small, t)
        tab.free()


def test_eval3_over_many_workgroups_matches_oracle(ctx):
    """k_horner3_strided: coefficients 256 apart per lane, lane and workgroup weights -- lengths around the 4096-coefficient workgroup tile"""
    u = rand_fr(3, 221)
    for m in (1, 255, 256, 257, 4095, 4096, 4097, 3 * 4096 + 17, 1 << 16):
        f = rand_fr(m, 250 + m % 97)
        tab = ctx.upload(f)
        v = ctx.hyperkzg_eval3([tab], u)
        for t in range(3):
            assert np.array_equal(v[t, 0], O.kzg_eval_univariate(f, u[t])), (m, t)
        tab.free()


@pytest.mark.parametrize("ell,window_bits", [(11, 10), (12, 11), (13, 12), (16, 13)])
def test_open_with_one_pass_quotient_and_batched_levels_is_the_oracle_proof(ctx, ell, window_bits):
    """openings of 2^11 ... 2^16 coefficients on the window tables: the pair's quotient by X^2 - r^2 in one pass (suffix_scan<2>, one and two levels of chunk values),
    h at r^2 through suffix_scan<1>, the evaluations at r / -r / r^2 through k_horner3_strided<PLUS_MINUS>, and the level commitments below the tables' crossover as
    ONE batch of the per-window kernels (msm.hip msm_batch_enqueue): point for point the oracle's proof"""
    n = 1 << ell
    beta = rand_fr(1, 800 + ell)[0]
    host_srs = O.srs_setup_from_secret(beta, n + 1)
    srs = ctx.srs_upload(host_srs)
    ctx.srs_precompute_windows(srs, window_bits, 1 << (ell - 2))  # the top two levels and the witness MSMs on the tables, the rest in the batch
    evals = rand_fr(n, 810 + ell)
    evals[n // 3:n // 3 + 40] = evals[7]  # a repeated scalar: an over-full bucket in some windows of the batch
    point = np.stack([rand_challenge(830 + k) for k in range(ell)])
    tab = ctx.upload(evals)
    got = ctx.hyperkzg_open(srs, tab, point, label=12)
    want = O.hyperkzg_open(host_srs, evals, point, label=12)
    assert np.array_equal(got["challenges"], want["challenges"])
    assert np.array_equal(got["v"], want["v"])
    for i in range(ell - 1):
        assert same_point(got["com"][i], want["com"][i]), i
    for t in range(3):
        assert same_point(got["w"][t], want["w"][t]), t
    tab.free()


@pytest.mark.parametrize("log_n", [20, 22])
def test_full_size_msm_is_the_kzg_commitment(ctx, log_n):
    """N = 2^20 / 2^22 terms (BASELINE configs[2] scale): commit(p) with bases beta^i*G must equal p(beta)*G
    (size-independent identity; p(beta) from the oracle's Horner evaluation, one scalar multiplication)."""
    n = 1 << log_n
    beta = rand_fr(1, 400)[0]
    g = O.g1_generator()
    srs = ctx.srs_setup_from_secret(beta, n, g)
    for kind in ("full", "u64"):
        scalars = rand_fr(n, 401) if kind == "full" else O.fr_from_u64(np.random.default_rng(402).integers(0, 2**64, size=n, dtype=np.uint64))
        got = ctx.msm(srs, ctx.upload(scalars))
        pbeta = O.kzg_eval_univariate(scalars, beta)
        assert same_point(got, O.g1_scalar_mul(g, pbeta)), kind


@pytest.mark.parametrize("lds_sort", ["0", "1"])
def test_msm_sort_paths_agree(monkeypatch, lds_sort):
    """Both counting sorts of the MSM (per-workgroup LDS histograms / one global atomic per key) give the point
    p(beta) G for uniform, 64-bit and all-equal scalars at 2^17 terms (above the size where the LDS path switches on)."""
    monkeypatch.setenv("JOLT_MSM_LDS_SORT", lds_sort)
    c = ffi.Context(0)
    n = 1 << 17
    beta = rand_fr(1, 301)[0]
    srs = c.srs_setup_from_secret(beta, n, O.g1_generator())
    cases = [rand_fr(n, 302), O.fr_from_u64(np.random.default_rng(303).integers(0, 2**64, size=n, dtype=np.uint64)), O.fr_from_u64(np.full(n, 3, dtype=np.uint64))]
    for scalars in cases:
        got = c.msm(srs, c.upload(scalars))
        want = O.g1_scalar_mul(O.g1_generator(), O.kzg_eval_univariate(scalars, beta))
        assert same_point(got, want)
    c.close()


def test_fixed_base_msm_matches_oracle_and_per_window_path(ctx):
    """jolt_srs_precompute_windows: with 2^(c*w) * P_i resident all windows of an MSM share one bucket set (msm_fixed.hip).  Same
    points as the oracle's Pippenger for uniform scalars of every length class, the field's corner values, repeated bases and an
    identity among the bases; skewed (witness-like) scalars are routed back to the per-window method and agree as well."""
    n_srs = 2200
    beta = rand_fr(1, 600)[0]
    host = O.srs_setup_from_secret(beta, n_srs)
    host[7] = host[3]            # repeated base
    host[11] = O.g1_identity()   # identity among the bases
    host[12] = O.g1_neg(host[5])
    dev = ctx.srs_upload(host)
    ctx.srs_precompute_windows(dev, 10, 1)  # c = 10: 26 windows, 2 segments of 512 buckets, every MSM length takes the new path
    for n in (1, 2, 63, 64, 65, 1000, 2048, 2200):
        scalars = rand_fr(n, 610 + n)
        assert same_point(ctx.msm(dev, scalars), O.g1_msm_pippenger(host[:n], scalars)), n
    corner = O.to_mont([0, 1, R - 1, 2, R - 2, (R - 1) // 2, 1 << 253, (1 << 200) - 1, 511, 512, 513, 1 << 9, (1 << 10) - 1, 1 << 10])
    assert same_point(ctx.msm(dev, corner), O.g1_msm_pippenger(host[: len(corner)], corner))
    same = np.repeat(rand_fr(1, 620), 2048, axis=0)  # every digit equal: 26 buckets of 2048 points (the workgroup-wide bucket sum)
    assert same_point(ctx.msm(dev, same), O.g1_msm_pippenger(host[:2048], same))
    five = O.to_mont([sum(5 << (10 * w) for w in range(25))] * 2048)  # every window's digit is 5: ONE bucket of 25 * 2048 points
    assert same_point(ctx.msm(dev, five), O.g1_msm_pippenger(host[:2048], five))
    small = O.fr_from_u64(np.random.default_rng(621).integers(0, 2, size=2048, dtype=np.uint64))  # 0/1 flags: one huge bucket -> fallback
    assert same_point(ctx.msm(dev, small), O.g1_msm_pippenger(host[:2048], small))
    zeros = np.zeros((500, 4), dtype=np.uint64)
    assert O.g1_is_identity(ctx.msm(dev, zeros))


@pytest.mark.parametrize("window_bits", [25, 26])
def test_fixed_base_msm_wide_windows_match_oracle(ctx, window_bits):
    """25/26-bit windows (segments of 2048 buckets, 10 windows at c = 26): same points as the oracle's Pippenger on uniform, corner
    and all-equal scalars"""
    n_srs = 3000
    host = O.srs_setup_from_secret(rand_fr(1, 640)[0], n_srs)
    dev = ctx.srs_upload(host)
    ctx.srs_precompute_windows(dev, window_bits, 1)
    for n in (1, 65, 3000):
        scalars = rand_fr(n, 641 + n)
        assert same_point(ctx.msm(dev, scalars), O.g1_msm_pippenger(host[:n], scalars)), n
    corner = O.to_mont([0, 1, R - 1, 2, R - 2, (R - 1) // 2, 1 << 253, (1 << 25) - 1, 1 << 25, (1 << 25) + 1, (1 << 26) - 1, 1 << 24, 2047, 2048, 2049])
    assert same_point(ctx.msm(dev, corner), O.g1_msm_pippenger(host[: len(corner)], corner))
    same = np.repeat(rand_fr(1, 650), 3000, axis=0)
    assert same_point(ctx.msm(dev, same), O.g1_msm_pippenger(host, same))
    dev.free()


@pytest.mark.parametrize("window_bits", [0, 13, 26])
def test_fixed_base_msm_at_2_20_is_the_kzg_commitment(ctx, window_bits):
    """2^20 terms over window-precomputed bases (auto: c = 18, 15 windows; c = 13: 20 windows; c = 26: 10): commit(p) == p(beta) G, for uniform
    scalars (new path) and 64-bit scalars (skew fallback), and a prefix MSM of 2^19 + 5 terms."""
    n = 1 << 20
    beta = rand_fr(1, 630)[0]
    g = O.g1_generator()
    srs = ctx.srs_setup_from_secret(beta, n, g)
    ctx.srs_precompute_windows(srs, window_bits, 1 << 12)
    full = rand_fr(n, 631)
    assert same_point(ctx.msm(srs, ctx.upload(full)), O.g1_scalar_mul(g, O.kzg_eval_univariate(full, beta)))
    m = (1 << 19) + 5
    assert same_point(ctx.msm(srs, ctx.upload(full[:m])), O.g1_scalar_mul(g, O.kzg_eval_univariate(full[:m], beta)))
    u64 = O.fr_from_u64(np.random.default_rng(632).integers(0, 2**64, size=n, dtype=np.uint64))
    assert same_point(ctx.msm(srs, ctx.upload(u64)), O.g1_scalar_mul(g, O.kzg_eval_univariate(u64, beta)))
    srs.free()


@pytest.mark.parametrize("window_bits", [0, 13, 23])
def test_capacity_sort_is_the_kzg_commitment_and_falls_back_on_the_device(ctx, window_bits):
    """The sort without a histogram pass (msm_fixed.hip section 2d: capacity regions from the digit model of uniform scalars, taken when the caller marks the scalars
    full-width).  Uniform scalars: commit(p) == p(beta) G at 2^20 terms and on a ragged prefix, like the exact sort.  Scalars that are NOT what the caller claimed --
    all equal, 64-bit, a few thousand distinct values, zero -- overflow their regions; the exact passes enqueued behind the capacity sort must then produce the result
    (no host round trip: the flag lives in device memory), again equal to p(beta) G and to the point the exact sort (full_width = False) gives."""
    n = 1 << 20
    beta = rand_fr(1, 640)[0]
    g = O.g1_generator()
    srs = ctx.srs_setup_from_secret(beta, n, g)
    ctx.srs_precompute_windows(srs, window_bits, 1 << 12)
    full = rand_fr(n, 641)
    want = O.g1_scalar_mul(g, O.kzg_eval_univariate(full, beta))
    tab = ctx.upload(full)
    assert same_point(ctx.msm(srs, tab, full_width=True), want)
    assert same_point(ctx.msm(srs, tab, full_width=True), ctx.msm(srs, tab))  # twice: the workspace of the first run (flag, cursors) does not leak into the second
    m = (1 << 19) + 77
    assert same_point(ctx.msm(srs, tab, m, full_width=True), O.g1_scalar_mul(g, O.kzg_eval_univariate(full[:m], beta)))
    rng = np.random.default_rng(642)
    skewed = {"all equal": np.repeat(rand_fr(1, 643), n, axis=0),
              "64-bit": O.fr_from_u64(rng.integers(0, 2**64, size=n, dtype=np.uint64)),
              "4096 distinct values": rand_fr(4096, 644)[rng.integers(0, 4096, size=n)],
              "zero": np.zeros((n, 4), dtype=np.uint64)}
    for name, scalars in skewed.items():
        t = ctx.upload(np.ascontiguousarray(scalars))
        got = ctx.msm(srs, t, full_width=True)
        assert same_point(got, ctx.msm(srs, t)), name
        if name != "4096 distinct values":
            assert same_point(got, O.g1_scalar_mul(g, O.kzg_eval_univariate(np.ascontiguousarray(scalars), beta))), name
        t.free()
    assert same_point(ctx.msm(srs, tab, full_width=True), want)  # and a uniform one again after the fallbacks
    tab.free()
    srs.free()


@pytest.mark.parametrize("world,block,n_global", [(2, 64, 1024), (4, 32, 1024), (8, 16, 2048), (2, 2048, 16384)])
def test_block_cyclic_shares_add_up_to_the_msm(ctx, world, block, n_global):
    """The block-cyclic term assignment of the sharded PCS legs: rank g's compact SRS holds the powers beta^i with
    (i / block) % world == g in index order, and the ranks' shares of any prefix MSM -- full blocks, a ragged last block, less than
    one block -- add up to the MSM over the full SRS.  (2, 2048, 16384): the compact SRS is long enough for window tables.)"""
    beta = rand_fr(1, 91)[0]
    full = ctx.srs_setup_from_secret(beta, n_global, O.g1_generator())
    full_pts = full.download()
    shares = [ctx.srs_setup_from_secret_blocks(beta, n_global, O.g1_generator(), block, g, world) for g in range(world)]
    for g, s in enumerate(shares):
        assert len(s) == n_global // world
        pts = s.download()
        own = np.concatenate([np.arange(b * block, (b + 1) * block) for b in range(n_global // block) if b % world == g])
        assert all(same_point(pts[j], full_pts[i]) for j, i in list(enumerate(own))[:: max(1, len(own) // 64)])
        if len(s) >= 4096:
            ctx.srs_precompute_windows(s, min_terms=1024)
    scalars = rand_fr(n_global, 92)
    scalars[5] = 0
    scalars[7] = O.fr_array(1)[0]
    table = ctx.upload(scalars)
    for n in (n_global, n_global - 1, 3 * block + 17, block - 5, 1, 0):
        want = ctx.msm(full, table, n) if n else O.g1_identity()
        acc = O.g1_identity()
        for g, s in enumerate(shares):
            acc = ffi.host_g1_add(acc, ctx.msm_blocks(s, table, n, block, g, world))
        assert same_point(acc, want), n
    with pytest.raises(ffi.JoltError):
        ctx.srs_setup_from_secret_blocks(beta, n_global + 1, O.g1_generator(), block, 0, world)


@pytest.mark.parametrize("world,n_global", [(2, 256), (4, 1024), (8, 2048), (2, 16384)])
def test_subtree_shares_add_up_to_the_msm(ctx, world, n_global):
    """The subtree term assignment (term_map.hip.h; tests/subtree_model.py): rank g's compact SRS holds the powers of the indices it
    owns in index order, and the ranks' shares of any prefix MSM add up to the MSM over the full SRS ((2, 16384): with window tables)."""
    import subtree_model as M
    gamma = world.bit_length() - 1
    beta = rand_fr(1, 93)[0]
    full = ctx.srs_setup_from_secret(beta, n_global, O.g1_generator())
    full_pts = full.download()
    shares = [ctx.srs_setup_from_secret_subtree(beta, n_global, O.g1_generator(), g, world) for g in range(world)]
    for g, s in enumerate(shares):
        assert len(s) == n_global // world
        pts = s.download()
        for c in list(range(0, len(s), max(1, len(s) // 48))) + [len(s) - 1]:
            assert same_point(pts[c], full_pts[M.insert(c, g, gamma)]), (g, c)
        if len(s) >= 4096:
            ctx.srs_precompute_windows(s, min_terms=1024)
    scalars = rand_fr(n_global, 94)
    scalars[3] = 0
    table = ctx.upload(scalars)
    for n in (n_global, n_global - 1, n_global // 2, n_global // 2 + 5, world + 1, world, world - 1, 1, 0):
        want = ctx.msm(full, table, n) if n else O.g1_identity()
        acc = O.g1_identity()
        for g, s in enumerate(shares):
            acc = ffi.host_g1_add(acc, ctx.msm_subtree(s, table, n, g, world))
        assert same_point(acc, want), n
    with pytest.raises(ffi.JoltError):
        ctx.srs_setup_from_secret_subtree(beta, n_global, O.g1_generator(), 0, 3)
