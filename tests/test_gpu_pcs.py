"""GPU parity of the PCS legs of the benchmark step (BASELINE configs[2]): committed columns over the shared commitment grid,
the joint polynomial of the homomorphic batch, and the HyperKZG opening at BASELINE scale.

Small sizes: bit-exact against the oracle's restatements (kzg_commit of the grid-embedded coefficient vector, the RLC of the
embedded polynomials, hyperkzg_open).  2^20 / 2^22 / the bench's 2^26 grid: the size-independent identities of tests/kzg_check.py
(beta is known to the test), the same way the reference pins HyperKZG by commit -> open -> verify round trips
(crates/jolt-hyperkzg/tests/commit_open_verify.rs:29-60, crates/jolt-openings/tests/homomorphic_hyperkzg.rs:17-44)."""
import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import ffi
from jolt_amd.workload import DeviceWorkload, G1_GENERATOR
from kzg_check import check_opening, same_point
from util import rand_challenge, rand_fr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = ffi.Context(0)
    yield c
    c.close()


def embed_onehot(idx_col, k, cycles):
    """coefficient vector of a one-hot column on the K x T grid, cycle-major placement (index = address * T + cycle;
    TracePlacement, crates/jolt-kernels/src/optimized/opening.rs:340-372)"""
    out = np.zeros((k * cycles, 4), dtype=np.uint64)
    hot = idx_col != 0xFF
    j = np.nonzero(hot)[0]
    out[idx_col[hot].astype(np.int64) * cycles + j] = O.to_mont([1])[0]
    return out


def test_generator_constant_matches_oracle():
    assert np.array_equal(G1_GENERATOR, O.g1_generator())


def test_table_from_resident_integers(ctx):
    rng = np.random.default_rng(1)
    u = rng.integers(0, 2**64, size=1000, dtype=np.uint64)
    i = rng.integers(-2**63, 2**63, size=1000, dtype=np.int64)
    i[:3] = [0, -1, -2**63]
    assert np.array_equal(ctx.table_from_ints(ctx.ints(u)).download(), O.fr_from_u64(u))
    assert np.array_equal(ctx.table_from_ints(ctx.ints(i)).download(), O.fr_from_i64(i))
    assert np.array_equal(ctx.table_from_ints(ctx.ints(u), 17, 100).download(), O.fr_from_u64(u[17:117]))
    big = [0, 1, -1, 2**127 - 1, -2**127, 2**64, -2**64 - 5, 123456789 << 70]
    got = ctx.table_from_ints(ctx.ints(big, "i128")).download()
    want = O.to_mont([v % O.R_MOD for v in big])
    assert np.array_equal(got, want)
    with pytest.raises(ffi.JoltError) as e:
        ctx.table_from_ints(ctx.ints(u), 990, 20)
    assert e.value.status == 5


@pytest.mark.parametrize("log_t,k,cold", [(4, 16, 0.0), (6, 16, 0.4), (5, 4, 0.2)])
def test_grid_commitments_and_joint_polynomial_match_oracle(ctx, log_t, k, cold):
    T = 1 << log_t
    log_k = 4
    K = 1 << log_k
    rng = np.random.default_rng(10 + log_t)
    beta = rand_fr(1, 20 + log_t)[0]
    host_srs = O.srs_setup_from_secret(beta, K * T)
    srs = ctx.srs_upload(host_srs)
    idx_a = rng.integers(0, k, size=(3, T), dtype=np.uint8)
    idx_b = rng.integers(0, k, size=(2, T), dtype=np.uint8)
    if cold:
        idx_a[rng.random((3, T)) < cold] = 0xFF
    idx_a[0, :] = 0xFF if cold else idx_a[0, :]  # an entirely cold column commits to the identity
    src_a, src_b = ctx.onehot(idx_a, k), ctx.onehot(idx_b, k)
    for src, idx in ((src_a, idx_a), (src_b, idx_b)):
        got = ctx.grid_commit_onehot(srs, src)
        for p in range(idx.shape[0]):
            want = O.kzg_commit(embed_onehot(idx[p], K, T), host_srs)
            assert same_point(got[p], want), p
    dense_vals = [rng.integers(0, 2**64, size=T, dtype=np.uint64), rng.integers(-2**40, 2**40, size=T, dtype=np.int64)]
    dense = [ctx.from_u64(dense_vals[0]), ctx.from_i64(dense_vals[1])]
    dense_host = [O.fr_from_u64(dense_vals[0]), O.fr_from_i64(dense_vals[1])]
    s_oh, s_d = rand_fr(5, 31), rand_fr(2, 32)
    s_d[1] = O.to_mont([1])[0]  # unit coefficient path
    joint = ctx.grid_joint_polynomial([src_a, src_b], s_oh, dense, s_d, log_k)
    want = np.zeros((K * T, 4), dtype=np.uint64)
    for p, col in enumerate(list(idx_a) + list(idx_b)):
        e = embed_onehot(col, K, T)
        want = O.fr_add(want, O.fr_mul(e, np.repeat(s_oh[p].reshape(1, 4), K * T, axis=0)))
    for d in range(2):
        want[:T] = O.fr_add(want[:T], O.fr_mul(dense_host[d], np.repeat(s_d[d].reshape(1, 4), T, axis=0)))
    assert np.array_equal(joint.download(), want)
    # the ranks' compact arrays under the subtree assignment are the owned coefficients of the same polynomial, in index order
    for world in (2, 4, 8):
        for g in range(world):
            compact = ctx.grid_joint_polynomial_subtree([src_a, src_b], s_oh, dense, s_d, log_k, g, world).download()
            own = [ffi.host_subtree_term_index(c, g, world) for c in range(K * T // world)]
            assert compact.shape[0] == len(own) and np.array_equal(compact, want[own]), (world, g)
    # a source wider than the grid / an SRS shorter than the grid are refused
    with pytest.raises(ffi.JoltError) as e:
        ctx.grid_commit_onehot(ctx.srs_upload(host_srs[: K * T // 2]), ctx.onehot(np.full((1, T), k - 1, dtype=np.uint8), K))
    assert e.value.status == 9  # HyperKZGError::SrsTooSmall
    with pytest.raises(ffi.JoltError):
        ctx.grid_joint_polynomial([ctx.onehot(idx_b, 32)], s_oh[:2], [], [], log_k)


@pytest.mark.parametrize("log_t,shift,cold", [(4, 1, 0.0), (6, 1, 0.4), (6, 2, 0.3), (5, 3, 0.2)])
def test_class_sums_of_onehot_columns_match_oracle(ctx, log_t, shift, cold):
    """jolt_grid_commit_onehot_classes: class c of column p = the kzg_commit of the 0/1 polynomial on the FOLDED grid (K x T / 2^shift) that is hot at
    (hot_p(j), j >> shift) for the cycles j = c mod 2^shift -- the oracle's commitment of that polynomial over the prefix of the same SRS"""
    T, K = 1 << log_t, 16
    rng = np.random.default_rng(40 + log_t + shift)
    host_srs = O.srs_setup_from_secret(rand_fr(1, 60 + log_t)[0], K * T)
    srs = ctx.srs_upload(host_srs)
    idx = rng.integers(0, K, size=(3, T), dtype=np.uint8)
    if cold:
        idx[rng.random((3, T)) < cold] = 0xFF
    idx[2, 1::2] = 0xFF  # a column with an empty class: the identity
    got = ctx.grid_commit_onehot_classes(srs, ctx.onehot(idx, K), shift)
    assert got.shape == (1 << shift, 3, 12)
    Tf = T >> shift
    for c in range(1 << shift):
        for p in range(3):
            part = idx[p, c::1 << shift]  # the cycles of class c, in order: cycle j sits at folded cycle j >> shift
            want = O.kzg_commit(embed_onehot(part, K, Tf), host_srs[: K * Tf])
            assert same_point(got[c, p], want), (c, p)


@pytest.mark.parametrize("log_t,levels,background,fixed_base", [(5, 2, False, False), (6, 3, True, False), (7, 4, True, False), (10, 2, True, True), (12, 4, False, True)])
def test_grid_hint_and_the_opening_by_linearity_are_the_plain_opening(ctx, log_t, levels, background, fixed_base):
    """jolt_grid_hint_begin: the hint's class sums are jolt_grid_commit_onehot_classes of each source (oracle-checked above), whether they ran in the background (lowest
    priority stream, LDS-limited to one wavefront per SIMD) or on the main stream; and jolt_host_hyperkzg_open_grid -- the first levels by linearity from the hint,
    the dense folds' MSMs in the level pipeline -- returns the proof of jolt_host_hyperkzg_open of the same joint polynomial: same transcript, equal points"""
    T, K, log_k = 1 << log_t, 16, 4
    rng = np.random.default_rng(70 + log_t + levels)
    srs = ctx.srs_setup_from_secret(rand_fr(1, 80 + log_t)[0], K * T, G1_GENERATOR)
    if fixed_base:
        ctx.srs_precompute_windows(srs, 10, 1)
    idx_a = rng.integers(0, K, size=(3, T), dtype=np.uint8)
    idx_a[:, rng.random(T) < 0.4] = 0xFF
    idx_b = rng.integers(0, K, size=(5, T), dtype=np.uint8)
    idx_b[4, 1::2] = 0xFF
    sources = [ctx.onehot(idx_a, K), ctx.onehot(idx_b, K)]
    dense = [ctx.from_u64(rng.integers(0, 2**64, size=T, dtype=np.uint64)), ctx.from_i64(rng.integers(-2**62, 2**62, size=T, dtype=np.int64))]
    s_oh, s_d, point = rand_fr(8, 90 + log_t), rand_fr(2, 91 + log_t), rand_fr(log_k + log_t, 92 + log_t)
    hint = ctx.grid_hint(srs, sources, levels, background=background)
    for s_ in range(1, levels + 1):
        want = np.concatenate([ctx.grid_commit_onehot_classes(srs, src, s_) for src in sources], axis=1)
        got = hint.download(s_)
        assert got.shape == want.shape == (1 << s_, 8, 12)
        assert all(same_point(got[c, p], want[c, p]) for c in range(1 << s_) for p in range(8)), s_
    joint = ctx.grid_joint_polynomial(sources, s_oh, dense, s_d, log_k)
    plain = ctx.hyperkzg_open(srs, joint, point, label=9)
    for lv in range(1, levels + 1):
        got = ctx.hyperkzg_open_grid(srs, joint, point, hint, lv, s_oh, dense, s_d, label=9)
        assert np.array_equal(got["challenges"], plain["challenges"]) and np.array_equal(got["v"], plain["v"]), lv
        assert all(same_point(got["com"][i], plain["com"][i]) for i in range(log_k + log_t - 1)), lv
        assert all(same_point(got["w"][t], plain["w"][t]) for t in range(3)), lv
    only_onehot = ctx.grid_joint_polynomial(sources, s_oh, [], [], log_k)  # no dense columns: the levels are the hint's combination alone
    plain = ctx.hyperkzg_open(srs, only_onehot, point, label=10)
    got = ctx.hyperkzg_open_grid(srs, only_onehot, point, hint, levels, s_oh, [], [], label=10)
    assert np.array_equal(got["challenges"], plain["challenges"]) and all(same_point(got["com"][i], plain["com"][i]) for i in range(log_k + log_t - 1))
    hint.free()
    for t in dense + [joint, only_onehot]:
        t.free()
    for src in sources:
        src.free()
    srs.free()


def test_open_with_a_supplied_level_commitment_is_the_same_proof(ctx):
    """jolt_host_hyperkzg_open_with_levels: the first level commitment handed in (here: the one the plain opening computes) is absorbed and returned like a computed
    one -- same challenges, same proof; a WRONG one changes the transcript (it is the caller's responsibility: the verifier rejects such a proof)"""
    ell = 7
    n = 1 << ell
    host_srs = O.srs_setup_from_secret(rand_fr(1, 91)[0], n + 1)
    srs = ctx.srs_upload(host_srs)
    tab, point = ctx.upload(rand_fr(n, 92)), np.stack([rand_challenge(93 + k) for k in range(ell)])
    plain = ctx.hyperkzg_open(srs, tab, point, label=4)
    for n_known in (1, 2, ell - 1):
        again = ctx.hyperkzg_open(srs, tab, point, label=4, known_levels=plain["com"][:n_known])
        assert np.array_equal(again["challenges"], plain["challenges"]) and np.array_equal(again["v"], plain["v"])
        for i in range(ell - 1):
            assert same_point(again["com"][i], plain["com"][i])
        for t in range(3):
            assert same_point(again["w"][t], plain["w"][t])
    wrong = ctx.hyperkzg_open(srs, tab, point, label=4, known_levels=plain["com"][1:2])
    assert not np.array_equal(wrong["challenges"], plain["challenges"])
    # ... but it must at least BE a point: coordinates off the curve or not canonical are refused before anything is absorbed
    off_curve = np.array(plain["com"][:1], dtype=np.uint64).copy()
    off_curve[0, 0] ^= np.uint64(1)
    not_canonical = np.array(plain["com"][:1], dtype=np.uint64).copy()
    not_canonical[0, :4] = np.uint64(2**64 - 1)
    for bad in (off_curve, not_canonical):
        with pytest.raises(ffi.JoltError) as e:
            ctx.hyperkzg_open(srs, tab, point, label=4, known_levels=bad)
        assert e.value.status == 1


@pytest.mark.parametrize("n_vars", [4, 6])
def test_workload_step_commit_and_open_bit_exact_with_oracle(ctx, n_vars):
    """The whole PCS side of DeviceWorkload.step at toy size: commitments of all 38 committed columns and the opening of the joint
    polynomial equal the oracle's kzg_commit / hyperkzg_open on the grid-embedded coefficient vectors, transcript byte for byte;
    the combined commitment (AdditivelyHomomorphic::combine, crates/jolt-hyperkzg/src/scheme.rs:346-352) is the joint one."""
    wl = DeviceWorkload(ctx, n_vars, seed=3, pcs="grid")
    T, K = 1 << n_vars, 16
    host_srs = O.srs_setup_from_secret(wl.beta, K * T)
    out = wl.step(label=40)
    cols = []
    for i in sorted(wl.sources):
        ms = wl.members_spec[i]
        cols += [wl.tables_spec[t].data for t in ms.tables[1:]]
    joint = np.zeros((K * T, 4), dtype=np.uint64)
    combined = O.g1_identity()
    for p, col in enumerate(cols):
        e = embed_onehot(col, K, T)
        assert same_point(out["commit"]["onehot"][p], O.kzg_commit(e, host_srs)), p
        joint = O.fr_add(joint, O.fr_mul(e, np.repeat(wl.rlc_onehot[p].reshape(1, 4), K * T, axis=0)))
        combined = O.g1_add(combined, O.g1_scalar_mul(out["commit"]["onehot"][p], wl.rlc_onehot[p]))
    for d, name in enumerate(wl.committed_dense):
        vals = O.fr_from_u64(wl.tables_spec[name].data)
        emb = np.zeros((K * T, 4), dtype=np.uint64)
        emb[:T] = vals
        assert same_point(out["commit"]["dense"][d], O.kzg_commit(emb, host_srs)), name
        joint[:T] = O.fr_add(joint[:T], O.fr_mul(vals, np.repeat(wl.rlc_dense[d].reshape(1, 4), T, axis=0)))
        combined = O.g1_add(combined, O.g1_scalar_mul(out["commit"]["dense"][d], wl.rlc_dense[d]))
    assert same_point(combined, O.kzg_commit(joint, host_srs))
    want = O.hyperkzg_open(host_srs, joint, wl.open_point, label=40)
    got = out["open"]
    assert np.array_equal(got["challenges"], want["challenges"]) and np.array_equal(got["v"], want["v"])
    for i in range(wl.grid_vars - 1):
        assert same_point(got["com"][i], want["com"][i])
    for t in range(3):
        assert same_point(got["w"][t], want["w"][t])
    again = wl.step(label=40)  # a second proof over rebuilt tables: same bytes
    assert np.array_equal(again["open"]["v"], got["v"])
    for st in out["stages"]:
        assert np.array_equal(again["stages"][st]["polys"], out["stages"][st]["polys"])
    wl.close()


@pytest.mark.parametrize("ell,kind", [(20, "full"), (22, "full"), (22, "u64")])
def test_hyperkzg_open_at_baseline_scale(ctx, ell, kind):
    """open() at 2^20 / 2^22 coefficients: the four pipelined MSM lanes, the recursive suffix scan over many chunk levels and the
    blocked Horner with u^(4096*block) weights are size-dependent paths -- checked through the beta-known identities."""
    n = 1 << ell
    beta = rand_fr(1, 500 + ell)[0]
    srs = ctx.srs_setup_from_secret(beta, n, G1_GENERATOR)
    if kind == "full":
        point = rand_fr(ell, 510)
        tab = ctx.eq_evals(rand_fr(ell, 511))  # full-width pseudo-random evaluations built on the device
    else:
        point = np.stack([rand_challenge(520 + k) for k in range(ell)])
        tab = ctx.from_u64(np.random.default_rng(521).integers(0, 2**64, size=n, dtype=np.uint64))
    proof = ctx.hyperkzg_open(srs, tab, point, label=77)
    claimed = ctx.evaluate(tab, point)
    p_beta = check_opening(ctx, tab, point, proof, beta, claimed)
    assert same_point(ctx.hyperkzg_commit(srs, tab), O.g1_scalar_mul(O.g1_generator(), p_beta))
    srs.free()
    tab.free()
    ctx.trim()


def test_bench_step_at_configs2_scale():
    """BASELINE configs[2] as bench.py runs it: T = 2^22 cycles, 2^26-coefficient commitment grid.  The step's commitments combine
    to the joint polynomial's commitment J(beta) G, the opening passes the beta-known identities with EVERY level taken from the oracle
    (the 2 GiB joint polynomial is downloaded once, folded and Horner-evaluated by the oracle; the claimed evaluation is the oracle's
    too), and two steps give identical bytes."""
    c = ffi.Context(0)
    wl = DeviceWorkload(c, 22, pcs="grid")
    out = wl.step(label=7)
    joint = wl.joint_polynomial()
    claimed = c.evaluate(joint, wl.open_point)
    j_beta = check_opening(c, joint, wl.open_point, out["open"], wl.beta, claimed, max_download_log=26)
    joint.free()
    combined = O.g1_identity()
    for p in range(out["commit"]["onehot"].shape[0]):
        combined = O.g1_add(combined, O.g1_scalar_mul(out["commit"]["onehot"][p], wl.rlc_onehot[p]))
    for d in range(out["commit"]["dense"].shape[0]):
        combined = O.g1_add(combined, O.g1_scalar_mul(out["commit"]["dense"][d], wl.rlc_dense[d]))
    assert same_point(combined, O.g1_scalar_mul(O.g1_generator(), j_beta))
    again = wl.step(label=7)
    assert np.array_equal(again["open"]["v"], out["open"]["v"]) and np.array_equal(again["open"]["challenges"], out["open"]["challenges"])
    for st in out["stages"]:
        assert np.array_equal(again["stages"][st]["polys"], out["stages"][st]["polys"])
    stats = c.memory_stats()
    assert stats["peak_bytes"] < 200 * 2**30  # fits one MI355X with room to spare
    # ... the MSM workspaces outside the pool included (jolt_ctx_workspace_stats: four lanes + the batch of short MSMs), and they are not nothing at this size
    assert stats["msm_lanes_bytes"] > 2**30 and stats["peak_bytes"] + stats["msm_lanes_bytes"] + stats["msm_batch_bytes"] < 240 * 2**30
    wl.close()
    c.close()


def test_pool_reuses_and_trims(ctx):
    ctx.trim()
    before = ctx.memory_stats()
    t = ctx.alloc(1 << 16)
    p0 = t.device_ptr()
    t.free()
    mid = ctx.memory_stats()
    assert mid["cached_bytes"] >= before["cached_bytes"] + (1 << 21)
    t2 = ctx.alloc(1 << 16)  # same size class: the cached block comes back
    assert t2.device_ptr() == p0
    assert np.all(t2.download() == 0)  # and is zeroed again by jolt_table_alloc
    t2.free()
    ctx.trim()
    assert ctx.memory_stats()["cached_bytes"] == 0


@pytest.mark.parametrize("ell", [1, 2, 7, 13])
def test_open_under_the_callers_transcript(ctx, ell):
    """jolt_host_hyperkzg_open_with_transcript (CommitmentScheme::open's `transcript: &mut impl Transcript`, crates/jolt-openings/src/schemes.rs:66-72): the caller's
    hook absorbs the level commitments, the 3 ell evaluations and the witness commitments and returns r, q, d_0.  Driven with the library's own test transcript from
    the outside it must give the proof of jolt_host_hyperkzg_open with the same label -- and the absorbed data must be exactly the proof's fields, in order."""
    beta = rand_fr(1, 900 + ell)[0]
    srs = ctx.srs_setup_from_secret(beta, 1 << ell, G1_GENERATOR)
    poly = ctx.upload(rand_fr(1 << ell, 901 + ell))
    point = rand_fr(ell, 902 + ell)
    want = ctx.hyperkzg_open(srs, poly, point, label=77)
    tr = ffi.HostTranscript(77)
    seen = {"points": [], "values": []}

    def absorb_points(pts):
        seen["points"].append(pts)
        for p in pts:
            tr.append_bytes(ffi.host_g1_serialize_compressed(p))

    def absorb_values(vals):
        seen["values"].append(vals)
        tr.append(vals)

    got = ctx.hyperkzg_open_with_transcript(srs, poly, point, absorb_points, absorb_values, tr.challenge)
    for key in ("v", "challenges"):
        assert np.array_equal(got[key], want[key]), key
    assert all(same_point(got["com"][i], want["com"][i]) for i in range(ell - 1)) and all(same_point(got["w"][t], want["w"][t]) for t in range(3))
    assert len(seen["values"]) == 1 and np.array_equal(seen["values"][0], got["v"].reshape(-1, 4))
    assert sum(p.shape[0] for p in seen["points"]) == (ell - 1) + 3
    tr.close()
    poly.free()
    srs.free()


def test_streamed_commitment_windows_match_kzg_commit(ctx):
    """StreamingCommitment for a KZG-type scheme through jolt_msm_g1_window (crates/jolt-openings/src/schemes.rs:167-222; the streaming commit kernels feed a column as
    row_width windows in coefficient order, crates/jolt-kernels/src/reference/commitment.rs:86-121): a polynomial fed as windows of mixed kinds -- field elements,
    u64, i64, i128, a run of zeros (feed_zeros: the offset moves, nothing is added), an empty window -- has the commitment kzg_commit gives the concatenation."""
    rng = np.random.default_rng(33)
    n = 1 << 9
    beta = rand_fr(1, 70)[0]
    host_srs = O.srs_setup_from_secret(beta, n)
    srs = ctx.srs_upload(host_srs)
    pieces, acc, off = [], None, 0
    fr_part = rand_fr(100, 71)
    u_part = rng.integers(0, 2**64, size=64, dtype=np.uint64)
    i_part = rng.integers(-2**63, 2**63, size=64, dtype=np.int64)
    big = [int(v) for v in rng.integers(-2**62, 2**62, size=60)]
    big[:4] = [2**127 - 1, -2**127, -1, 0]
    i128_part = np.array([[v & (2**64 - 1), (v >> 64) & (2**64 - 1)] for v in big], dtype=np.uint64)
    for values, kind, as_fr in ((fr_part, "fr", fr_part), (u_part, "u64", O.fr_from_u64(u_part)), (np.zeros((0, 4), dtype=np.uint64), "fr", np.zeros((0, 4), dtype=np.uint64)),
                                (i_part, "i64", O.fr_from_i64(i_part)), (None, "zeros", np.zeros((96, 4), dtype=np.uint64)),
                                (i128_part, "i128", O.to_mont([v % O.R_MOD for v in big])), (rand_fr(n - 384, 72), "fr", None)):
        if kind == "zeros":
            off += 96
            pieces.append(as_fr)
            continue
        acc = ctx.msm_window(srs, off, values, kind, acc)
        off += values.shape[0]
        pieces.append(values if as_fr is None else as_fr)
    poly = np.concatenate(pieces)
    assert off == n and poly.shape == (n, 4)
    assert same_point(acc, O.kzg_commit(poly, host_srs))
    # a clone of the partial commitment is a value: feeding the last window again from the saved point gives the same commitment
    with pytest.raises(ffi.JoltError) as e:
        ctx.msm_window(srs, n - 10, rand_fr(11, 73), "fr", None)
    assert e.value.status == 9  # JOLT_ERR_SRS_TOO_SMALL
