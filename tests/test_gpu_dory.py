"""GPU parity: Dory tier-1 (G1) streaming row commitments through the C ABI vs the CPU oracle (SURVEY.md section 8(f) row 2;
crates/jolt-dory/src/streaming.rs feed_u64 / feed_i128 / feed_i128_rows_with and one_hot_chunk_commitments).  Points are
compared as group elements."""
import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import ffi
from util import rand_fr

pytestmark = pytest.mark.gpu
R = O.R_MOD


@pytest.fixture(scope="module")
def ctx():
    c = ffi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def srs(ctx):
    beta = rand_fr(1, 91)[0]
    dev = ctx.srs_setup_from_secret(beta, 1 << 12, O.g1_generator())
    return beta, dev.download(), dev


def _pack_i128(vals):
    return np.array([[v & (2**64 - 1), (v >> 64) & (2**64 - 1)] for v in vals], dtype=np.uint64).reshape(-1, 2)


def _eval_point(beta, ints):
    """(sum_j v_j beta^j) * G: the commitment to a row over the bases beta^j G, from field arithmetic alone."""
    return O.g1_scalar_mul(O.g1_generator(), O.kzg_eval_univariate(O.to_mont([v % R for v in ints]), beta))


@pytest.mark.parametrize("width", [8, 64, 256])
def test_rows_match_oracle_all_kinds(ctx, srs, width):
    """width 8: several rows share a wavefront; 64 / 256: one and four wavefronts per row."""
    _, host, dev = srs
    rows = 5
    rng = np.random.default_rng(width)
    u = rng.integers(0, 2**63, size=rows * width, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    u[:width] = 0            # an all-zero row between non-zero rows -> identity
    u[width + 1] = 2**64 - 1
    u[2 * width:3 * width] = rng.integers(0, 256, size=width, dtype=np.uint64)  # byte-sized row
    got = ctx.dory_commit_rows(dev, ctx.ints(u), width)
    want = O.dory_commit_rows(host[:width], u, "u64", width)
    assert all(O.g1_eq(got[r], want[r]) for r in range(rows))
    assert O.g1_eq(got[0], O.g1_identity())
    s = rng.integers(-2**63, 2**63, size=rows * width, dtype=np.int64)
    s[0], s[1], s[2], s[3] = -2**63, 2**63 - 1, -1, 0
    got = ctx.dory_commit_rows(dev, ctx.ints(s), width)
    want = O.dory_commit_rows(host[:width], s, "i64", width)
    assert all(O.g1_eq(got[r], want[r]) for r in range(rows))
    big = [int(rng.integers(0, 2**63)) * int(rng.integers(0, 2**63)) * (1 if k % 3 else -1) for k in range(rows * width)]
    big[0], big[1], big[2], big[3] = -2**127, 2**127 - 1, -1, 0
    got = ctx.dory_commit_rows(dev, ctx.ints(big, "i128"), width)
    want = O.dory_commit_rows(host[:width], _pack_i128(big), "i128", width)
    assert all(O.g1_eq(got[r], want[r]) for r in range(rows))


def test_rows_small_magnitudes_and_all_zero(ctx, srs):
    """Only the windows the largest magnitude needs are processed: 1-bit, 9-bit and zero batches."""
    _, host, dev = srs
    width = 128
    for hi in (2, 300):
        v = np.random.default_rng(hi).integers(0, hi, size=3 * width, dtype=np.uint64)
        got = ctx.dory_commit_rows(dev, ctx.ints(v), width)
        want = O.dory_commit_rows(host[:width], v, "u64", width)
        assert all(O.g1_eq(got[r], want[r]) for r in range(3))
    got = ctx.dory_commit_rows(dev, ctx.ints(np.zeros(2 * width, dtype=np.uint64)), width)
    assert all(O.g1_eq(got[r], O.g1_identity()) for r in range(2))


def test_rows_heavy_buckets(ctx, srs):
    """Every value of a 4096-wide row equal (one bucket holds the whole row -> the segmented heavy path), both signs."""
    beta, host, dev = srs
    width = 1 << 12
    v = np.ones(2 * width, dtype=np.int64)
    v[width:] = -7
    got = ctx.dory_commit_rows(dev, ctx.ints(v), width)
    assert O.g1_eq(got[0], _eval_point(beta, [1] * width))
    assert O.g1_eq(got[1], _eval_point(beta, [-7] * width))


def test_rows_identity_at_scale(ctx, srs):
    """Size-independent property: over the bases beta^j G the row commitment is p_row(beta) G (64 rows x 4096 columns, u64 and
    i128 values), and the batch equals the concatenation of its halves."""
    beta, _, dev = srs
    width, rows = 1 << 12, 64
    rng = np.random.default_rng(3)
    u = rng.integers(0, 2**63, size=rows * width, dtype=np.uint64)
    ints = ctx.ints(u)
    got = ctx.dory_commit_rows(dev, ints, width)
    for r in (0, 17, 63):
        assert O.g1_eq(got[r], _eval_point(beta, [int(x) for x in u[r * width:(r + 1) * width]]))
    half = ctx.dory_commit_rows(dev, ctx.ints(u[: rows * width // 2]), width)
    assert all(O.g1_eq(half[r], got[r]) for r in range(rows // 2))
    big = [int(a) * int(b) - int(c) * 2**100 for a, b, c in zip(u[:4 * width], u[4 * width:8 * width], u[8 * width:12 * width] % np.uint64(2**20))]
    got = ctx.dory_commit_rows(dev, ctx.ints(big, "i128"), width)
    for r in (0, 3):
        assert O.g1_eq(got[r], _eval_point(beta, big[r * width:(r + 1) * width]))


def test_rows_wide_rows(ctx):
    """8192-column rows (256 buckets per window: the per-window fold kernel), mixed magnitudes, against p_row(beta) G."""
    beta = rand_fr(1, 92)[0]
    width = 1 << 13
    dev = ctx.srs_setup_from_secret(beta, width, O.g1_generator())
    rng = np.random.default_rng(13)
    v = rng.integers(-2**63, 2**63, size=3 * width, dtype=np.int64)
    v[width:2 * width] = rng.integers(-3, 4, size=width)
    got = ctx.dory_commit_rows(dev, ctx.ints(v), width)
    for r in range(3):
        assert O.g1_eq(got[r], _eval_point(beta, [int(x) for x in v[r * width:(r + 1) * width]]))


def test_rows_multiple_batches(ctx, srs):
    """2^22 i128 values as 1024 rows x 4096: more keys than one workspace batch holds (the rows are committed in several passes);
    first, middle and last row against p_row(beta) G, and equal to the same rows committed on their own."""
    beta, _, dev = srs
    width, rows = 1 << 12, 1 << 10
    rng = np.random.default_rng(17)
    lo = rng.integers(0, 2**64, size=rows * width, dtype=np.uint64)
    hi = rng.integers(0, 2**64, size=rows * width, dtype=np.uint64)
    packed = np.stack([lo, hi], axis=1)
    got = ctx.dory_commit_rows(dev, ctx.ints(packed, "i128"), width)
    assert got.shape[0] == rows

    def as_int(r, j):
        v = int(lo[r * width + j]) | (int(hi[r * width + j]) << 64)
        return v - (1 << 128) if v >> 127 else v
    for r in (0, 511, 1023):
        assert O.g1_eq(got[r], _eval_point(beta, [as_int(r, j) for j in range(width)])), r
    tail = ctx.dory_commit_rows(dev, ctx.ints(packed[(rows - 2) * width:], "i128"), width)
    assert O.g1_eq(tail[0], got[rows - 2]) and O.g1_eq(tail[1], got[rows - 1])


def test_rows_argument_checks(ctx, srs):
    _, _, dev = srs
    v = ctx.ints(np.arange(96, dtype=np.uint64))
    with pytest.raises(ffi.JoltError) as e:  # streaming.rs:99-102 row width must be a power of two
        ctx.dory_commit_rows(dev, v, 48)
    assert e.value.status == 1
    with pytest.raises(ffi.JoltError) as e:  # :192-195 batch length must be a multiple of the row width
        ctx.dory_commit_rows(dev, v, 64)
    assert e.value.status == 5
    with pytest.raises(ffi.JoltError) as e:  # :103-108 row width exceeds the SRS
        ctx.dory_commit_rows(dev, ctx.ints(np.zeros(1 << 13, dtype=np.uint64)), 1 << 13)
    assert e.value.status == 9
    assert ctx.dory_commit_rows(dev, ctx.ints(np.zeros(0, dtype=np.uint64)), 64).shape[0] == 0


@pytest.mark.parametrize("chunk_width", [16, 256])
def test_onehot_chunks_match_oracle(ctx, srs, chunk_width):
    _, host, dev = srs
    k, cycles, n_polys = 16, 1024, 3
    rng = np.random.default_rng(chunk_width)
    idx = rng.integers(0, k - 1, size=(n_polys, cycles)).astype(np.uint8)  # row k-1 is never hit -> identity
    idx[rng.random((n_polys, cycles)) < 0.4] = 0xFF
    idx[1, :chunk_width] = 0xFF  # a chunk with no hot column at all
    oh = ctx.onehot(idx, k)
    for poly in (0, 1, 2):
        got = ctx.dory_commit_onehot(dev, oh, poly, chunk_width)
        assert got.shape[:2] == (cycles // chunk_width, k)
        for ch in range(cycles // chunk_width):
            want = O.dory_onehot_chunk(host[:chunk_width], idx[poly, ch * chunk_width:(ch + 1) * chunk_width], k)
            assert all(O.g1_eq(got[ch, row], want[row]) for row in range(k)), (poly, ch)
        assert all(O.g1_eq(got[ch, k - 1], O.g1_identity()) for ch in range(cycles // chunk_width))
    oh.free()


def test_onehot_identity_at_scale_and_skew(ctx, srs):
    """2^16 cycles in 4096-column chunks, K = 16 and K = 255: commitment[row] = (sum of beta^col over the row's columns) G;
    one chunk has every column on the same row (a 4096-point bucket)."""
    beta, _, dev = srs
    width, cycles = 1 << 12, 1 << 16
    for k in (16, 255):
        rng = np.random.default_rng(k)
        idx = rng.integers(0, k, size=(1, cycles)).astype(np.uint8)
        idx[0, rng.random(cycles) < 0.4] = 0xFF
        idx[0, 2 * width:3 * width] = 5
        oh = ctx.onehot(idx, k)
        got = ctx.dory_commit_onehot(dev, oh, 0, width)
        for ch, row in ((0, 0), (2, 5), (2, 4), (15, k - 1)):
            col = idx[0, ch * width:(ch + 1) * width]
            assert O.g1_eq(got[ch, row], _eval_point(beta, [int(x) for x in (col == row)])), (k, ch, row)
        oh.free()
    with pytest.raises(ffi.JoltError) as e:
        ctx.dory_commit_onehot(dev, ctx.onehot(np.zeros((1, 96), dtype=np.uint8), 16), 0, 48)
    assert e.value.status == 1
