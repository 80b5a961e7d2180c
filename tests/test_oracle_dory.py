"""Pins the oracle's Dory tier-1 restatement (oracle/g1.c orc_dory_*; crates/jolt-dory/src/streaming.rs:115-205,366-419) against
the oracle's own naive MSM on the same integers lifted to Fr -- the reference's tests for this path are commit/open/verify
round trips (no golden points), so the definition sum_j v_j G_j is what there is to pin."""
import numpy as np

import oracle_lib as O
from util import rand_fr

R = O.R_MOD


def _srs(n, seed):
    return O.srs_setup_from_secret(rand_fr(1, seed)[0], n)


def test_rows_u64_i64_i128_match_naive_msm():
    width, rows = 8, 3
    srs = _srs(width, 5)
    rng = np.random.default_rng(11)
    u = rng.integers(0, 2**63, size=rows * width, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=rows * width, dtype=np.uint64)
    u[3] = 0
    u[4] = np.uint64(2**64 - 1)
    got = O.dory_commit_rows(srs, u, "u64", width)
    for r in range(rows):
        want = O.g1_msm_naive(srs, O.fr_from_u64(u[r * width:(r + 1) * width]))
        assert O.g1_eq(got[r], want)
    s = rng.integers(-2**63, 2**63, size=rows * width, dtype=np.int64)
    s[0], s[1], s[2] = -2**63, 2**63 - 1, -1
    got = O.dory_commit_rows(srs, s, "i64", width)
    for r in range(rows):
        want = O.g1_msm_naive(srs, O.fr_from_i64(s[r * width:(r + 1) * width]))
        assert O.g1_eq(got[r], want)
    big = [int(rng.integers(0, 2**62)) * int(rng.integers(0, 2**62)) * (1 if k % 2 else -1) for k in range(rows * width)]
    big[0], big[1], big[2], big[3] = -2**127, 2**127 - 1, -1, 0
    packed = np.array([[v & (2**64 - 1), (v >> 64) & (2**64 - 1)] for v in big], dtype=np.uint64)
    got = O.dory_commit_rows(srs, packed, "i128", width)
    for r in range(rows):
        want = O.g1_msm_naive(srs, O.to_mont([v % R for v in big[r * width:(r + 1) * width]]))
        assert O.g1_eq(got[r], want)


def test_onehot_chunk_matches_zero_one_msm():
    width, k = 32, 4
    srs = _srs(width, 6)
    rng = np.random.default_rng(12)
    idx = rng.integers(0, 3, size=width).astype(np.uint8)  # row 3 is never hit -> identity
    idx[rng.random(width) < 0.3] = 0xFF
    got = O.dory_onehot_chunk(srs, idx, k)
    for row in range(k):
        want = O.g1_msm_naive(srs, O.fr_from_u64((idx == row).astype(np.uint64)))
        assert O.g1_eq(got[row], want)
    assert O.g1_eq(got[3], O.g1_identity())
