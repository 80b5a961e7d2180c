#!/usr/bin/env python3
"""MSM / HyperKZG timing on the GPU (BASELINE configs[2] pieces): G1 MSM at several sizes (uniform 254-bit and
witness-like 64-bit scalars), HyperKZG commit and open.  Prints one JSON line per measurement."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from jolt_amd import ffi  # noqa: E402

P_TOP = 0x30644E72E131A029


def rand_fr(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    a[:, 3] %= np.uint64(P_TOP)
    return a


def main():
    max_log = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ctx = ffi.Context(0)
    g = np.zeros(12, dtype=np.uint64)
    one_q = [0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f]  # Montgomery 1 in Fq
    two_q = [(2 * sum(v << (64 * i) for i, v in enumerate(one_q)) % 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47 >> (64 * i)) & (2**64 - 1) for i in range(4)]
    g[0:4], g[4:8], g[8:12] = one_q, two_q, one_q  # generator (1, 2, 1)
    beta = rand_fr(1, 1)[0]
    t0 = time.perf_counter()
    srs = ctx.srs_setup_from_secret(beta, (1 << max_log) + 1, g)
    ctx.synchronize()
    print(json.dumps({"what": "srs_setup_from_secret", "n": (1 << max_log) + 1, "ms": round((time.perf_counter() - t0) * 1e3, 2)}), flush=True)
    for log_n in range(12, max_log + 1, 2):
        n = 1 << log_n
        for kind in ("uniform254", "u64"):
            if kind == "uniform254":
                tab = ctx.upload(rand_fr(n, 10 + log_n))
            else:
                tab = ctx.from_u64(np.random.default_rng(20 + log_n).integers(0, 2**64, size=n, dtype=np.uint64))
            ctx.msm(srs, tab)  # warm
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.msm(srs, tab)
            ms = (time.perf_counter() - t0) / reps * 1e3
            print(json.dumps({"what": "msm_g1", "n": n, "scalars": kind, "ms": round(ms, 3), "terms_per_s": round(n / ms * 1e3)}), flush=True)
    ell = max_log
    evals = ctx.upload(rand_fr(1 << ell, 99))
    point = rand_fr(ell, 98)
    point[:, 0] = 0
    point[:, 1] = 0
    point[:, 3] &= np.uint64((1 << 61) - 1)
    ctx.hyperkzg_commit(srs, evals)
    t0 = time.perf_counter()
    ctx.hyperkzg_commit(srs, evals)
    print(json.dumps({"what": "hyperkzg_commit", "n": 1 << ell, "ms": round((time.perf_counter() - t0) * 1e3, 2)}), flush=True)
    t0 = time.perf_counter()
    ctx.hyperkzg_open(srs, evals, point, label=1)
    print(json.dumps({"what": "hyperkzg_open", "n": 1 << ell, "ms": round((time.perf_counter() - t0) * 1e3, 2)}), flush=True)


if __name__ == "__main__":
    main()
