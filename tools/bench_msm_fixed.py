#!/usr/bin/env python3
"""Fixed-base (window-precomputed) MSM vs the per-window bucket method at several sizes / window widths.
usage: bench_msm_fixed.py <log_n> [window_bits ...]   -- prints one JSON line per measurement
       JOLT_BENCH_PREFIXES="20 22 24" adds prefix MSMs of those log-lengths over the same tables (per-window vs fixed-base)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from jolt_amd import ffi  # noqa: E402
from jolt_amd.workload import G1_GENERATOR, rand_fr  # noqa: E402


def timed(ctx, srs, tab, reps=3):
    ctx.msm(srs, tab)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.msm(srs, tab)
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    widths = [int(a) for a in sys.argv[2:]] or [0]
    n = 1 << log_n
    rng = np.random.default_rng(3)
    ctx = ffi.Context(0)
    beta = rand_fr(1, rng)[0]
    tab = ctx.eq_evals(rand_fr(log_n, rng))  # full-width pseudo-random scalars built on the device
    srs = ctx.srs_setup_from_secret(beta, n, G1_GENERATOR)
    ctx.synchronize()
    print(json.dumps({"what": "per-window", "n": n, "ms": round(timed(ctx, srs, tab), 3)}), flush=True)
    prefixes = [int(x) for x in os.environ.get("JOLT_BENCH_PREFIXES", "").split()]
    tabs = {lg: ctx.eq_evals(rand_fr(lg, rng)) for lg in prefixes}
    for lg, t in tabs.items():
        print(json.dumps({"what": "per-window prefix", "n": 1 << lg, "ms": round(timed(ctx, srs, t), 3)}), flush=True)
    srs.free()
    for c in widths:
        srs = ctx.srs_setup_from_secret(beta, n, G1_GENERATOR)
        t0 = time.perf_counter()
        ctx.srs_precompute_windows(srs, c, 1)
        pre_ms = (time.perf_counter() - t0) * 1e3
        print(json.dumps({"what": "fixed-base", "n": n, "window_bits": c, "precompute_ms": round(pre_ms, 1), "ms": round(timed(ctx, srs, tab), 3)}), flush=True)
        for lg, t in tabs.items():
            print(json.dumps({"what": "fixed-base prefix", "n": 1 << lg, "window_bits": c, "ms": round(timed(ctx, srs, t), 3)}), flush=True)
        srs.free()


if __name__ == "__main__":
    main()
