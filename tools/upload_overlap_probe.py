#!/usr/bin/env python3
"""Where does the overlapped witness upload lose its time?  A 1 GB page-locked buffer copied (jolt_rows_upload_begin) alone and under a running 2^26-term MSM:
host time of the begin call, time until the copy has landed, and what the MSM pays for the company."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from jolt_amd import ffi  # noqa: E402
from jolt_amd.workload import G1_GENERATOR, rand_fr  # noqa: E402


def main():
    log_n = 26
    rng = np.random.default_rng(3)
    ctx = ffi.Context(0)
    tab = ctx.eq_evals(rand_fr(log_n, rng))
    srs = ctx.srs_setup_from_secret(rand_fr(1, rng)[0], 1 << log_n, G1_GENERATOR)
    ctx.srs_precompute_windows(srs, 0, 1)
    ctx.msm(srs, tab, full_width=True)
    pin = ffi.PinnedBuffer(ctx, (1 << 22, 240))
    pin.array[...] = 7
    out = {}

    def copy_alone():
        ctx.synchronize()
        t0 = time.perf_counter()
        r = ffi.Rows.begin(ctx, pin.array)
        t1 = time.perf_counter()
        r.wait()
        ctx.synchronize()
        t2 = time.perf_counter()
        r.free()
        return (t1 - t0) * 1e3, (t2 - t0) * 1e3

    out["copy_alone_ms(begin, landed)"] = [copy_alone() for _ in range(3)]
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.msm(srs, tab, full_width=True)
    out["msm_alone_ms"] = (time.perf_counter() - t0) * 1e3

    # the copy begun first, then the MSM (synchronous call: returns when its result is on the host), then wait for the copy
    res = []
    for _ in range(3):
        ctx.synchronize()
        t0 = time.perf_counter()
        r = ffi.Rows.begin(ctx, pin.array)
        t1 = time.perf_counter()
        ctx.msm(srs, tab, full_width=True)
        t2 = time.perf_counter()
        r.wait()
        ctx.synchronize()
        t3 = time.perf_counter()
        r.free()
        res.append({"begin_ms": round((t1 - t0) * 1e3, 3), "msm_ms": round((t2 - t1) * 1e3, 3), "copy_tail_after_msm_ms": round((t3 - t2) * 1e3, 3)})
    out["copy_under_msm"] = res
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
