#!/usr/bin/env python3
"""One fixed-base MSM of 2^log_n uniform scalars with the bucket-sum kernel bracketed by HIP events (jolt_msm_profile_buckets): the A/B and PMC target of the
round-5 index staging (JOLT_FX_STAGE_IDX=0 / 1, read once per process).   usage: msm_bucket_one.py [log_n = 26] [reps = 3]"""
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
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    n = 1 << log_n
    rng = np.random.default_rng(3)
    ctx = ffi.Context(0)
    beta = rand_fr(1, rng)[0]
    tab = ctx.eq_evals(rand_fr(log_n, rng))
    srs = ctx.srs_setup_from_secret(beta, n, G1_GENERATOR)
    ctx.srs_precompute_windows(srs, 0, 1)
    first = ctx.msm(srs, tab, full_width=True)
    ctx.msm_profile_buckets(True)
    ms_b, ms_all = [], []
    for _ in range(reps):
        ctx.synchronize()
        t0 = time.perf_counter()
        p = ctx.msm(srs, tab, full_width=True)
        ms_all.append((time.perf_counter() - t0) * 1e3)
        b, adds = ctx.msm_profile_buckets_last()
        ms_b.append(b)
        assert ffi.host_g1_eq(p, first)  # the same group element (the Jacobian coordinates depend on the order the sort's atomics left the lists in)
    print(json.dumps({"stage_idx": os.environ.get("JOLT_FX_STAGE_IDX", "1"), "capacity": os.environ.get("JOLT_FX_CAPACITY", "1"), "log_n": log_n, "additions": adds, "bucket_ms": [round(x, 3) for x in ms_b],
                      "msm_ms": [round(x, 3) for x in ms_all], "adds_per_s_G": round(adds / (min(ms_b) * 1e-3) / 1e9, 3),
                      "point": [int(x) for x in np.asarray(first).reshape(-1)[:4]]}), flush=True)


if __name__ == "__main__":
    main()
