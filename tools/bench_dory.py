#!/usr/bin/env python3
"""Dory tier-1 (G1) row-commitment timing on the GPU: batches of integer rows of several
magnitudes and one-hot chunk commitments, inputs resident in HBM.  Prints one JSON line per measurement; with --cpu also times
the oracle (scalar C restatement, one core) on a bounded sample."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from jolt_amd import ffi  # noqa: E402
from tools.bench_msm import rand_fr  # noqa: E402


def main():
    log_t = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 22
    log_w = (log_t + 1) // 2 if log_t < 24 else 12
    width, count = 1 << log_w, 1 << log_t
    ctx = ffi.Context(0)
    g = np.zeros(12, dtype=np.uint64)
    one_q = [0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f]
    two_q = [0xa6ba871b8b1e1b3a, 0x14f1d651eb8e167b, 0xccdd46def0f28c58, 0x1c14ef83340fbe5e]
    g[0:4], g[4:8], g[8:12] = one_q, two_q, one_q
    srs = ctx.srs_setup_from_secret(rand_fr(1, 1)[0], width, g)
    rng = np.random.default_rng(7)
    cases = [("u8", rng.integers(0, 2**8, size=count, dtype=np.uint64), None),
             ("u32", rng.integers(0, 2**32, size=count, dtype=np.uint64), None),
             ("u64", rng.integers(0, 2**64, size=count, dtype=np.uint64), None),
             ("i64", rng.integers(-2**63, 2**63, size=count, dtype=np.int64), None)]
    lo = rng.integers(0, 2**64, size=count, dtype=np.uint64)
    hi = rng.integers(0, 2**64, size=count, dtype=np.uint64)
    cases.append(("i128", np.stack([lo, hi], axis=1), "i128"))
    for name, vals, kind in cases:
        ints = ctx.ints(vals, kind)
        ctx.dory_commit_rows(srs, ints, width)
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.dory_commit_rows(srs, ints, width)
        ms = (time.perf_counter() - t0) / reps * 1e3
        print(json.dumps({"what": "dory_commit_rows", "values": name, "count": count, "row_width": width, "rows": count // width, "ms": round(ms, 3),
                          "values_per_s": round(count / ms * 1e3)}), flush=True)
        ints.free()
    for k in (16, 255):
        idx = rng.integers(0, k, size=(1, count)).astype(np.uint8)
        idx[0, rng.random(count) < 0.4] = 0xFF
        oh = ctx.onehot(idx, k)
        ctx.dory_commit_onehot(srs, oh, 0, width)
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.dory_commit_onehot(srs, oh, 0, width)
        ms = (time.perf_counter() - t0) / reps * 1e3
        print(json.dumps({"what": "dory_commit_onehot", "k": k, "cycles": count, "chunk_width": width, "ms": round(ms, 3),
                          "cycles_per_s": round(count / ms * 1e3)}), flush=True)
        oh.free()
    if "--cpu" in sys.argv:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        host = srs.download()
        sample_rows = 2
        v = rng.integers(0, 2**64, size=sample_rows * width, dtype=np.uint64)
        t0 = time.perf_counter()
        O.dory_commit_rows(host, v, "u64", width)
        dt = time.perf_counter() - t0
        print(json.dumps({"what": "cpu_oracle_rows_u64", "kind": "port", "cores": 1, "sample": f"{sample_rows} rows x {width}", "values_per_s": round(sample_rows * width / dt)}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
