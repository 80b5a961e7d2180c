#!/usr/bin/env python3
"""Sparse read-write matrix (RAM read/write checking) on the GPU: all log_t + log_k rounds at trace scale.
usage: bench_rw.py [log_t] [log_k]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from jolt_amd import ffi  # noqa: E402
from jolt_amd.workload import rand_fr  # noqa: E402


def main():
    log_t = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    log_k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    T, K = 1 << log_t, 1 << log_k
    rng = np.random.default_rng(1)
    addresses = rng.integers(0, K, size=T, dtype=np.uint64)
    addresses[rng.random(T) < 0.4] = np.uint64(0xFFFFFFFFFFFFFFFF)  # ~40 % of the cycles make no RAM access
    pre = rng.integers(0, 2**64, size=T, dtype=np.uint64)  # timing only: values need not be consistent
    post = rng.integers(0, 2**64, size=T, dtype=np.uint64)
    ctx = ffi.Context(0)
    inc, val_init = ctx.eq_evals(rand_fr(log_t, rng)), ctx.eq_evals(rand_fr(log_k, rng))
    tau, gamma = rand_fr(log_t, rng), rand_fr(1, rng)[0]
    chal = rand_fr(log_t + log_k, rng)
    chal[:, 0] = 0
    chal[:, 1] = 0
    chal[:, 3] &= np.uint64((1 << 61) - 1)
    for rep in range(3):
        t0 = time.perf_counter()
        m = ctx.rw_matrix(addresses, pre, post, inc, val_init, tau, gamma)
        ctx.synchronize()
        t1 = time.perf_counter()
        bind = None
        for r in range(log_t + log_k):
            m.prove_round(bind)
            bind = chal[r]
        m.finish(bind)
        m.final_values()
        t2 = time.perf_counter()
        entries = int((addresses != np.uint64(0xFFFFFFFFFFFFFFFF)).sum())
        m.free()
        print(json.dumps({"what": "ram_read_write (sparse matrix, all rounds)", "log_t": log_t, "log_k": log_k, "entries": entries,
                          "create_ms": round((t1 - t0) * 1e3, 2), "rounds_ms": round((t2 - t1) * 1e3, 2)}), flush=True)


if __name__ == "__main__":
    main()
