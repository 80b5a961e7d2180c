#!/usr/bin/env python3
"""Instruction read+RAF scans on the GPU at trace scale: all 16 phases (condense + RAF sums + suffix accumulators) and the cycle columns.
usage: bench_read_raf.py [log_t] [n_tables]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from jolt_amd import ffi  # noqa: E402
from jolt_amd.workload import rand_fr  # noqa: E402
from read_raf_fixture import suffix_lists  # noqa: E402  (table -> suffix-kind lists covering all 48 kinds)


def main():
    log_t = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    n_tables = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    T = 1 << log_t
    rng = np.random.default_rng(1)
    idx = np.frombuffer(rng.bytes(16 * T), dtype=np.uint64).reshape(T, 2)
    table = rng.integers(0, n_tables, size=T).astype(np.uint8)
    table[rng.random(T) < 0.1] = 0xFF
    raf = (rng.random(T) < 0.3).astype(np.uint8)
    lists = suffix_lists(n_tables, 2)
    ctx = ffi.Context(0)
    rr = ctx.read_raf(idx, table, raf, n_tables)
    u = ctx.eq_evals(rand_fr(log_t, rng))
    v = rand_fr(256, rng)
    out = {"log_t": log_t, "tables": n_tables, "suffixes": sum(len(l) for l in lists)}
    rr.phase_scan(u, 120, 128, lists)
    ctx.synchronize()
    t_scan = t_cond = 0.0
    for phase in range(16):
        suffix_len = 128 - 8 * (phase + 1)
        if phase:
            t0 = time.perf_counter()
            rr.condense(u, v, suffix_len + 8)
            ctx.synchronize()
            t_cond += time.perf_counter() - t0
        t0 = time.perf_counter()
        rr.phase_scan(u, suffix_len, 128, lists)
        t_scan += time.perf_counter() - t0
    out["sixteen_phase_scans_ms"] = round(t_scan * 1e3, 2)
    out["fifteen_condensations_ms"] = round(t_cond * 1e3, 2)
    tv, ri, rid = rand_fr(n_tables, rng), rand_fr(1, rng)[0], rand_fr(1, rng)[0]
    vt = rand_fr(16 * 256, rng)
    ctx.synchronize()
    t0 = time.perf_counter()
    combined, ra = rr.cycle_tables(tv, ri, rid, vt, 128, 4)
    ctx.synchronize()
    out["cycle_columns_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
