#!/usr/bin/env python3
"""Spartan outer T-scale work on the GPU at trace scale: uni-skip extended-node sums, Az / Bz materialisation, evaluation of all
inputs at one point.  usage: bench_r1cs.py [log_t] [n_inputs]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from jolt_amd import ffi  # noqa: E402
from jolt_amd.workload import rand_fr  # noqa: E402


def timed(ctx, fn, reps=5):
    fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    log_t = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    n_in = int(sys.argv[2]) if len(sys.argv) > 2 else 35
    T, nodes = 1 << log_t, 9
    rng = np.random.default_rng(2)
    ctx = ffi.Context(0)
    raw = [rng.integers(0, 2, size=T, dtype=np.uint64) if v % 3 else rng.integers(0, 2**64, size=T, dtype=np.uint64) for v in range(n_in)]  # flags and registers
    ints = [ctx.ints(r) for r in raw]
    inputs = [ctx.from_u64(r) for r in raw]
    eq = ctx.eq_evals(rand_fr(log_t + 1, rng))  # (cycle || stream) weights
    wa = rand_fr(nodes * 2 * (1 + n_in), rng).reshape(nodes * 2, 1 + n_in, 4)
    wb = rand_fr(nodes * 2 * (1 + n_in), rng).reshape(nodes * 2, 1 + n_in, 4)
    out = {"log_t": log_t, "inputs": n_in}
    out["uniskip_sums_ms"] = round(timed(ctx, lambda: ctx.r1cs_uniskip_sums(inputs, eq, wa, wb)), 3)

    def mat():
        az, bz = ctx.r1cs_materialize(inputs, wa[:2], wb[:2])
        az.free()
        bz.free()

    out["materialize_az_bz_ms"] = round(timed(ctx, mat), 3)
    point = rand_fr(log_t, rng)
    out["evaluate_all_inputs_ms"] = round(timed(ctx, lambda: ctx.tables_evaluate(inputs, point)), 3)
    out["input_bytes_gb"] = round(n_in * T * 32 / 1e9, 3)
    # the same three operators off the integer columns (integer column weights for the uni-skip sums: ~40 % non-zero)
    iwa = rng.integers(-2**20, 2**20, size=(nodes, 2, 1 + n_in)).astype(np.int64) * (rng.random((nodes, 2, 1 + n_in)) < 0.4)
    iwb = rng.integers(-2**40, 2**40, size=(nodes, 2, 1 + n_in)).astype(np.int64) * (rng.random((nodes, 2, 1 + n_in)) < 0.4)
    out["small_uniskip_sums_ms"] = round(timed(ctx, lambda: ctx.r1cs_uniskip_sums_small(ints, eq, iwa, iwb)), 3)

    def mat_small():
        az, bz = ctx.r1cs_materialize_small(ints, wa[:2], wb[:2])
        az.free()
        bz.free()

    out["small_materialize_az_bz_ms"] = round(timed(ctx, mat_small), 3)
    out["small_evaluate_all_inputs_ms"] = round(timed(ctx, lambda: ctx.ints_evaluate(ints, point)), 3)
    out["int_input_bytes_gb"] = round(n_in * T * 8 / 1e9, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
