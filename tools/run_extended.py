#!/usr/bin/env python3
"""The stage 1 / 2 / 4 / 5 / 6a / 6b / 7 operators of the step, PROOFS times after one warm-up (the command the rocprofv3 passes of tools/pmc_extended.sh wrap):
run_extended.py [log_t] [proofs].  Prints the wall time of the last proof."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from jolt_amd import ffi  # noqa: E402
from jolt_amd import stages as S  # noqa: E402


def main():
    log_t = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    proofs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    ctx = ffi.Context(0)
    e = S.DeviceExtended(ctx, log_t)
    e.prove(label=3)
    ctx.synchronize()
    for _ in range(proofs):
        t0 = time.perf_counter()
        e.prove(label=3)
        ctx.synchronize()
        dt = time.perf_counter() - t0
    print(f"stage operators, last proof: {dt * 1e3:.2f} ms", flush=True)
    e.close()
    ctx.close()


if __name__ == "__main__":
    main()
