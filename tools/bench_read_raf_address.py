#!/usr/bin/env python3
"""Host time of the 128 read-RAF address rounds of one proof (jolt_host_read_raf_address_*), no device needed: all 42 tables present, the scan sums of a small
trace fed to every phase, a 0.8 ms pause between phases where the device would condense and scan.   bench_read_raf_address.py [threads ...]
The sums come from the CPU oracle (test infrastructure) -- this is a tool, not the product path."""
import os
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def one():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import numpy as np
    import oracle_lib as O
    from jolt_amd import ffi
    from lookup_table_fixture import all_table_rows
    idx, tab, raf = all_table_rows(7, 11)
    lists = ffi.lookup_suffix_lists()
    u = O.eq_evals(O.to_mont([1000 + 37 * i for i in range(7)]))
    gamma = O.to_mont([0xACE157EF])[0]
    raf_sums, suffix_sums = O.read_raf_phase_scan(idx, tab, raf, 42, u, 120, 128, lists)
    best = None
    for _ in range(5):
        spent = 0.0
        t0 = time.perf_counter(); state = ffi.HostReadRafAddress(gamma, np.ones(42, dtype=np.uint8)); tr = ffi.HostTranscript(5); spent += time.perf_counter() - t0
        claim = O.to_mont([5])[0]
        for phase in range(16):
            time.sleep(0.0008)
            t0 = time.perf_counter(); state.init_phase(phase, raf_sums, suffix_sums); claim, _, _ = state.prove_phase(claim, tr); spent += time.perf_counter() - t0
        t0 = time.perf_counter(); state.finish(); state.close(); spent += time.perf_counter() - t0
        best = spent if best is None else min(best, spent)
    print(f"JOLT_HOST_THREADS={os.environ.get('JOLT_HOST_THREADS', '(default)')}: {best * 1e3:.2f} ms of host time per proof (best of 5)")


if __name__ == "__main__":
    if os.environ.get("_RR_ADDRESS_CHILD"):
        one()
    else:
        for n in (sys.argv[1:] or ["1", "4", "8"]):
            subprocess.run([sys.executable, __file__], env=dict(os.environ, JOLT_HOST_THREADS=n, OMP_NUM_THREADS="1", _RR_ADDRESS_CHILD="1"))
