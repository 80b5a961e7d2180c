#!/usr/bin/env python3
"""A/B of the two sparse read-write operators alone at T = 2^22 (a process per setting: the knobs are read once): ab_rw.py"""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CHILD = r'''
import sys, time
sys.path.insert(0, %r)
from jolt_amd import ffi, stages as S
ctx = ffi.Context(0)
e = S.DeviceExtended(ctx, 22)
out = []
for name, fn in (("ram", lambda: e.ram_read_write(8)), ("registers", lambda: e.registers_read_write(9))):
    fn(); ctx.synchronize()
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); fn(); ctx.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
    out.append(f"{name} {best:.2f}")
print(" ".join(out))
''' % ROOT
for cfg in sys.argv[1:] or ["JOLT_RW_GRID_MULT=4", "JOLT_RW_GRID_MULT=8", "JOLT_RW_GRID_MULT=16", "JOLT_RW_GRID_MULT=32"]:
    env = dict(os.environ)
    for kv in cfg.split():
        k, v = kv.split("=")
        env[k] = v
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(cfg, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
