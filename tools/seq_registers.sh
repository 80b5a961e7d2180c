#!/bin/bash
# kernel sequence of ONE registers (or RAM: $2 = ram) read/write proof at T = 2^$1 (after a warm-up proof): per-launch durations in launch order
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p_seq
cat > /tmp/seq_one.py <<PY
import sys
sys.path.insert(0, "/root/repo")
from jolt_amd import ffi, stages as S
ctx = ffi.Context(0)
e = S.DeviceExtended(ctx, int(sys.argv[1]))
fn = e.ram_read_write if sys.argv[2] == "ram" else e.registers_read_write
fn(1); ctx.synchronize()
fn(2); ctx.synchronize()
PY
timeout 300 rocprofv3 --kernel-trace -d /tmp/p_seq -o s -- python /tmp/seq_one.py ${1:-22} ${2:-registers} > /tmp/seq.txt 2>&1
f=$(find /tmp/p_seq -name "*.db" | head -1); python /root/repo/profiles/kernel_sequence.py "$f" 2>/dev/null | grep -E "k_rw|k_reg|k_scan|copy|fill" | tail -${3:-400}
