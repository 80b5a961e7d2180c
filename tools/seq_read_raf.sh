#!/bin/bash
# kernel sequence (start time, name, grid, duration) of the first three address phases of ONE read-RAF proof at T = 2^$1, after a warm-up proof:
# where a phase's 0.88 ms of wall time goes between its kernels (sort passes, gather, accumulate, fold), the read-back and the host rounds.
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p_seq
cat > /tmp/seq_rr.py <<PY
import sys
sys.path.insert(0, "/root/repo")
from jolt_amd import ffi, stages as S
ctx = ffi.Context(0)
e = S.DeviceExtended(ctx, int(sys.argv[1]))
e.instruction_read_raf(1); ctx.synchronize()
e.instruction_read_raf(2); ctx.synchronize()
PY
timeout 300 rocprofv3 --kernel-trace -d /tmp/p_seq -o s -- python /tmp/seq_rr.py ${1:-22} > /tmp/seq.txt 2>&1
f=$(find /tmp/p_seq -name "*.db" | head -1)
python /root/repo/profiles/kernel_sequence.py "$f" 2>/dev/null > /tmp/seq_all.txt
# the second proof = the second half of the launches; its first three phases
python - <<'PY'
rows = [l.rstrip("\n") for l in open("/tmp/seq_all.txt") if l.strip()]
half = rows[len(rows) // 2:]
starts = [i for i, l in enumerate(half) if "k_rr_accumulate" in l]
end = starts[3] if len(starts) > 3 else len(half)
print("# launches of the second proof up to its fourth scan (start us since the trace began, kernel, grid, duration)")
for l in half[:end + 1]:
    print(l)
print("# ... and its tail: cycle tables, cycle rounds, output claims")
for l in half[-60:]:
    print(l)
PY
