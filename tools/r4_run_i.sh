#!/bin/bash
# Round-4 call I: the benchmarked step's own opening (structured joint polynomial) under the timeline reader
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04i
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_os
timeout 500 rocprofv3 --kernel-trace -d /tmp/p_os -o o -- python "$ROOT/tools/open_step.py" 22 1 > "$OUT/open_step.txt" 2>&1
grep "open ms" "$OUT/open_step.txt"
f=$(find /tmp/p_os -name "*.db" | head -1)
python "$ROOT/profiles/open_exposed.py" "$f" 34 > "$OUT/open_exposed_step.txt" 2>&1
cat "$OUT/open_exposed_step.txt" | cut -c1-120
python "$ROOT/profiles/kernel_sequence.py" "$f" 400 > "$OUT/open_sequence_step.txt" 2>&1 || true
cd "$ROOT"
echo "[3 reps] $(timeout 300 python tools/open_step.py 22 3 2>&1 | grep 'open ms')"
