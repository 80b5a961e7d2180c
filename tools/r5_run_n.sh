#!/bin/bash
# Round-5 call N: the capacity sort (no histogram pass for uniform scalars; exact passes as a device-side fallback): parity, then A/B of one MSM and of the step
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05n
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_msm.py -q -m gpu -x --durations=5 ) > "$OUT/pytest_msm.txt" 2>&1
echo "rc $?" >> "$OUT/pytest_msm.txt"
tail -8 "$OUT/pytest_msm.txt"
for V in 0 1 0 1; do
  JOLT_FX_CAPACITY=$V timeout 300 python tools/msm_bucket_one.py 26 3 >> "$OUT/ab.jsonl" 2>> "$OUT/ab.err"
done
cut -c1-230 "$OUT/ab.jsonl"; tail -2 "$OUT/ab.err"
