#!/bin/bash
# Round-5 call I: witness upload after pooling the extracted columns: parity tests that touch the rows path, then resident / pinned / overlapped bench runs on one box
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05i
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_workload.py tests/test_gpu_onehot.py tests/test_gpu_dory.py tests/test_gpu_small_scalar.py -q -m gpu -x -k "witness_upload or rows or onehot or dory or ints" --durations=5 ) > "$OUT/pytest.txt" 2>&1
echo "rc $?" >> "$OUT/pytest.txt"
tail -5 "$OUT/pytest.txt"
for M in resident upload-pinned upload-overlapped; do
  timeout 300 python bench.py --no-cpu-baseline --no-msm-roofline --no-upload-rate --witness $M > "$OUT/bench_$M.json" 2> "$OUT/bench_$M.err"
  python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench_$M.json") if l.startswith("{")][-1])
print("$M", d["ms_per_step"], d["value"], d["config"]["ms_per_step_split"].get("witness_upload"), (d["config"].get("witness") or {}).get("h2d_GBps"))
PY
done
timeout 300 python tools/upload_overlap_probe.py > "$OUT/probe.json" 2> "$OUT/probe.err"; cut -c1-600 "$OUT/probe.json"
