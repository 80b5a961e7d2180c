#!/bin/bash
# Round-5 call G: the overlapped witness upload (jolt_rows_upload_begin / _wait): parity test, then bench.py's default run with value_with_upload
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05g
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_workload.py -q -m gpu -x -k "witness_upload" --durations=5 ) > "$OUT/pytest.txt" 2>&1
echo "rc $?" >> "$OUT/pytest.txt"
tail -6 "$OUT/pytest.txt"
( time timeout 600 python bench.py --no-cpu-baseline --no-msm-roofline ) > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -3 "$OUT/bench.err"
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][-1])
print("bench", d["ms_per_step"], d["value"], "with upload", d.get("value_with_upload"), d["config"].get("witness_upload"))
PY
for M in upload-pinned upload-overlapped; do
  timeout 300 python bench.py --no-cpu-baseline --no-msm-roofline --no-upload-rate --witness $M > "$OUT/bench_$M.json" 2> "$OUT/bench_$M.err"
  python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench_$M.json") if l.startswith("{")][-1])
print("$M", d["ms_per_step"], d["value"], d["config"]["ms_per_step_split"].get("witness_upload"))
PY
done
