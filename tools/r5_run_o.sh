#!/bin/bash
# Round-5 call O: the capacity sort inside the step: PCS / workload parity at the benchmarked sizes, then the step A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05o
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_pcs.py tests/test_gpu_workload.py -q -m gpu -x --durations=6 ) > "$OUT/pytest_pcs_workload.txt" 2>&1
echo "rc $?" >> "$OUT/pytest_pcs_workload.txt"
tail -12 "$OUT/pytest_pcs_workload.txt"
for V in 0 1 0 1; do
  JOLT_FX_CAPACITY=$V timeout 300 python bench.py --no-cpu-baseline --no-msm-roofline --no-upload-rate --steps 6 --warmup 2 > "$OUT/bench_cap$V.json" 2> "$OUT/bench_cap$V.err"
  python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench_cap$V.json") if l.startswith("{")][-1])
print("JOLT_FX_CAPACITY=$V", d["ms_per_step"], d["value"], d["config"]["ms_per_step_split"]["open"], d["config"]["ms_per_step_split"]["commit"])
PY
done
