#!/bin/bash
# Round-4 call J: joint polynomial by rows, the h(r^2) MSM begun under the pair quotient's scan -- parity tests, A/B on the step's own opening (min of 3), bench
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04j
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_pcs.py -q -m gpu -x --durations=4 > "$OUT/pytest_msm_pcs.txt" 2>&1
tail -4 "$OUT/pytest_msm_pcs.txt"
timeout 900 python -m pytest tests/test_gpu_distributed.py -q -m gpu -x -k "open or pcs or subtree or commit" > "$OUT/pytest_dist.txt" 2>&1
tail -2 "$OUT/pytest_dist.txt"
for cfg in "" "JOLT_JOINT_ROWS=0" "JOLT_KZG_EARLY=0" "JOLT_MSM_PAIR_OVERLAP=0"; do
  echo "[$cfg] $(env $cfg timeout 300 python tools/open_step.py 22 3 2>&1 | grep 'open ms')"
done | tee "$OUT/open_step_ab.txt"
timeout 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("bench", d["ms_per_step"], d["value"], d["config"].get("ms_per_step_split"))
PY
