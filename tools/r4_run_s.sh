#!/bin/bash
# Round-4 call S: the first level commitment of the step's opening by linearity (class sums of the one-hot columns) -- parity, A/B, bench
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04s
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python -m pytest tests/test_gpu_pcs.py -q -m gpu -x --durations=4 > "$OUT/pytest_pcs.txt" 2>&1
tail -4 "$OUT/pytest_pcs.txt"
for cfg in "" "JOLT_OPEN_LINEAR_LEVELS=1" "JOLT_OPEN_LINEAR_LEVELS=3" "JOLT_OPEN_LEVEL1=0"; do
  echo "step opening [$cfg] $(env $cfg timeout 300 python tools/open_step.py 22 3 2>&1 | grep 'open ms')"
done | tee "$OUT/open_level1_ab.txt"
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("bench", d["ms_per_step"], d["config"].get("ms_per_step_split"))
PY
