#!/bin/bash
# Round-4 call T: after the level commitments by linearity -- the workload / sharded commit tests and the driver's command
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04t
mkdir -p "$OUT"
cd "$ROOT"
timeout 500 python -m pytest tests/test_gpu_workload.py tests/test_gpu_distributed.py -q -m gpu -x -k "not stage_operators and not sharded_device_workload" > "$OUT/pytest.txt" 2>&1
tail -3 "$OUT/pytest.txt"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("bench", d["ms_per_step"], d["value"], d["config"].get("ms_per_step_split"), d["roofline"]["frac"])
PY
