#!/bin/bash
# Round-4 call C: cross-rank stage operators (2 / 4 ranks on one GPU against the global oracle), the all-nodes uni-skip kernel, then the default bench.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04c
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_distributed.py -q -m gpu -x --durations=8 -k "stage_operators or bench_multi_rank or sharded_device_workload" > "$OUT/pytest_dist.txt" 2>&1
tail -25 "$OUT/pytest_dist.txt"
timeout 600 python -m pytest tests/test_gpu_small_scalar.py tests/test_gpu_r1cs.py tests/test_gpu_product.py tests/test_gpu_rw.py tests/test_gpu_registers.py "tests/test_gpu_extended.py::test_extended_stages_match_oracle" -q -m gpu -x --durations=5 > "$OUT/pytest_ops.txt" 2>&1
tail -12 "$OUT/pytest_ops.txt"
for u in 1 0; do
  JOLT_UNISKIP_ALL=$u timeout 200 python tools/time_extended.py 22 > "$OUT/time_extended_uniskip$u.txt" 2>&1
  grep -E "spartan_outer|^ram|registers|booleanity" "$OUT/time_extended_uniskip$u.txt" | sed "s/^/uniskip_all=$u /"
done
timeout 400 python bench.py --steps 10 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("bench", d["ms_per_step"], d["value"], d["config"].get("ms_per_step_split"), d["roofline"]["frac"], d.get("roofline_sumcheck",{}).get("frac"), d["cpu_baseline"]["value"])
PY
