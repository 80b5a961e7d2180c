#!/bin/bash
# one against two operator contexts per stage (A/B), after the step-shaped tests
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06b
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_workload.py tests/test_gpu_pcs.py tests/test_gpu_extended.py -m gpu -x -q 2>&1 | tail -5
: > "$OUT/stage_contexts_ab.txt"
for k in 2 1 2 1; do
  JOLT_STAGE_CONTEXTS=$k timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-upload-rate --no-msm-roofline > "$OUT/bench_slots$k.json" 2> "$OUT/bench_slots$k.err"
  python -c "import json; d=json.loads(open('$OUT/bench_slots$k.json').read().strip().splitlines()[-1]); s=d['config']['ms_per_step_split']; print('operator contexts $k', d['ms_per_step'], 'stages', s['stages_1_to_7_as_in_the_step'], 'open', s['open'])" | tee -a "$OUT/stage_contexts_ab.txt"
done
