#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06b
mkdir -p "$OUT"; cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_onehot.py tests/test_gpu_onehot_wide.py tests/test_gpu_extended.py tests/test_gpu_workload.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_extended_t22.py -m gpu -x -q -k "pushforward" 2>&1 | tail -2
: > "$OUT/bool_lds_ab.txt"
for a in 1 0 1 0; do
  JOLT_BOOL_LDS=$a timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-upload-rate --no-msm-roofline > "$OUT/bench_boollds$a.json" 2>/dev/null
  python -c "import json; d=json.loads(open('$OUT/bench_boollds$a.json').read().strip().splitlines()[-1]); s=d['config']['ms_per_step_split']; print('JOLT_BOOL_LDS=$a step', d['ms_per_step'], 'booleanity_cycle alone', s['booleanity_cycle'], 'stages', s['stages_1_to_7_as_in_the_step'])" | tee -a "$OUT/bool_lds_ab.txt"
done
