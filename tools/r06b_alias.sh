#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06b
mkdir -p "$OUT"; cd "$ROOT"
timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_pcs.py tests/test_gpu_msm_fixed.py -m gpu -x -q 2>&1 | tail -3
: > "$OUT/fx_alias_keys_ab.txt"
for a in 1 0 1 0; do
  JOLT_FX_ALIAS_KEYS=$a timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-upload-rate --no-msm-roofline --no-split > "$OUT/bench_alias$a.json" 2> "$OUT/bench_alias$a.err"
  python -c "import json; d=json.loads(open('$OUT/bench_alias$a.json').read().strip().splitlines()[-1]); print('JOLT_FX_ALIAS_KEYS=$a', d['ms_per_step'], d['config']['device_pool_gib'])" | tee -a "$OUT/fx_alias_keys_ab.txt"
done
