#!/bin/bash
# A/B of JOLT_MSM_CU_SPLIT (sort phases and bucket sums of the fixed-base MSM on disjoint CU sets): bash tools/r3_cu_split.sh <tag>
set -u
TAG=${1:-r3d}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
JOLT_MSM_CU_SPLIT=1 timeout 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_pcs.py -q -x -m gpu > "$OUT/pytest_split.txt" 2>&1; tail -2 "$OUT/pytest_split.txt"
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-split --steps 6 --warmup 2 --stages 2-6b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'])"; }
run JOLT_X=0 | tee "$OUT/ab.txt"
for k in 1 2 3; do
  run JOLT_MSM_CU_SPLIT=$k | tee -a "$OUT/ab.txt"
  run JOLT_MSM_CU_SPLIT=$k JOLT_MSM_STAGGER=1 | tee -a "$OUT/ab.txt"
done
run JOLT_MSM_CU_SPLIT=1 JOLT_MSM_LANES=2 | tee -a "$OUT/ab.txt"
