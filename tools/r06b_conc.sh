#!/bin/bash
# round 6, second session: booleanity_cycle + the per-stage concurrency of operators and catalogue (A/B)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06b
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests/test_gpu_extended.py tests/test_gpu_reference_transcript.py tests/test_gpu_workload.py tests/test_gpu_pcs.py -m gpu -x -q --durations=5 > "$OUT/pytest_conc.txt" 2>&1
tail -10 "$OUT/pytest_conc.txt"
for c in 1 0 1 0; do
  JOLT_STAGE_CONCURRENCY=$c timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-upload-rate --no-msm-roofline > "$OUT/bench_conc$c.json" 2> "$OUT/bench_conc$c.err"
  python -c "import json; d=json.loads(open('$OUT/bench_conc$c.json').read().strip().splitlines()[-1]); print('concurrency $c', d['ms_per_step'], d['config']['ms_per_step_split'])" | tee -a "$OUT/stage_concurrency_ab.txt"
done
