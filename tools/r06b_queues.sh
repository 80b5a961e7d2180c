#!/bin/bash
# per-stage concurrency (two contexts) against the number of hardware queues the runtime multiplexes the process's streams onto
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06b
mkdir -p "$OUT"
cd "$ROOT"
: > "$OUT/stage_concurrency_queues_ab.txt"
for q in 8 16 24 8 16 24; do
  GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-upload-rate --no-msm-roofline > "$OUT/bench_q$q.json" 2> "$OUT/bench_q$q.err"
  python -c "import json; d=json.loads(open('$OUT/bench_q$q.json').read().strip().splitlines()[-1]); s=d['config']['ms_per_step_split']; print('queues $q', d['ms_per_step'], 'stages', s['stages_1_to_7_as_in_the_step'], 'open', s['open'], 'commit', s['commit'])" | tee -a "$OUT/stage_concurrency_queues_ab.txt"
done
JOLT_STAGE_CONCURRENCY=0 GPU_MAX_HW_QUEUES=16 timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-upload-rate --no-msm-roofline > "$OUT/bench_q16_c0.json" 2> "$OUT/bench_q16_c0.err"
python -c "import json; d=json.loads(open('$OUT/bench_q16_c0.json').read().strip().splitlines()[-1]); s=d['config']['ms_per_step_split']; print('queues 16 no concurrency', d['ms_per_step'], 'open', s['open'])" | tee -a "$OUT/stage_concurrency_queues_ab.txt"
