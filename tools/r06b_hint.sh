#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06b
mkdir -p "$OUT"; cd "$ROOT"
: > "$OUT/hint_under_concurrent_stages_ab.txt"
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-upload-rate --no-msm-roofline > "$OUT/bench_hint_$name.json" 2>/dev/null
  python -c "import json; d=json.loads(open('$OUT/bench_hint_$name.json').read().strip().splitlines()[-1]); s=d['config']['ms_per_step_split']; print('$name', d['ms_per_step'], 'commit', s['commit'], 'stages', s.get('stages_1_to_7_as_in_the_step'), 'open', s['open'])" | tee -a "$OUT/hint_under_concurrent_stages_ab.txt"
}
for rep in 1 2; do
  run background_default JOLT_X=1
  run main_stream_at_commit JOLT_HINT_BACKGROUND=0
  run at_the_opening JOLT_HINT_AT_COMMIT=0
  run three_levels JOLT_OPEN_LINEAR_LEVELS=3
done
