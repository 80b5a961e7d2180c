#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06b
mkdir -p "$OUT"; cd "$ROOT"
: > "$OUT/host_threads_ab.txt"
for t in 16 8 16 8 32; do
  JOLT_HOST_THREADS=$t timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-upload-rate --no-msm-roofline > "$OUT/bench_ht$t.json" 2> "$OUT/bench_ht$t.err"
  python -c "import json; d=json.loads(open('$OUT/bench_ht$t.json').read().strip().splitlines()[-1]); s=d['config']['ms_per_step_split']; print('host threads $t', d['ms_per_step'], 'read_raf alone', s['instruction_read_raf'], 'stages', s['stages_1_to_7_as_in_the_step'])" | tee -a "$OUT/host_threads_ab.txt"
done
