#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06b
mkdir -p "$OUT"; cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_read_raf.py tests/test_gpu_extended.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_extended_t22.py -m gpu -x -q -k "read_raf" 2>&1 | tail -2
: > "$OUT/raf_group_ab.txt"
for g in 16 64 16 64; do
  JOLT_RAF_GROUP=$g timeout 300 python tools/time_extended.py 22 2>/dev/null | grep "scan" | tail -1 | sed "s/^/lanes per item $g: /" | tee -a "$OUT/raf_group_ab.txt"
done
for g in 16 64 16 64; do
  JOLT_RAF_GROUP=$g timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-upload-rate --no-msm-roofline > "$OUT/bench_group$g.json" 2>/dev/null
  python -c "import json; d=json.loads(open('$OUT/bench_group$g.json').read().strip().splitlines()[-1]); s=d['config']['ms_per_step_split']; print('lanes per item $g: step', d['ms_per_step'], 'read_raf alone', s['instruction_read_raf'], 'stages', s['stages_1_to_7_as_in_the_step'])" | tee -a "$OUT/raf_group_ab.txt"
done
