#!/bin/bash
# Round-4 call L: scalar histogram with plain / ballot-aggregated LDS atomics, 2 / 3 / 4 MSM lanes -- on the step's own (structured) opening and on a uniform one, min of 3
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04l
mkdir -p "$OUT"
cd "$ROOT"
for cfg in "" "JOLT_FX_HIST_PLAIN=0" "JOLT_MSM_LANES=3" "JOLT_MSM_LANES=2" "" "JOLT_FX_HIST_PLAIN=0"; do
  echo "step opening [$cfg] $(env $cfg timeout 300 python tools/open_step.py 22 3 2>&1 | grep 'open ms')"
done | tee "$OUT/open_step_ab.txt"
for cfg in "" "JOLT_FX_HIST_PLAIN=0"; do
  echo "uniform opening [$cfg] $(env $cfg timeout 300 python tools/open_one.py 26 3 2>&1 | grep 'open ms')"
done | tee -a "$OUT/open_step_ab.txt"
