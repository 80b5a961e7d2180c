#!/bin/bash
# A/B of where the opening hint's class sums run (one box, back to back): background at commit (default), more hardware queues, main stream at commit, at the opening
B="python bench.py --steps 8 --warmup 2 --no-split --no-cpu-baseline --no-upload-rate --no-msm-roofline"
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for cfg in "JOLT_X=0" "GPU_MAX_HW_QUEUES=8" "JOLT_HINT_BACKGROUND=0" "JOLT_HINT_AT_COMMIT=0" "JOLT_OPEN_LINEAR_LEVELS=3" "JOLT_OPEN_LINEAR_LEVELS=4" "JOLT_OPEN_LINEAR_LEVELS=3 GPU_MAX_HW_QUEUES=8" "JOLT_OPEN_LINEAR_LEVELS=0" "JOLT_X=0"; do
  echo "$cfg $(env $cfg $B 2>/dev/null | ms)"
done
