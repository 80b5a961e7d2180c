#!/bin/bash
# A/B of the number of hardware queues the HIP runtime multiplexes the streams onto (one box, back to back)
B="python bench.py --steps 8 --warmup 2 --no-split --no-cpu-baseline --no-upload-rate --no-msm-roofline"
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for cfg in "GPU_MAX_HW_QUEUES=4" "GPU_MAX_HW_QUEUES=6" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=12" "GPU_MAX_HW_QUEUES=16" "JOLT_X=0" "GPU_MAX_HW_QUEUES=4"; do
  echo "$cfg $(env $cfg $B 2>/dev/null | ms)"
done
