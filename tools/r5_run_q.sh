#!/bin/bash
# Round-5 call Q: the bucket kernels' occupancy attribute with the staged-index kernel: -DJOLT_BUCKET_WAVES=2 / 4 against the default 3 (rebuilt ON THE GPU BOX)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
mkdir -p gpurun_out/r05q
for W in 2 4 3; do
  touch jolt_amd/csrc/msm_kernels.hip.h
  JOLT_EXTRA_HIPCC_FLAGS="-DJOLT_BUCKET_WAVES=$W" python -m jolt_amd.build > /dev/null 2>&1 || { echo "build failed for $W"; continue; }
  echo "waves $W: $(timeout 300 python tools/msm_bucket_one.py 26 3 2>/dev/null | cut -c1-200)" | tee -a gpurun_out/r05q/waves.txt
done
