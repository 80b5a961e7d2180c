#!/bin/bash
# Round-4 call K: the mid table set's 13-window MSMs sorted from the scalars (5-byte entries) -- the 2^26 step check against the oracle, A/B on the step's opening
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04k
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_pcs.py tests/test_gpu_msm.py -q -m gpu -x -k "bench_step or baseline_scale or fixed_base or window_tables or batched" > "$OUT/pytest.txt" 2>&1
tail -3 "$OUT/pytest.txt"
for cfg in "" "JOLT_FX_SOA13=0" "" "JOLT_FX_SOA13=0"; do
  echo "[$cfg] $(env $cfg timeout 300 python tools/open_step.py 22 3 2>&1 | grep 'open ms')"
done | tee "$OUT/open_step_ab.txt"
