#!/bin/bash
# Round-5 call C: the teardown fix -- the round-4 selection + the streamed-commitment test, leaving through the NORMAL exit path (no hook in conftest any more)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05c
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_workload.py tests/test_gpu_distributed.py tests/test_gpu_pcs.py -q -m gpu -x -k "not sharded_device_workload and not bench_step_at_configs2 and not baseline_scale" --durations=8 ) > "$OUT/pytest.txt" 2>&1
echo "rc $?" >> "$OUT/pytest.txt"
tail -20 "$OUT/pytest.txt"
