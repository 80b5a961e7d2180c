#!/bin/bash
# Round-4 call O: the extended-stage tests with the trimmed from-the-definition sample (timing of the heaviest GPU tests)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04o
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests/test_gpu_extended.py tests/test_gpu_extended_t22.py -q -m gpu -x --durations=10 > "$OUT/pytest_extended.txt" 2>&1
tail -14 "$OUT/pytest_extended.txt"
