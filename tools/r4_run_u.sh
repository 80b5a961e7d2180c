#!/bin/bash
# Round-4 call U: does the interpreter exit cleanly after the sharded commit tests (a teardown abort was seen once after 33 passed tests)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04u
mkdir -p "$OUT"
cd "$ROOT"
timeout 150 python -m pytest tests/test_gpu_distributed.py -q -m gpu -x -k "commit_and_open and (range or cyclic)" > "$OUT/pytest.txt" 2>&1
echo "exit code $?" | tee -a "$OUT/pytest.txt"
tail -4 "$OUT/pytest.txt"
