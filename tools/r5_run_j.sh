#!/bin/bash
# Round-5 call J: the whole GPU suite at the round's code state (normal exit path: the os._exit hook is gone), smoke, and the exit status of each
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05j
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
( time timeout 1700 python -m pytest tests -q -m gpu --durations=15 ) > "$OUT/pytest_gpu.txt" 2>&1
echo "pytest exit status $?" >> "$OUT/pytest_gpu.txt"
tail -25 "$OUT/pytest_gpu.txt"
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) | tee "$OUT/smoke.txt"
