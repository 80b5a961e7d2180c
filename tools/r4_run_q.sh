#!/bin/bash
# Round-4 last call: the whole GPU suite + smoke at the final tree (tests with the oracle on one thread per physical core)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04q
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -q -m gpu --durations=12 > "$OUT/pytest_gpu.txt" 2>&1
tail -3 "$OUT/pytest_gpu.txt"
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) | tee "$OUT/smoke.txt"
