#!/bin/bash
# Round-4: the multi-rank bench tests again after their banner check was made robust
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04r
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python -m pytest tests/test_gpu_distributed.py -q -m gpu -k "bench_multi_rank" > "$OUT/pytest.txt" 2>&1
tail -3 "$OUT/pytest.txt"
