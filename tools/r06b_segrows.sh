#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06b
mkdir -p "$OUT"; cd "$ROOT"
timeout 600 python -m pytest tests/test_gpu_read_raf.py -m gpu -x -q 2>&1 | tail -2
JOLT_RAF_SEG_ROWS=128 timeout 600 python -m pytest tests/test_gpu_read_raf.py tests/test_gpu_extended.py -m gpu -x -q 2>&1 | tail -2
: > "$OUT/raf_seg_rows_ab.txt"
for r in 1024 256 128 64 512 1024 128; do
  JOLT_RAF_SEG_ROWS=$r timeout 300 python tools/time_extended.py 22 2>/dev/null | grep "scan" | tail -1 | sed "s/^/seg rows $r: /" | tee -a "$OUT/raf_seg_rows_ab.txt"
done
