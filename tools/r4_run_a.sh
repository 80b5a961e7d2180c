#!/bin/bash
# Round-4 call A: the new T = 2^22 per-operator parity tests, the 2^20 run with its sampled from-the-definition phases, and the suites whose code the ADVICE fixes touched.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04a
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests/test_gpu_extended_t22.py tests/test_gpu_extended.py tests/test_gpu_rw.py tests/test_gpu_registers.py tests/test_gpu_read_raf.py -q -m gpu --durations=20 > "$OUT/pytest_ext.txt" 2>&1
tail -30 "$OUT/pytest_ext.txt"
