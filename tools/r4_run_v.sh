#!/bin/bash
# Round-4 call V: a GPU pytest run ends with pytest's exit status through os._exit (tests/conftest.py): summary printed, status 0
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04v
mkdir -p "$OUT"
cd "$ROOT"
timeout 100 python -m pytest tests/test_gpu_poly.py tests/test_gpu_msm.py -q -m gpu -x -k "not full_size and not baseline and not 2_20" > "$OUT/pytest.txt" 2>&1
echo "exit code $?" | tee -a "$OUT/pytest.txt"
tail -3 "$OUT/pytest.txt"
