#!/bin/bash
# Round-5 call B (measurement only, no new code): (1) the teardown abort of round 4 hunted under rocgdb -- the same test selection, leaving through the NORMAL exit path
# with glibc's heap checks on; (2) SQ counters of the fixed-base MSM kernels (roofline_msm); (3) Dory tier-1 rows re-measured
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05b
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
python -c "import torch" 2>/dev/null
export JOLT_TEST_NORMAL_EXIT=1
( MALLOC_CHECK_=3 timeout 600 rocgdb -batch -ex "set pagination off" -ex "set confirm off" -ex "handle SIGABRT stop print" -ex "handle SIG32 SIG33 SIG34 SIG35 nostop noprint pass" -ex run -ex bt -ex "info sharedlibrary" \
    --args python -m pytest tests/test_gpu_workload.py tests/test_gpu_distributed.py -q -m gpu -x -k "not stage_operators and not sharded_device_workload" ) > "$OUT/gdb_pytest.txt" 2>&1
echo "rc $?" >> "$OUT/gdb_pytest.txt"
grep -n "passed\|failed\|double free\|SIGABRT\|#[0-9]" "$OUT/gdb_pytest.txt" | head -40
# the same once more without the debugger (exit status of the normal path)
( timeout 600 python -m pytest tests/test_gpu_workload.py tests/test_gpu_distributed.py -q -m gpu -x -k "not stage_operators and not sharded_device_workload" ) > "$OUT/pytest_normal_exit.txt" 2>&1
echo "rc $?" >> "$OUT/pytest_normal_exit.txt"
tail -4 "$OUT/pytest_normal_exit.txt"
bash tools/pmc_msm.sh r05b > "$OUT/pmc_msm.log" 2>&1
tail -12 "$OUT/pmc_msm.log"
timeout 300 python tools/bench_dory.py 22 > "$OUT/dory_tier1.txt" 2> "$OUT/dory.err"
cat "$OUT/dory_tier1.txt" | cut -c1-300
