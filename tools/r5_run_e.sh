#!/bin/bash
# Round-5 call E: base indices of the bucket lists staged through LDS (k_fx_buckets_ordered_staged) -- parity, A/B of the kernel, FETCH_SIZE of both variants
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05e
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_msm.py -q -m gpu -x --durations=5 ) > "$OUT/pytest_msm.txt" 2>&1
echo "rc $?" >> "$OUT/pytest_msm.txt"
tail -6 "$OUT/pytest_msm.txt"
for V in 0 1; do
  JOLT_FX_STAGE_IDX=$V timeout 300 python tools/msm_bucket_one.py 26 3 >> "$OUT/ab.jsonl" 2>> "$OUT/ab.err"
done
cat "$OUT/ab.jsonl"
cd /tmp
for V in 0 1; do
  rm -rf /tmp/p_f$V
  JOLT_FX_STAGE_IDX=$V timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f$V -o p -- python $ROOT/tools/msm_bucket_one.py 26 1 > /dev/null 2> "$OUT/pmc$V.err"
  f=$(find /tmp/p_f$V -name "*.db" | head -1)
  echo "# JOLT_FX_STAGE_IDX=$V" >> "$OUT/fetch.txt"
  python $ROOT/profiles/pmc_kernel_summary.py "$f" k_fx_buckets >> "$OUT/fetch.txt" 2>&1
  python $ROOT/profiles/pmc_kernel_summary.py "$f" k_fx_heavy_seg >> "$OUT/fetch.txt" 2>&1
done
cat "$OUT/fetch.txt" | cut -c1-160
