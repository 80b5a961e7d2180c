#!/bin/bash
# Round-5 call P: rocprofv3's OWN --kernel-trace --stats summaries (csv) of the two roofline commands: the bind kernel (bench.py --roofline-only) and the bucket-sum
# kernel (tools/msm_bucket_one.py = what bench.py's roofline_msm leg runs)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05p
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_st1 /tmp/p_st2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_st1 -o bind -- python "$ROOT/bench.py" --roofline-only > "$OUT/bind_roofline_bench.json" 2> "$OUT/bind.err"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_st2 -o msm -- python "$ROOT/tools/msm_bucket_one.py" 26 3 > "$OUT/msm_bucket_one.json" 2> "$OUT/msm.err"
find /tmp/p_st1 /tmp/p_st2 -name "*stats*" | head
for f in $(find /tmp/p_st1 -name "*kernel_stats.csv"); do head -6 "$f" > "$OUT/bind_roofline_rocprofv3_kernel_stats.csv"; done
for f in $(find /tmp/p_st2 -name "*kernel_stats.csv"); do head -14 "$f" > "$OUT/msm_bucket_rocprofv3_kernel_stats.csv"; done
cat "$OUT/bind_roofline_rocprofv3_kernel_stats.csv" | cut -c1-220
cat "$OUT/msm_bucket_rocprofv3_kernel_stats.csv" | cut -c1-220
cut -c1-300 "$OUT/bind_roofline_bench.json"; cut -c1-300 "$OUT/msm_bucket_one.json"
