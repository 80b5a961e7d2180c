# rocprofv3 kernel trace of the default bench step (BASELINE configs[2]) -> gpurun_out/<tag>/bench_kernel_stats.txt
set -u
TAG=${1:-r02}
OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_step
timeout 600 rocprofv3 --kernel-trace -d /tmp/p_step -o b -- python /root/repo/bench.py --no-cpu-baseline --no-split --no-upload-rate --no-msm-roofline --steps 2 --warmup 1 > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
f=$(find /tmp/p_step -name "*.db" | head -1); [ -n "$f" ] && timeout 120 python /root/repo/profiles/summarize_rocprof.py "$f" > $OUT/bench_kernel_stats.txt
head -40 $OUT/bench_kernel_stats.txt | cut -c1-160
