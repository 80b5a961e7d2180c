# Refreshes the measurements quoted in DESIGN.md (run on the GPU box; outputs under gpurun_out/final/)
set -u
OUT=/root/repo/gpurun_out/final; mkdir -p $OUT
cd /root/repo
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python bench.py --scale 22 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_T22.json 2>> $OUT/bench.err
timeout 300 python bench.py --scale 24 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_T24.json 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d /tmp/p1 -o b -- python /root/repo/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_under_rocprof.json 2>/dev/null
f=$(find /tmp/p1 -name "*.db" | head -1); [ -n "$f" ] && timeout 60 python /root/repo/profiles/summarize_rocprof.py "$f" > $OUT/bench_kernel_stats.txt
timeout 300 rocprofv3 --kernel-trace -d /tmp/p2 -o r -- python /root/repo/bench.py --roofline-only > $OUT/roofline_bench.json 2>/dev/null
f=$(find /tmp/p2 -name "*.db" | head -1); [ -n "$f" ] && timeout 60 python /root/repo/profiles/summarize_rocprof.py "$f" > $OUT/roofline_kernel_stats.txt
cd /root/repo
timeout 100 python tools/stage_times.py > $OUT/stage_times.txt 2>&1
tail -c 600 $OUT/bench.json; echo; cat $OUT/bench_T22.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('T22', d['ms_per_step'], d['value'])"; cat $OUT/bench_T24.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('T24', d['ms_per_step'], d['value'])"; head -8 $OUT/roofline_kernel_stats.txt | cut -c1-150; tail -2 $OUT/stage_times.txt | head -1
