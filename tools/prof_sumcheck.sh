#!/bin/bash
# the sumcheck legs alone (BASELINE configs[1] shape): per-stage host timings and the per-kernel table with all kernels of a round on one stream
# bash tools/prof_sumcheck.sh <log_t> <tag>
set -u
LOGT=${1:-22}; TAG=${2:-r3s}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd "$ROOT"
python tools/stage_times.py $LOGT > "$OUT/stage_times_$LOGT.txt" 2>&1; cat "$OUT/stage_times_$LOGT.txt"
python bench.py --no-msm --stages 2-6b --scale $LOGT --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no-msm', d['ms_per_step'], d['config']['ms_per_step_split'])" | tee "$OUT/bench_nomsm_$LOGT.txt"
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p_sc
JOLT_SERIAL_STREAMS=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/p_sc -o s -- python "$ROOT/bench.py" --no-msm --stages 2-6b --scale $LOGT --steps 3 --warmup 1 --no-cpu-baseline --no-split > /dev/null 2>&1
f=$(find /tmp/p_sc -name "*.db" | head -1); [ -n "$f" ] && python "$ROOT/profiles/summarize_rocprof.py" "$f" > "$OUT/sumcheck_kernel_stats_$LOGT.txt"
head -34 "$OUT/sumcheck_kernel_stats_$LOGT.txt" | cut -c1-160
