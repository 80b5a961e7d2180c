#!/bin/bash
# Everything the round's measurements in DESIGN.md section 4 come from, in one GPU-box call:
#   bash tools/collect_round.sh r02 [notests]
# writes gpurun_out/<tag>/...; the files to keep are then copied into profiles/ as <tag>_* (tools/keep_round.sh).
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
if [ "${2:-}" != "notests" ]; then
  timeout 1800 python -m pytest tests -q -m gpu --durations=12 > "$OUT/pytest_gpu.txt" 2>&1
  tail -3 "$OUT/pytest_gpu.txt"
fi
# the driver's command (default flags)
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
cut -c1-260 "$OUT/bench.json"
# kernel trace of the same step
bash tools/prof_step.sh "$TAG/step" > /dev/null 2>&1
head -12 "$OUT/step/bench_kernel_stats.txt" | cut -c1-150
# fixed-base MSM: timings (full length + prefixes) and the kernel sequence of one 2^26-term MSM
JOLT_BENCH_PREFIXES="20 21 22 23 24 25" timeout 600 python tools/bench_msm_fixed.py 26 23 26 > "$OUT/msm_fixed.jsonl" 2> "$OUT/msm_fixed.err"
JOLT_MSM_LANES=1 bash tools/prof_msm_fixed.sh 26 23 30 > "$OUT/msm_fixed_kernels.txt" 2>&1
# sparse read-write matrix and Spartan outer sums at trace scale
timeout 300 python tools/bench_rw.py 20 16 > "$OUT/rw_matrix.txt" 2>&1
timeout 300 python tools/bench_rw.py 22 16 >> "$OUT/rw_matrix.txt" 2>&1
[ -f tools/bench_r1cs.py ] && timeout 300 python tools/bench_r1cs.py 22 > "$OUT/r1cs.txt" 2>&1
timeout 300 python tools/bench_read_raf.py 20 > "$OUT/read_raf.txt" 2>&1
timeout 300 python tools/bench_read_raf.py 22 >> "$OUT/read_raf.txt" 2>&1
# sharded commit / open: both ranks on this GPU (a code-path / memory check at real table sizes, not a measurement), then the two-rank
# 2^23-coefficient opening against the beta-known identities
timeout 600 bash tools/bench_two_ranks_one_gpu.sh 20 > "$OUT/two_ranks_one_gpu.txt" 2>&1
timeout 900 python tools/check_subtree_scale.py 18 > "$OUT/subtree_scale.txt" 2>&1
tail -2 "$OUT/subtree_scale.txt"
# bind kernel: HIP-event roofline leg, the same command under rocprofv3, and its HBM traffic from PMC (separate passes)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_bind
timeout 300 rocprofv3 --kernel-trace -d /tmp/p_bind -o b -- python "$ROOT/bench.py" --roofline-only > "$OUT/bind_roofline_bench.json" 2> "$OUT/bind_roofline.err"
f=$(find /tmp/p_bind -name "*.db" | head -1); [ -n "$f" ] && python "$ROOT/profiles/summarize_rocprof.py" "$f" | head -8 > "$OUT/bind_roofline_kernel_stats.txt"
cd "$ROOT" && bash profiles/collect_bind_traffic.sh > /dev/null 2>&1; cp gpurun_out/bind_traffic.json "$OUT/bind_traffic.json" 2>/dev/null
# SQ counters of the round kernels (sumcheck legs only, one stream so that kernels do not overlap)
cd /tmp
rm -rf /tmp/p_pmc
JOLT_SERIAL_STREAMS=1 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/p_pmc -o p -- \
  python "$ROOT/bench.py" --no-msm --no-cpu-baseline --no-split --steps 1 --warmup 1 > "$OUT/pmc_bench.json" 2> "$OUT/pmc.err"
f=$(find /tmp/p_pmc -name "*.db" | head -1); [ -n "$f" ] && python "$ROOT/profiles/pmc_kernel_summary.py" "$f" > "$OUT/pmc_round_kernels.txt"
ls -la "$OUT"
