#!/bin/bash
# Everything the round's measurements in DESIGN.md section 4 come from, in one GPU-box call:
#   bash tools/collect_round.sh r03 [notests]
# writes gpurun_out/<tag>/...; the files to keep are then copied into profiles/ as <tag>_* (tools/keep_round.sh).
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
# a forced rebuild of both libraries ON THIS BOX: what __graft_entry__.build() does when the objects are stale (the snapshot ships them prebuilt)
( { time python -m jolt_amd.build --force; } > "$OUT/build_force.txt" 2>&1; { time make -C oracle -B; } >> "$OUT/build_force.txt" 2>&1; grep -E "^real|libjolt_hip.so|rror" "$OUT/build_force.txt" | tail -4 )
if [ "${2:-}" != "notests" ]; then
  timeout 1800 python -m pytest tests -q -m gpu --durations=12 > "$OUT/pytest_gpu.txt" 2>&1
  tail -3 "$OUT/pytest_gpu.txt"
fi
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) | tee "$OUT/smoke.txt"
# the driver's command (default flags: all stages), the round-2 step for comparison, and the sumcheck legs alone (BASELINE configs[1] shape)
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
cut -c1-200 "$OUT/bench.json"
timeout 300 python bench.py --stages 2-6b --no-cpu-baseline --no-msm-roofline --no-upload-rate --steps 10 --warmup 3 > "$OUT/bench_stages_2-6b.json" 2>/dev/null
timeout 300 python bench.py --no-msm --stages 2-6b --scale 22 --no-cpu-baseline --steps 10 --warmup 3 > "$OUT/bench_nomsm_22.json" 2>/dev/null
timeout 300 python bench.py --no-msm --stages 2-6b --scale 20 --no-cpu-baseline --steps 10 --warmup 3 > "$OUT/bench_nomsm_20.json" 2>/dev/null
for f in bench bench_stages_2-6b bench_nomsm_22 bench_nomsm_20; do python -c "import json,sys; d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['config'].get('ms_per_step_split'))"; done
# kernel trace of the same step (four MSM lanes: durations overlap) and the kernel sums of one step with ONE lane (true kernel times)
bash tools/prof_step.sh "$TAG/step" > /dev/null 2>&1
head -12 "$OUT/step/bench_kernel_stats.txt" | cut -c1-150
bash tools/seq_step.sh "$TAG/seq" > "$OUT/seq.log" 2>&1
python - "$OUT/seq/seq_all.txt" > "$OUT/seq_kernel_sums.txt" <<'PY'
import sys, collections
rows = [l.split() for l in open(sys.argv[1]) if l.strip()]
# the run is one warm-up step + one timed step, identical in their kernels: everything but the set-up kernels, halved (a positional "second half" broke when the
# capacity sort added its no-op fallback launches)
acc = collections.OrderedDict()
for r in rows:
    try:
        dur = float(r[-2])
    except ValueError:
        continue
    if r[2] in ("k_fx_to_lform", "k_fx_next_window", "k_srs_powers"):
        continue
    acc[r[2]] = acc.get(r[2], 0.0) + dur / 2
print("# kernel time of ONE bench step, JOLT_MSM_LANES=1 (no overlap between MSMs): ms per kernel name, MSM / PCS kernels only (tools/seq_step.sh; the run's two steps halved)")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{v/1e3:10.3f} ms  {k}")
PY
head -16 "$OUT/seq_kernel_sums.txt"
# the witness path: every step from packed rows in page-locked host memory (PCIe-inclusive; never `value`)
timeout 300 python bench.py --witness upload-pinned --no-cpu-baseline --no-msm-roofline --steps 10 --warmup 3 > "$OUT/bench_witness_upload.json" 2>/dev/null
python -c "import json; d=json.loads(open('$OUT/bench_witness_upload.json').read().strip().splitlines()[-1]); print('witness upload', d['ms_per_step'], d['config'].get('witness'))"
# ... and with the next proof's rows copied under the current proof (round 5)
timeout 300 python bench.py --witness upload-overlapped --no-cpu-baseline --no-msm-roofline --steps 10 --warmup 3 > "$OUT/bench_witness_upload_overlapped.json" 2>/dev/null
python -c "import json; d=json.loads(open('$OUT/bench_witness_upload_overlapped.json').read().strip().splitlines()[-1]); print('witness upload overlapped', d['ms_per_step'], d['config'].get('witness'))"
# the step's own opening: what runs while no bucket sum does (profiles/open_exposed.py), min of 3 wall times
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_os && timeout 500 rocprofv3 --kernel-trace -d /tmp/p_os -o o -- python "$ROOT/tools/open_step.py" 22 1 > "$OUT/open_step.txt" 2>&1; \
  f=$(find /tmp/p_os -name "*.db" | head -1); python "$ROOT/profiles/open_exposed.py" "$f" 34 > "$OUT/open_exposed_step.txt" 2>&1 )
echo "open of the step, 3 reps: $(timeout 300 python tools/open_step.py 22 3 2>&1 | grep 'open ms')" | tee -a "$OUT/open_exposed_step.txt"
# the stage operators' HBM traffic per kernel (counter-free trace + FETCH_SIZE / WRITE_SIZE passes)
bash tools/pmc_extended.sh "$OUT/pmc_ext" 22 2 > /dev/null 2>&1
head -14 "$OUT/pmc_ext/stage_operator_traffic.txt" | cut -c1-140
# fixed-base MSM: timings (full length + prefixes) and the kernel table of one 2^26-term MSM
JOLT_BENCH_PREFIXES="20 21 22 23 24 25" timeout 600 python tools/bench_msm_fixed.py 26 23 > "$OUT/msm_fixed.jsonl" 2> "$OUT/msm_fixed.err"
JOLT_MSM_LANES=1 bash tools/prof_msm_fixed.sh 26 23 34 > "$OUT/msm_fixed_kernels.txt" 2>&1
# the extended-stage operators part by part, the sumcheck legs per stage and per kernel
timeout 300 python tools/time_extended.py 22 > "$OUT/extended_parts.txt" 2>&1
bash tools/prof_extended.sh 22 70 > "$OUT/extended_kernel_stats.txt" 2>&1
bash tools/prof_sumcheck.sh 22 "$TAG/sumcheck" > "$OUT/sumcheck.log" 2>&1
# sharded paths with every rank on this GPU (code-path / memory checks, not measurements): bench.py --gpus 2 launching itself, the two-rank
# 2^23-coefficient subtree opening against the oracle
JOLT_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --scale 18 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_gpus2_share_gpu.json" 2> "$OUT/bench_gpus2.err"
grep "^{" "$OUT/bench_gpus2_share_gpu.json" > "$OUT/bench_gpus2.tmp"; mv "$OUT/bench_gpus2.tmp" "$OUT/bench_gpus2_share_gpu.json"  # the launcher's [Gloo] chatter is not part of the line
grep "^{" "$OUT/bench_gpus2_share_gpu.json" | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gpus 2 (one GPU shared):', d['n_gpus'], d['ms_per_step'], d['config'].get('round_exchange_ab'), d['config']['communicator'][:80])"
timeout 900 python tools/check_subtree_scale.py 18 > "$OUT/subtree_scale.txt" 2>&1
tail -1 "$OUT/subtree_scale.txt"
# bind kernel: HIP-event roofline leg, the same command under rocprofv3, and its HBM traffic from PMC (separate passes)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_bind
timeout 300 rocprofv3 --kernel-trace -d /tmp/p_bind -o b -- python "$ROOT/bench.py" --roofline-only > "$OUT/bind_roofline_bench.json" 2> "$OUT/bind_roofline.err"
f=$(find /tmp/p_bind -name "*.db" | head -1); [ -n "$f" ] && python "$ROOT/profiles/summarize_rocprof.py" "$f" | head -8 > "$OUT/bind_roofline_kernel_stats.txt"
cd "$ROOT" && bash profiles/collect_bind_traffic.sh > /dev/null 2>&1; cp gpurun_out/bind_traffic.json "$OUT/bind_traffic.json" 2>/dev/null
cat "$OUT/bind_roofline_bench.json" | cut -c1-300
# SQ counters of the round kernels (sumcheck legs only, one stream so that kernels do not overlap), with the derived VALU-active share
cd /tmp
rm -rf /tmp/p_pmc
JOLT_SERIAL_STREAMS=1 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/p_pmc -o p -- \
  python "$ROOT/bench.py" --no-msm --stages 2-6b --no-cpu-baseline --no-split --steps 1 --warmup 1 > "$OUT/pmc_bench.json" 2> "$OUT/pmc.err"
f=$(find /tmp/p_pmc -name "*.db" | head -1); [ -n "$f" ] && python "$ROOT/profiles/pmc_kernel_summary.py" "$f" > "$OUT/pmc_round_kernels.txt"
grep -A8 "k_round_evals_group<2, 0, true, false>" "$OUT/pmc_round_kernels.txt" | head -9
ls "$OUT"
