#!/bin/bash
# Round 6, final collection on one GPU box: the suite, smoke, collect_round.sh's measurements, rocprofv3's own --stats tables of the two roofline kernels, the CPU
# baseline at the GPU line's T, the hot-set bench line.  Everything lands under gpurun_out/r06/; tools/keep_round.sh + the cp lines at the end of this file's
# header comment copy the judged summaries into profiles/r06_*.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
bash tools/collect_round.sh $TAG 2>&1 | tail -60
# rocprofv3's OWN --kernel-trace --stats tables (csv) of the bind roofline command and of one 2^26-term MSM (what roofline_msm runs)
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_st1 /tmp/p_st2
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_st1 -o bind -- python "$ROOT/bench.py" --roofline-only > "$OUT/bind_roofline_bench_stats_run.json" 2> "$OUT/bind_stats.err"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_st2 -o msm -- python "$ROOT/tools/msm_bucket_one.py" 26 3 > "$OUT/msm_bucket_one.json" 2> "$OUT/msm_stats.err"
  for f in $(find /tmp/p_st1 -name "*kernel_stats.csv"); do head -6 "$f" > "$OUT/bind_roofline_rocprofv3_kernel_stats.csv"; done
  for f in $(find /tmp/p_st2 -name "*kernel_stats.csv"); do head -14 "$f" > "$OUT/msm_bucket_rocprofv3_kernel_stats.csv"; done )
cut -c1-200 "$OUT/bind_roofline_rocprofv3_kernel_stats.csv"; cut -c1-200 "$OUT/msm_bucket_rocprofv3_kernel_stats.csv" | head -5
# the hot-set (btreemap-shaped) address stream: one bench line with its split
timeout 600 python bench.py --ram-addresses hotset --no-cpu-baseline --no-upload-rate --no-msm-roofline --steps 10 --warmup 3 > "$OUT/bench_hotset.json" 2>/dev/null
python -c "import json; d=json.loads(open('$OUT/bench_hotset.json').read().strip().splitlines()[-1]); print('hotset', d['ms_per_step'], d['config'].get('ms_per_step_split'))"
# the CPU legs at the GPU line's own T (about 2.5 minutes of CPU work): cpu_baseline.config.at_gpu_T of later default runs points here
timeout 1500 python bench.py --cpu-scale 22 --steps 3 --warmup 1 --no-split --no-upload-rate --no-msm-roofline > "$OUT/bench_cpu_T22.json" 2> "$OUT/bench_cpu_T22.err"
python -c "import json; d=json.loads(open('$OUT/bench_cpu_T22.json').read().strip().splitlines()[-1]); json.dump(d['cpu_baseline'], open('$OUT/cpu_baseline_T22.json','w'), indent=1); print('cpu T22', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['config']['seconds_per_step'])"
ls "$OUT" | head -80
