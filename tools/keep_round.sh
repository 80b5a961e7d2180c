#!/bin/bash
# copy the judged summaries of a collect_round.sh run from gpurun_out/<src>/ into profiles/ as <tag>_*:  bash tools/keep_round.sh r03 r03
set -eu
SRC=gpurun_out/${1:?source dir under gpurun_out}; TAG=${2:?tag}
cd "$(dirname "$0")/.."
keep() { [ -f "$SRC/$1" ] && cp "$SRC/$1" "profiles/${TAG}_$2" || echo "missing $1"; }
keep bench.json bench.json
keep bench_stages_2-6b.json bench_stages_2-6b.json
keep bench_nomsm_22.json bench_nomsm_T22.json
keep bench_nomsm_20.json bench_nomsm_T20.json
keep step/bench_kernel_stats.txt bench_kernel_stats.txt
keep step/bench_under_rocprof.json bench_under_rocprof.json
keep seq_kernel_sums.txt step_kernel_sums_one_lane.txt
keep msm_fixed.jsonl msm_fixed_base.jsonl
keep msm_fixed_kernels.txt msm_fixed_base_kernels.txt
keep extended_parts.txt extended_stage_parts.txt
keep extended_kernel_stats.txt extended_kernel_stats.txt
keep sumcheck/stage_times_22.txt sumcheck_stage_times_T22.txt
keep sumcheck/sumcheck_kernel_stats_22.txt sumcheck_kernel_stats_T22.txt
keep bench_gpus2_share_gpu.json bench_gpus2_share_gpu.json
keep subtree_scale.txt subtree_scale_check.txt
keep bind_roofline_bench.json bind_roofline_bench.json
keep bind_roofline_kernel_stats.txt bind_roofline_kernel_stats.txt
keep pmc_round_kernels.txt pmc_round_kernels.txt
keep pytest_gpu.txt pytest_gpu.txt
keep bench_witness_upload.json bench_witness_upload.json
keep bench_witness_upload_overlapped.json bench_witness_upload_overlapped.json
keep open_exposed_step.txt open_exposed_step.txt
keep pmc_ext/stage_operator_traffic.txt stage_operator_traffic.txt
keep build_force.txt build_force.txt
[ -f "$SRC/bind_traffic.json" ] && cp "$SRC/bind_traffic.json" profiles/bind_traffic.json
ls profiles | grep "^${TAG}_"
