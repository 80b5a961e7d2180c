#!/bin/bash
# copy the judged summaries of a collect_round.sh run from gpurun_out/<src>/ into profiles/ as <tag>_*:  bash tools/keep_round.sh r02a r02
set -eu
SRC=gpurun_out/${1:?source dir under gpurun_out}; TAG=${2:?tag}
cd "$(dirname "$0")/.."
cp $SRC/bench.json profiles/${TAG}_bench.json
cp $SRC/step/bench_kernel_stats.txt profiles/${TAG}_bench_kernel_stats.txt
cp $SRC/step/bench_under_rocprof.json profiles/${TAG}_bench_under_rocprof.json
cp $SRC/msm_fixed.jsonl profiles/${TAG}_msm_fixed_base.jsonl
cp $SRC/msm_fixed_kernels.txt profiles/${TAG}_msm_fixed_base_kernels.txt
cp $SRC/rw_matrix.txt profiles/${TAG}_rw_matrix.txt
[ -f $SRC/r1cs.txt ] && cp $SRC/r1cs.txt profiles/${TAG}_spartan_outer.txt
[ -f $SRC/read_raf.txt ] && cp $SRC/read_raf.txt profiles/${TAG}_read_raf.txt
cp $SRC/bind_roofline_bench.json profiles/${TAG}_bind_roofline_bench.json
cp $SRC/bind_roofline_kernel_stats.txt profiles/${TAG}_bind_roofline_kernel_stats.txt
cp $SRC/bind_traffic.json profiles/bind_traffic.json
cp $SRC/pmc_round_kernels.txt profiles/${TAG}_pmc_round_kernels.txt
[ -f $SRC/pytest_gpu.txt ] && cp $SRC/pytest_gpu.txt profiles/${TAG}_pytest_gpu.txt
ls profiles | grep "^${TAG}_"
