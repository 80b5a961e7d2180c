# SQ counters of the fixed-base MSM kernels at 2^26 terms (one --pmc pass with --kernel-trace only) -> gpurun_out/<tag>/pmc_msm.txt
set -u
TAG=${1:-r02}
OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_pm
JOLT_MSM_LANES=1 timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/p_pm -o p -- python /root/repo/tools/bench_msm_fixed.py 26 23 > /dev/null 2> $OUT/pmc_msm.err
f=$(find /tmp/p_pm -name "*.db" | head -1)
python /root/repo/profiles/pmc_kernel_summary.py "$f" k_fx_ > $OUT/pmc_msm.txt
python /root/repo/profiles/pmc_kernel_summary.py "$f" k_msm_window_reduce >> $OUT/pmc_msm.txt
grep -A8 "k_fx_buckets_ordered" $OUT/pmc_msm.txt | head -10
