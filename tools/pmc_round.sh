# rocprofv3 --pmc pass (SQ wave / wait / VALU counters, no other trace domains) over one bench pass; per-kernel averages of the
# round kernels into gpurun_out/pmc_round_group.txt (profiles/pmc_kernel_summary.py).
cd /tmp && export TMPDIR=/tmp
JOLT_SERIAL_STREAMS=1 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace -d /tmp/pmc -o p -- python /root/repo/bench.py --no-cpu-baseline --steps 1 --warmup 1 > /tmp/o.txt 2>&1
f=$(find /tmp/pmc -name "*.db" | head -1)
[ -z "$f" ] && { tail -5 /tmp/o.txt; echo nodb; exit 1; }
timeout 100 python /root/repo/profiles/pmc_kernel_summary.py "$f" k_round_evals_group > /root/repo/gpurun_out/pmc_round_group.txt 2>&1
timeout 100 python /root/repo/profiles/pmc_kernel_summary.py "$f" k_bind_low >> /root/repo/gpurun_out/pmc_round_group.txt 2>&1
timeout 100 python /root/repo/profiles/pmc_kernel_summary.py "$f" lazy_lds >> /root/repo/gpurun_out/pmc_round_group.txt 2>&1
cat /root/repo/gpurun_out/pmc_round_group.txt | cut -c1-120
