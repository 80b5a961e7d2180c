# time-ordered kernel list of the last step of the bench (MSM lanes serialised): seq_step.sh <tag>
set -u
TAG=${1:-seq}
OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_seq
JOLT_MSM_LANES=1 timeout 600 rocprofv3 --kernel-trace -d /tmp/p_seq -o b -- python /root/repo/bench.py --no-cpu-baseline --no-split --no-upload-rate --no-msm-roofline --steps 1 --warmup 1 > $OUT/bench.json 2> $OUT/err.txt
f=$(find /tmp/p_seq -name "*.db" | head -1)
python /root/repo/profiles/kernel_sequence.py "$f" | grep -E "k_msm|k_fx|k_grid|k_horner|k_suffix|k_rlc" | grep -v "k_fx_next_window\|k_srs" > $OUT/seq_all.txt
# only kernels >= 300 us, last step = second half
awk '{ if ($(NF-1)+0 >= 300) print }' $OUT/seq_all.txt | tail -260 > $OUT/seq_big.txt
wc -l $OUT/seq_all.txt $OUT/seq_big.txt; cut -c1-200 $OUT/bench.json
