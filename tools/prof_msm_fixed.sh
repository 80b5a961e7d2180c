# kernel breakdown of one fixed-base MSM configuration: prof_msm_fixed.sh <log_n> <window_bits>
set -u
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm; timeout 300 rocprofv3 --kernel-trace -d /tmp/pm -o m -- python /root/repo/tools/bench_msm_fixed.py $1 $2 > /dev/null 2>&1
f=$(find /tmp/pm -name "*.db" | head -1); python /root/repo/profiles/summarize_rocprof.py "$f" | head -${3:-22} | cut -c1-170
