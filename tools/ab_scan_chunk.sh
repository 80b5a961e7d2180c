# total time of the suffix-scan and Horner kernels of the bench step for several JOLT_SCAN_CHUNK values (rocprofv3 kernel trace, 3 steps each)
set -u
cd /tmp && export TMPDIR=/tmp
for C in 8 16 32 64; do
  rm -rf /tmp/p_sc
  JOLT_SCAN_CHUNK=$C timeout 300 rocprofv3 --kernel-trace -d /tmp/p_sc -o b -- python /root/repo/bench.py --no-cpu-baseline --no-split --steps 2 --warmup 1 > /tmp/sc.json 2>/dev/null
  f=$(find /tmp/p_sc -name "*.db" | head -1)
  echo "chunk $C: $(python /root/repo/profiles/summarize_rocprof.py "$f" | grep -E "k_suffix|k_horner3" | awk '{s+=$(NF-4)} END {printf "%.2f ms over 3 steps", s}') step $(python -c "import json;print(json.load(open('/tmp/sc.json'))['ms_per_step'])")"
done
