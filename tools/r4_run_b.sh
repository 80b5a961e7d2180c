#!/bin/bash
# Round-4 call B: compact-scalar members (small-scalar round 0 + bind_to_field): parity suites, then the sumcheck legs with / without them, kernel table.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04b
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_small_round.py tests/test_gpu_sumcheck.py tests/test_gpu_workload.py tests/test_gpu_poly.py -q -m gpu -x --durations=8 > "$OUT/pytest_small.txt" 2>&1
tail -15 "$OUT/pytest_small.txt"
for sc in 22 20; do
  for sm in 1 0; do
    JOLT_SMALL_ROUND0=$sm timeout 300 python bench.py --scale $sc --no-msm --stages 2-6b --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/nomsm_T${sc}_small${sm}.json" 2> "$OUT/nomsm_T${sc}_small${sm}.err"
    python - <<PY
import json
d=json.loads(open("$OUT/nomsm_T${sc}_small${sm}.json").read().strip().splitlines()[-1])
print("T=2^$sc small=$sm", d["ms_per_step"], d["config"].get("ms_per_step_split"), d.get("roofline_sumcheck",{}).get("frac"))
PY
  done
done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p_sc
JOLT_SERIAL_STREAMS=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/p_sc -o s -- python "$ROOT/bench.py" --no-msm --stages 2-6b --scale 22 --steps 3 --warmup 1 --no-cpu-baseline --no-split > /dev/null 2>&1
f=$(find /tmp/p_sc -name "*.db" | head -1); [ -n "$f" ] && python "$ROOT/profiles/summarize_rocprof.py" "$f" > "$OUT/sumcheck_kernel_stats_T22.txt"
cd "$ROOT"
head -40 "$OUT/sumcheck_kernel_stats_T22.txt" | cut -c1-170
