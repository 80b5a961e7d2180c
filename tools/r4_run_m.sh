#!/bin/bash
# Round-4 call M: dense commitments in flight under the one-hot sums -- parity (commit against the oracle at small sizes and at the benchmarked one), A/B of the commit leg, bench
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04m
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_pcs.py tests/test_gpu_workload.py -q -m gpu -x --durations=4 > "$OUT/pytest.txt" 2>&1
tail -4 "$OUT/pytest.txt"
for cfg in "" "JOLT_COMMIT_OVERLAP=0"; do
  env $cfg timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_${cfg:-default}.json" 2> "$OUT/bench.err"
  python - <<PY
import json
d=json.loads(open("$OUT/bench_${cfg:-default}.json").read().strip().splitlines()[-1])
print("bench [$cfg]", d["ms_per_step"], d["config"].get("ms_per_step_split"))
PY
done | tee "$OUT/commit_overlap_ab.txt"
