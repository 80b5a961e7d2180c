#!/bin/bash
# Round-4 call E: the opening's timeline (what runs while no bucket sum does), the pinned witness upload
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04e
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python -m pytest tests/test_gpu_workload.py -q -m gpu -x -k "witness_upload" > "$OUT/pytest.txt" 2>&1
tail -3 "$OUT/pytest.txt"
timeout 300 python bench.py --steps 10 --warmup 3 --witness upload-pinned --no-cpu-baseline > "$OUT/bench_witness_upload_pinned.json" 2> "$OUT/bench_witness_upload_pinned.err"
python - <<PY
import json
d=json.loads(open("$OUT/bench_witness_upload_pinned.json").read().strip().splitlines()[-1])
print("upload-pinned", d["ms_per_step"], d["config"].get("ms_per_step_split"), d["config"].get("witness"))
PY
cd /tmp && export TMPDIR=/tmp
for lanes in 4 1; do
  rm -rf /tmp/p_open$lanes
  JOLT_MSM_LANES=$lanes timeout 400 rocprofv3 --kernel-trace -d /tmp/p_open$lanes -o o -- python "$ROOT/tools/open_one.py" 26 > "$OUT/open_one_lanes$lanes.txt" 2>&1
  grep "open ms" "$OUT/open_one_lanes$lanes.txt"
  f=$(find /tmp/p_open$lanes -name "*.db" | head -1)
  python "$ROOT/profiles/open_exposed.py" "$f" 30 > "$OUT/open_exposed_lanes$lanes.txt" 2>&1
  cat "$OUT/open_exposed_lanes$lanes.txt" | cut -c1-120
done
