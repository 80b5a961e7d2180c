#!/bin/bash
# Round-5 call F: timing A/B of the staged base indices (bucket kernel alone and the whole step)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05f
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
for V in 0 1 0 1; do
  JOLT_FX_STAGE_IDX=$V timeout 300 python tools/msm_bucket_one.py 26 3 >> "$OUT/ab.jsonl" 2>> "$OUT/ab.err"
done
cat "$OUT/ab.jsonl" | cut -c1-330
for V in 0 1; do
  JOLT_FX_STAGE_IDX=$V timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-msm-roofline > "$OUT/bench_stage$V.json" 2> "$OUT/bench_stage$V.err"
  python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench_stage$V.json") if l.startswith("{")][-1])
print("stage_idx=$V", d["ms_per_step"], d["value"], d["config"]["ms_per_step_split"]["open"], d["config"]["ms_per_step_split"]["commit"])
PY
done
