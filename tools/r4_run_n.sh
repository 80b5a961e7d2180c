#!/bin/bash
# Round-4 call N: side streams above the main stream in priority -- A/B of the whole step (commit leg, sumcheck legs, opening)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04n
mkdir -p "$OUT"
cd "$ROOT"
for cfg in "JOLT_SIDE_PRIORITY=1" "" "JOLT_SIDE_PRIORITY=1" ""; do
  env $cfg timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
  python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
sp=d["config"].get("ms_per_step_split")
print("bench [$cfg]", d["ms_per_step"], {k: sp[k] for k in ("prepare","commit","prove","open")}, round(sum(v for k,v in sp.items() if k not in ("prepare","commit","prove","open")),2))
PY
done | tee "$OUT/side_priority_ab.txt"
