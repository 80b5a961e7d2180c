#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05m
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python bench.py --no-cpu-baseline --no-msm-roofline --no-upload-rate --witness upload-overlapped --steps 5 --warmup 2 > "$OUT/a.json" 2> "$OUT/a.err"
timeout 300 python bench.py --no-cpu-baseline --no-msm-roofline --steps 10 --warmup 3 > "$OUT/b.json" 2> "$OUT/b.err"
timeout 300 python bench.py --no-cpu-baseline --no-msm-roofline --no-split --steps 5 --warmup 2 > "$OUT/c.json" 2> "$OUT/c.err"
python - <<PY
import json
for f in "abc":
    d=json.loads([l for l in open("$OUT/%s.json" % f) if l.startswith("{")][-1])
    print(f, d["ms_per_step"], d.get("value_with_upload"), (d["config"].get("witness_upload") or {}).get("ms_per_step"))
PY
