#!/bin/bash
# Round-5 call L: the opening's exposed-time table again (profiles/open_exposed.py now knows the staged bucket kernels by name) and the default bench line at the last code state
set -u
TAG=r05
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_os && timeout 500 rocprofv3 --kernel-trace -d /tmp/p_os -o o -- python "$ROOT/tools/open_step.py" 22 1 > "$OUT/open_step.txt" 2>&1; \
  f=$(find /tmp/p_os -name "*.db" | head -1); python "$ROOT/profiles/open_exposed.py" "$f" 34 > "$OUT/open_exposed_step.txt" 2>&1 )
echo "open of the step, 3 reps: $(timeout 300 python tools/open_step.py 22 3 2>&1 | grep 'open ms')" | tee -a "$OUT/open_exposed_step.txt"
head -16 "$OUT/open_exposed_step.txt" | cut -c1-120
( time timeout 900 python bench.py ) > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -4 "$OUT/bench.err"
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][-1])
print("bench", d["ms_per_step"], d["value"], d.get("value_with_upload"), d["config"]["witness_upload"]["ms_per_step"], d["cpu_baseline"]["value"], d["cpu_baseline"]["config"])
PY
