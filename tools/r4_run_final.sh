#!/bin/bash
# Round-4 final call: the whole GPU suite, smoke() and the driver's command at the last code state
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04final
mkdir -p "$OUT"
cd "$ROOT"
timeout 1800 python -m pytest tests -q -m gpu --durations=12 > "$OUT/pytest_gpu.txt" 2>&1
tail -3 "$OUT/pytest_gpu.txt"
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) | tee "$OUT/smoke.txt"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("bench", d["ms_per_step"], d["value"], d["config"].get("ms_per_step_split"), d["roofline"]["frac"], d["roofline_sumcheck"]["frac"], d["cpu_baseline"]["value"])
PY
timeout 300 python bench.py --stages 2-6b --no-cpu-baseline --steps 10 --warmup 3 > "$OUT/bench_stages_2-6b.json" 2>/dev/null
python -c "import json; d=json.loads(open('$OUT/bench_stages_2-6b.json').read().strip().splitlines()[-1]); print('stages 2-6b', d['ms_per_step'], d['config'].get('ms_per_step_split'))"
