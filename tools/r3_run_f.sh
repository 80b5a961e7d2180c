#!/bin/bash
# Round-3 last validation at the final code state: the whole GPU suite, smoke(), the driver's command.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03f
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests -q -m gpu --durations=12 > "$OUT/pytest_gpu.txt" 2>&1
tail -16 "$OUT/pytest_gpu.txt"
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.txt" 2>&1
tail -2 "$OUT/smoke.txt"
timeout 300 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
python -c "import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['value'], d['config'].get('ms_per_step_split'), d['roofline']['frac'], d['cpu_baseline']['value'])"
