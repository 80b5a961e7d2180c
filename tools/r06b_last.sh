#!/bin/bash
# the round's last GPU call: the whole suite, smoke and the driver's command at the final commit
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06b_last
mkdir -p "$OUT"; cd "$ROOT"
timeout 1800 python -m pytest tests -q -m gpu --durations=6 > "$OUT/pytest_gpu.txt" 2>&1; tail -3 "$OUT/pytest_gpu.txt"
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) | tee "$OUT/smoke.txt"
( time timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" ) 2>&1 | grep real
python -c "import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['ms_per_step_split'], d['config']['device_pool_gib'], d['roofline']['frac'], d['roofline_msm']['frac'])"
