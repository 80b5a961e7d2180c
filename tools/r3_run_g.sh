#!/bin/bash
# Output-claim flags of read-RAF through the per-lane pushforward (K = 16 columns): changed tests + the default step.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03h
mkdir -p "$OUT"
cd "$ROOT"
timeout 150 python -m pytest tests/test_gpu_extended.py -q -x -k "6-kw0 or 3-kw3 or 10-kw4 or 9-kw1" > "$OUT/pytest_claims.txt" 2>&1
tail -3 "$OUT/pytest_claims.txt"
timeout 150 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
python -c "import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['value'], d['config'].get('ms_per_step_split'))"
