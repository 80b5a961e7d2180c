#!/bin/bash
# round 6, second session: the GPU suite and one bench line after the transcript change
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06b
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > "$OUT/pytest_gpu.txt" 2>&1
tail -12 "$OUT/pytest_gpu.txt"
timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-upload-rate > "$OUT/bench.json" 2> "$OUT/bench.err"
python -c "import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['ms_per_step_split'], d['config']['transcript'][:60])"
