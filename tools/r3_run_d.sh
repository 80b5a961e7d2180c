#!/bin/bash
# Round-3 re-entry GPU call: the read-RAF kernel end to end (tests + bench + host-thread A/B).  bash tools/r3_run_d.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03d
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python -m pytest tests/test_gpu_read_raf.py tests/test_gpu_extended.py -q -x --durations=8 > "$OUT/pytest_read_raf_extended.txt" 2>&1
tail -14 "$OUT/pytest_read_raf_extended.txt"
timeout 400 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python -c "import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['value'], d['config'].get('ms_per_step_split'))"
for n in 1 4 8 16; do
  JOLT_HOST_THREADS=$n timeout 200 python tools/time_extended.py 22 2>&1 | head -2 | tail -1 | cut -c1-400 > "$OUT/extended_parts_threads$n.txt"
  echo "threads $n: $(cat $OUT/extended_parts_threads$n.txt)"
done
JOLT_HOST_THREADS=8 timeout 300 python bench.py --no-cpu-baseline --steps 8 --warmup 3 > "$OUT/bench_threads8.json" 2>/dev/null
python -c "import json; d=json.loads(open('$OUT/bench_threads8.json').read().strip().splitlines()[-1]); print('bench threads 8', d['ms_per_step'], d['config'].get('ms_per_step_split'))"
