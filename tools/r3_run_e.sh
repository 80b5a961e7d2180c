#!/bin/bash
# Round-3 final collection after the read-RAF kernel went end to end: changed tests, the driver's command, the round-2 step on the same box, kernel tables.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03e
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python -m pytest tests/test_gpu_extended.py tests/test_gpu_read_raf.py -q -x -k "6-kw0 or 3-kw3 or 10-kw4 or cycle_columns or 3-67890" > "$OUT/pytest_read_raf_claims.txt" 2>&1
tail -3 "$OUT/pytest_read_raf_claims.txt"
timeout 400 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python -c "import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['value'], d['config'].get('ms_per_step_split'), d['roofline'], d['cpu_baseline']['value'])"
timeout 200 python bench.py --stages 2-6b --no-cpu-baseline --steps 10 --warmup 3 > "$OUT/bench_stages_2-6b.json" 2>/dev/null
python -c "import json; d=json.loads(open('$OUT/bench_stages_2-6b.json').read().strip().splitlines()[-1]); print('stages 2-6b', d['ms_per_step'], d['config'].get('ms_per_step_split'))"
timeout 200 python tools/time_extended.py 22 > "$OUT/extended_parts.txt" 2>&1
head -3 "$OUT/extended_parts.txt" | cut -c1-420
bash tools/prof_extended.sh 22 60 > "$OUT/extended_kernel_stats.txt" 2>&1
head -24 "$OUT/extended_kernel_stats.txt" | cut -c1-150
bash tools/prof_step.sh r03e/step > /dev/null 2>&1
head -14 "$OUT/step/bench_kernel_stats.txt" | cut -c1-150
