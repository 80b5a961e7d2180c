#!/bin/bash
# GPU-box call: extended-stage parity, MSM tests with the new reduction, bench with / without the extended stages, one-lane kernel sums
set -u
TAG=${1:-r3c}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_extended.py tests/test_gpu_msm.py tests/test_gpu_pcs.py tests/test_gpu_rw.py -q -x -m gpu --durations=8 > "$OUT/pytest.txt" 2>&1; tail -12 "$OUT/pytest.txt"
for st in all 2-6b; do
  timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 --stages $st 2>"$OUT/bench_$st.err" | tee "$OUT/bench_$st.json" | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$st', d['ms_per_step'], d['config']['ms_per_step_split'])"
done
JOLT_FX_REDUCE=0 timeout 300 python bench.py --no-cpu-baseline --no-split --steps 6 --warmup 2 --stages 2-6b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2-6b, running-sum reduction', d['ms_per_step'])"
bash tools/seq_step.sh "$TAG/seq" > "$OUT/seq.log" 2>&1; tail -2 "$OUT/seq.log"
python - "$OUT/seq/seq_all.txt" > "$OUT/seq_summary.txt" <<'PY'
import sys, collections
rows = [l.split() for l in open(sys.argv[1]) if l.strip()]
half = rows[len(rows) // 2:]
acc = collections.OrderedDict()
for r in half:
    try:
        dur = float(r[-2])
    except ValueError:
        continue
    acc[r[2]] = acc.get(r[2], 0.0) + dur
for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:24]:
    print(f"{v/1e3:10.3f} ms  {k}")
PY
head -24 "$OUT/seq_summary.txt"
