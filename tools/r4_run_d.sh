#!/bin/bash
# Round-4 call D: the open-under-the-caller's-transcript and witness-upload tests, the workload tests again (shared cold mask), the default bench (+ witness upload mode),
# then the stage operators' roofline accounting (kernel trace + two PMC passes).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04d
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_pcs.py tests/test_gpu_workload.py -q -m gpu -x --durations=8 > "$OUT/pytest.txt" 2>&1
tail -15 "$OUT/pytest.txt"
timeout 500 python bench.py --steps 10 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 300 python bench.py --steps 10 --warmup 3 --witness upload --no-cpu-baseline > "$OUT/bench_witness_upload.json" 2> "$OUT/bench_witness_upload.err"
python - <<PY
import json
for f in ("bench", "bench_witness_upload"):
    try:
        d=json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], d["config"].get("ms_per_step_split"), d["roofline"]["frac"], d.get("roofline_sumcheck",{}).get("frac"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -3 "$OUT/bench.err" "$OUT/bench_witness_upload.err"
bash tools/pmc_extended.sh "$OUT/pmc_ext" 22 2
