#!/bin/bash
# Round-4 call P: timing of the mid-size extended tests with the oracle on 128 threads; the N = 2 step at 2 x 2^20 cycles with both ranks on this GPU (memory / code-path check)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04p
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest "tests/test_gpu_extended.py::test_extended_stages_match_oracle" -q -m gpu --durations=3 > "$OUT/pytest.txt" 2>&1
tail -6 "$OUT/pytest.txt"
JOLT_BENCH_SHARE_GPU=1 timeout 1200 python bench.py --gpus 2 --scale 20 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_gpus2_scale20.json" 2> "$OUT/bench_gpus2.err"
grep "^{" "$OUT/bench_gpus2_scale20.json" > "$OUT/tmp.json"; mv "$OUT/tmp.json" "$OUT/bench_gpus2_scale20.json"
python - <<PY
import json
d=json.loads(open("$OUT/bench_gpus2_scale20.json").read().strip().splitlines()[-1])
print("gpus 2 at 2 x 2^20 on one GPU:", d["n_gpus"], d["ms_per_step"], d["config"].get("ms_per_step_split"), d["config"]["baseline_config"][:90])
PY
tail -3 "$OUT/bench_gpus2.err" | cut -c1-300
