#!/bin/bash
# Round-5 call D: the new tests (read-RAF all 128 rounds at 2^16, streamed commitment windows, bench --gpus 8 on a shared GPU) and the driver's default bench command with
# roofline_msm and the reworked cpu_baseline
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05d
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_extended.py tests/test_gpu_pcs.py tests/test_gpu_distributed.py -q -m gpu -k "every_address_round or streamed_commitment or (bench_multi_rank and 8-1)" --durations=5 ) > "$OUT/pytest.txt" 2>&1
echo "rc $?" >> "$OUT/pytest.txt"
tail -12 "$OUT/pytest.txt"
( time timeout 900 python bench.py ) > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "rc $?"
tail -3 "$OUT/bench.err"
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][-1])
print("bench", d["ms_per_step"], d["value"], d["config"].get("ms_per_step_split"))
print("roofline", d["roofline"]["frac"], "msm", d.get("roofline_msm"))
cb=d.get("cpu_baseline",{})
print("cpu", cb.get("value"), cb.get("cores"), cb.get("config"), cb.get("gpu_same_legs"), cb.get("sample","")[:300])
PY
