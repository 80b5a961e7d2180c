#!/bin/bash
# Round-4 call H: one-pass pair quotient on the chunked scans, plus-minus Horner, plain-atomic histogram, first reduction of a pair under its second pass -- tests, A/B (min of 3), timeline, bench
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04h
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_pcs.py -q -m gpu -x --durations=4 > "$OUT/pytest_msm_pcs.txt" 2>&1
tail -8 "$OUT/pytest_msm_pcs.txt"
timeout 900 python -m pytest tests/test_gpu_distributed.py -q -m gpu -x -k "open or pcs or subtree or commit" > "$OUT/pytest_dist.txt" 2>&1
tail -3 "$OUT/pytest_dist.txt"
for cfg in "" "JOLT_MSM_PAIR_OVERLAP=0" "JOLT_KZG_QUOTIENT2=0" "JOLT_MSM_BATCH=0" "JOLT_HORNER_STRIDED=0"; do
  echo "[$cfg] $(env $cfg timeout 300 python tools/open_one.py 26 3 2>&1 | grep 'open ms')"
done | tee "$OUT/open_ab.txt"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_open4
timeout 400 rocprofv3 --kernel-trace -d /tmp/p_open4 -o o -- python "$ROOT/tools/open_one.py" 26 1 > "$OUT/open_one.txt" 2>&1
f=$(find /tmp/p_open4 -name "*.db" | head -1)
python "$ROOT/profiles/open_exposed.py" "$f" 30 > "$OUT/open_exposed.txt" 2>&1
cat "$OUT/open_exposed.txt" | cut -c1-120
python "$ROOT/profiles/kernel_sequence.py" "$f" 360 > "$OUT/open_sequence.txt" 2>&1 || true
cd "$ROOT"
timeout 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("bench", d["ms_per_step"], d["value"], d["config"].get("ms_per_step_split"))
PY
