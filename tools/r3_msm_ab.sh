#!/bin/bash
# Round-3 MSM measurements in one GPU-box call: bash tools/r3_msm_ab.sh <tag>
#   1. the MSM / PCS tests (heavy buckets by lane-per-segment)   2. kernel sequence of one step with ONE MSM lane (no overlap: true kernel times)
#   3. the default bench step   4. A/B of the bucket kernels at 2 waves per SIMD, with and without the sort token (JOLT_MSM_STAGGER)
set -u
TAG=${1:-r3b}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_pcs.py tests/test_gpu_dory.py -q -x -m gpu > "$OUT/pytest_msm.txt" 2>&1; tail -3 "$OUT/pytest_msm.txt"
bash tools/seq_step.sh "$TAG/seq" > "$OUT/seq.log" 2>&1; tail -2 "$OUT/seq.log"
python - "$OUT/seq/seq_all.txt" > "$OUT/seq_summary.txt" <<'PY'
import sys, collections
rows = [l.split() for l in open(sys.argv[1]) if l.strip()]
# kernel_sequence.py prints: start_ms dur_us name ... ; keep the last step: everything after the largest gap is too fragile -> take the second half
half = rows[len(rows) // 2:]
acc = collections.OrderedDict()
for r in half:
    try:
        dur = float(r[-2])
    except ValueError:
        continue
    name = r[2]
    acc[name] = acc.get(name, 0.0) + dur
for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:30]:
    print(f"{v/1e3:10.3f} ms  {k}")
PY
head -20 "$OUT/seq_summary.txt"
for rep in 1 2; do timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['ms_per_step'], d['config']['ms_per_step_split'])"; done | tee "$OUT/bench_default.txt"
JOLT_MSM_STAGGER=1 timeout 300 python bench.py --no-cpu-baseline --no-split --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves3 stagger', d['ms_per_step'])" | tee -a "$OUT/bench_default.txt"
# A/B: bucket kernels at 2 waves per SIMD (rebuilt here; the default build is restored afterwards)
touch jolt_amd/csrc/msm_kernels.hip.h
if JOLT_EXTRA_HIPCC_FLAGS="-DJOLT_BUCKET_WAVES=2" python -m jolt_amd.build > "$OUT/build_w2.log" 2>&1; then
  timeout 300 python bench.py --no-cpu-baseline --no-split --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves2', d['ms_per_step'])" | tee "$OUT/bench_w2.txt"
  JOLT_MSM_STAGGER=1 timeout 300 python bench.py --no-cpu-baseline --no-split --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves2 stagger', d['ms_per_step'])" | tee -a "$OUT/bench_w2.txt"
  JOLT_MSM_LANES=1 timeout 300 python tools/bench_msm_fixed.py 26 23 2>/dev/null | tail -1 | tee -a "$OUT/bench_w2.txt"
else
  echo "waves2 build failed"; tail -5 "$OUT/build_w2.log"
fi
touch jolt_amd/csrc/msm_kernels.hip.h
python -m jolt_amd.build > /dev/null 2>&1
