#!/bin/bash
# Round-5 call K: the two collection pieces whose bench invocation had picked up the new default legs (value_with_upload / roofline_msm steps in the trace)
set -u
TAG=r05
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
bash tools/seq_step.sh "$TAG/seq" > "$OUT/seq.log" 2>&1
python - "$OUT/seq/seq_all.txt" > "$OUT/seq_kernel_sums.txt" <<'PY'
import sys, collections
rows = [l.split() for l in open(sys.argv[1]) if l.strip()]
half = rows[len(rows) // 2:]  # the timed step (the warm-up step is the first half)
acc = collections.OrderedDict()
for r in half:
    try:
        dur = float(r[-2])
    except ValueError:
        continue
    acc[r[2]] = acc.get(r[2], 0.0) + dur
print("# kernel time of ONE bench step, JOLT_MSM_LANES=1 (no overlap between MSMs): ms per kernel name, MSM / PCS kernels only (tools/seq_step.sh)")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{v/1e3:10.3f} ms  {k}")
PY
head -16 "$OUT/seq_kernel_sums.txt"
bash tools/prof_sumcheck.sh 22 "$TAG/sumcheck" > "$OUT/sumcheck.log" 2>&1
tail -5 "$OUT/sumcheck.log"
