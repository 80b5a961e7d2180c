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
# the run is one warm-up step + one timed step, identical in their kernels: everything but the set-up kernels, halved (a positional "second half" broke when the
# capacity sort added its no-op fallback launches)
acc = collections.OrderedDict()
for r in rows:
    try:
        dur = float(r[-2])
    except ValueError:
        continue
    if r[2] in ("k_fx_to_lform", "k_fx_next_window", "k_srs_powers"):
        continue
    acc[r[2]] = acc.get(r[2], 0.0) + dur / 2
print("# kernel time of ONE bench step, JOLT_MSM_LANES=1 (no overlap between MSMs): ms per kernel name, MSM / PCS kernels only (tools/seq_step.sh; the run's two steps halved)")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{v/1e3:10.3f} ms  {k}")
PY
head -16 "$OUT/seq_kernel_sums.txt"
bash tools/prof_sumcheck.sh 22 "$TAG/sumcheck" > "$OUT/sumcheck.log" 2>&1
tail -5 "$OUT/sumcheck.log"
