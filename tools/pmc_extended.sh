#!/bin/bash
# Roofline accounting of the stage operators at T = 2^22: one counter-free kernel trace + FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (MI355X guide, HBM section)
# over tools/run_extended.py, joined by profiles/stage_operator_traffic.py.   bash tools/pmc_extended.sh <out dir> [log_t] [proofs]
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$ROOT/gpurun_out/pmc_ext}
LOGT=${2:-22}
PROOFS=${3:-2}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d "$OUT/trace" -o e -- python "$ROOT/tools/run_extended.py" $LOGT $PROOFS > "$OUT/trace.txt" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $C --kernel-trace -d "$OUT/$C" -o e -- python "$ROOT/tools/run_extended.py" $LOGT $PROOFS > "$OUT/$C.txt" 2>&1
done
python "$ROOT/profiles/stage_operator_traffic.py" "$OUT" $PROOFS > "$OUT/stage_operator_traffic.txt" 2>&1
f=$(find "$OUT/trace" -name "*.db" | head -1); python "$ROOT/profiles/summarize_rocprof.py" "$f" > "$OUT/kernel_stats.txt" 2>&1
find "$OUT" -name "*.db" -size +20M -delete
head -45 "$OUT/stage_operator_traffic.txt" | cut -c1-140
