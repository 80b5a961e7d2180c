#!/bin/bash
# HBM traffic per kernel of a 2^26-coefficient opening: one counter-free kernel trace + FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes over tools/open_one.py
# (two openings per run: the joiner divides by 2), joined by profiles/stage_operator_traffic.py.   bash tools/pmc_open.sh <out dir>
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$ROOT/gpurun_out/pmc_open}
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT" ;; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
JOLT_MSM_LANES=1 timeout 300 rocprofv3 --kernel-trace -d "$OUT/trace" -o e -- python "$ROOT/tools/open_one.py" 26 1 > "$OUT/trace.txt" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  JOLT_MSM_LANES=1 timeout 400 rocprofv3 --pmc $C --kernel-trace -d "$OUT/$C" -o e -- python "$ROOT/tools/open_one.py" 26 1 > "$OUT/$C.txt" 2>&1
done
python "$ROOT/profiles/stage_operator_traffic.py" "$OUT" 1 > "$OUT/open_traffic.txt" 2>&1
sed -i 's/^# per proof of the stage operators.*$/# per opening of 2^26 uniform coefficients, ONE MSM lane (kernels do not overlap); two openings in the run, the SRS and window-table builds of the set-up included (k_srs_powers, k_fx_next_window, k_fx_to_lform: divide by nothing, ignore)/' "$OUT/open_traffic.txt"
find "$OUT" -name "*.db" -size +20M -delete
head -40 "$OUT/open_traffic.txt" | cut -c1-140
