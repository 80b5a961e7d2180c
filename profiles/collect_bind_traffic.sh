#!/bin/bash
# HBM traffic of the bind kernel from the L2 memory-side counters, collected as the MI355X guide prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (they do not fit the TCC slots together), on the same
# command bench.py's roofline leg runs.  Run on the GPU box from the repo root:   bash profiles/collect_bind_traffic.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_bind
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d "$OUT/$C" -- python "$ROOT/bench.py" --roofline-only > "$OUT/$C.json" 2> "$OUT/$C.err"
done
python "$ROOT/profiles/parse_pmc.py" "$OUT" > "$ROOT/gpurun_out/bind_traffic.json"
cat "$ROOT/gpurun_out/bind_traffic.json"
