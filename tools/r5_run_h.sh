#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05h
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python tools/upload_overlap_probe.py > "$OUT/probe.json" 2> "$OUT/probe.err"; cat "$OUT/probe.json"; tail -3 "$OUT/probe.err"
HSA_ENABLE_SDMA=0 timeout 300 python tools/upload_overlap_probe.py > "$OUT/probe_nosdma.json" 2>> "$OUT/probe.err"; cat "$OUT/probe_nosdma.json"
timeout 300 python -m pytest tests/test_gpu_pcs.py -q -m gpu -k "supplied_level" 2>&1 | tail -3
