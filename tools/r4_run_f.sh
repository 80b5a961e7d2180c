#!/bin/bash
# Round-4 call F: read-RAF with the next phase's order built under the host rounds (tests, A/B, host threads), the opening's timeline with the window tables
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04f
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_read_raf.py "tests/test_gpu_extended.py::test_extended_stages_match_oracle" -q -m gpu -x --durations=5 > "$OUT/pytest.txt" 2>&1
tail -10 "$OUT/pytest.txt"
for cfg in "1 8" "0 8" "1 2" "1 4"; do
  set -- $cfg
  JOLT_RR_AHEAD=$1 JOLT_HOST_THREADS=$2 timeout 200 python tools/time_extended.py 22 > "$OUT/time_extended_ahead$1_threads$2.txt" 2>&1
  echo "ahead=$1 threads=$2: $(sed -n 2p "$OUT/time_extended_ahead$1_threads$2.txt")"
done
cd /tmp && export TMPDIR=/tmp
for lanes in 4 1; do
  rm -rf /tmp/p_open$lanes
  JOLT_MSM_LANES=$lanes timeout 400 rocprofv3 --kernel-trace -d /tmp/p_open$lanes -o o -- python "$ROOT/tools/open_one.py" 26 > "$OUT/open_one_lanes$lanes.txt" 2>&1
  grep "open ms" "$OUT/open_one_lanes$lanes.txt"
  f=$(find /tmp/p_open$lanes -name "*.db" | head -1)
  python "$ROOT/profiles/open_exposed.py" "$f" 30 > "$OUT/open_exposed_lanes$lanes.txt" 2>&1
  cat "$OUT/open_exposed_lanes$lanes.txt" | cut -c1-120
  python "$ROOT/profiles/kernel_sequence.py" "$f" 520 > "$OUT/open_sequence_lanes$lanes.txt" 2>&1 || true
done
