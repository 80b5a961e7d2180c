#!/bin/bash
# kernel sequence of ONE fixed-base prefix MSM of 2^$1 terms over the 2^26-point window tables (one lane): where a short MSM's time goes
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p_pre
cat > /tmp/pre_one.py <<PY
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
from jolt_amd import ffi
from jolt_amd.workload import G1_GENERATOR, rand_fr
lg = int(sys.argv[1])
rng = np.random.default_rng(3)
ctx = ffi.Context(0)
srs = ctx.srs_setup_from_secret(rand_fr(1, rng)[0], 1 << 26, G1_GENERATOR)
ctx.srs_precompute_windows(srs, 23, 1)
tab = ctx.eq_evals(rand_fr(lg, rng))
ctx.msm(srs, tab); ctx.synchronize()
ctx.msm(srs, tab); ctx.synchronize()
PY
JOLT_MSM_LANES=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/p_pre -o s -- python /tmp/pre_one.py ${1:-21} > /tmp/pre.txt 2>&1
f=$(find /tmp/p_pre -name "*.db" | head -1); python /root/repo/profiles/kernel_sequence.py "$f" 2>/dev/null | tail -${2:-32}
