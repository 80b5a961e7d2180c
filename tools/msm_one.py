"""One G1 MSM at 2^LOG terms (uniform 254-bit scalars) twice -- the rocprofv3 target for profiles/msm_breakdown.py."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from jolt_amd import ffi  # noqa: E402
from tools.bench_msm import rand_fr  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ctx = ffi.Context(0)
g = np.zeros(12, dtype=np.uint64)
g[0:4] = [0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f]
g[4:8] = [0xa6ba871b8b1e1b3a, 0x14f1d651eb8e167b, 0xccdd46def0f28c58, 0x1c14ef83340fbe5e]
g[8:12] = g[0:4]
srs = ctx.srs_setup_from_secret(rand_fr(1, 1)[0], 1 << log_n, g)
tab = ctx.upload(rand_fr(1 << log_n, 5))
ctx.msm(srs, tab)
ctx.msm(srs, tab)
ctx.close()
