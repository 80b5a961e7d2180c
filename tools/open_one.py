"""One HyperKZG commit + open at 2^LOG (the rocprofv3 target for the opening's GPU timeline); prints the wall time of the second open."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from jolt_amd import ffi  # noqa: E402
from tools.bench_msm import rand_fr  # noqa: E402

ell = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ctx = ffi.Context(0)
g = np.zeros(12, dtype=np.uint64)
g[0:4] = [0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f]
g[4:8] = [0xa6ba871b8b1e1b3a, 0x14f1d651eb8e167b, 0xccdd46def0f28c58, 0x1c14ef83340fbe5e]
g[8:12] = g[0:4]
srs = ctx.srs_setup_from_secret(rand_fr(1, 1)[0], (1 << ell) + 1, g)
ctx.srs_precompute_windows(srs)  # the window tables of the fixed-base MSM, as the benchmarked step has them (setup time)
evals = ctx.upload(rand_fr(1 << ell, 99))
point = rand_fr(ell, 98)
point[:, 0] = 0
point[:, 1] = 0
point[:, 3] &= np.uint64((1 << 61) - 1)
ctx.hyperkzg_open(srs, evals, point, label=7)
ctx.synchronize()
times = []
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 1):
    time.sleep(0.05)  # an idle gap in the kernel trace: profiles/open_exposed.py takes the last burst
    t0 = time.perf_counter()
    ctx.hyperkzg_open(srs, evals, point, label=7)
    times.append(round((time.perf_counter() - t0) * 1e3, 2))
print("open ms", min(times), times)
ctx.close()
