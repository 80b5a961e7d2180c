"""Pushforward G tables (booleanity address phase, optimized/booleanity.rs:24-31) for 36 one-hot columns at T = 2^LOG: time per call."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from jolt_amd import ffi  # noqa: E402
from tools.bench_msm import rand_fr  # noqa: E402

log_t = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = ffi.Context(0)
T, K, N = 1 << log_t, 16, 36
rng = np.random.default_rng(3)
idx = rng.integers(0, K, size=(N, T)).astype(np.uint8)
idx[rng.random((N, T)) < 0.3] = 0xFF
oh = ctx.onehot(idx, K)
pt = rand_fr(log_t, 9)
w = ctx.eq_evals(pt)
g = oh.pushforward(w)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    g = oh.pushforward(w)
ctx.synchronize()
print("pushforward ms", round((time.perf_counter() - t0) / 5 * 1e3, 3), "for", N, "columns x 2^%d cycles" % log_t)
ctx.close()
