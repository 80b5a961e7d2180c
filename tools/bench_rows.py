"""On-device expansion of packed witness rows at T = 2^LOG: an 80-byte record per cycle -> two signed
integer columns as Fr tables and 32 + 4 hot-index columns (RaChunkSelector semantics); time per call, inputs resident."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from jolt_amd import ffi  # noqa: E402

log_t = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T = 1 << log_t
ctx = ffi.Context(0)
rng = np.random.default_rng(4)
rows = rng.integers(0, 256, size=(T, 80), dtype=np.uint8)
r = ffi.Rows(ctx, rows)


def timed(name, fn, reps=5):
    out = fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        o = fn()
        if hasattr(o, "free"):
            o.free()
    ctx.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms")
    return out


timed("table_from_rows (i64 field)", lambda: r.table(0, 8, signed=True))
timed("onehot_from_rows (32 chunks of a 16-byte field, log_k = 4)", lambda: r.onehot(16, 16, [4 * i for i in range(32)], 4))
timed("onehot_from_rows (4 chunks of an 8-byte field, valid byte)", lambda: r.onehot(40, 8, [0, 4, 8, 12], 4, valid_offset=79))
ctx.close()
