"""The opening of the BENCHMARKED step alone (the joint polynomial of the configs[2] workload: 36 one-hot columns + 2 dense ones on the 2^26 grid, whose first folds hold a
few thousand distinct values many times each -- the over-full buckets a uniform polynomial never shows), REPS times with idle gaps in between: the rocprofv3 target of
profiles/open_exposed.py for the step's own opening.  open_step.py [log_t] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from jolt_amd import ffi  # noqa: E402
from jolt_amd.workload import DeviceWorkload  # noqa: E402

log_t = int(sys.argv[1]) if len(sys.argv) > 1 else 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = ffi.Context(0)
wl = DeviceWorkload(ctx, log_t, pcs="grid")
wl.prepare()
wl.commit()  # the commit leg leaves the opening hint (class sums of the one-hot columns) in flight, as in the step
wl.open(label=5)
ctx.synchronize()
times = []
for _ in range(reps):
    wl.commit()
    ctx.synchronize()  # (also joins the hint stream: in the step the sums finish under the stage operators; here the opening is timed with them landed)
    time.sleep(0.05)
    t0 = time.perf_counter()
    wl.open(label=5)
    times.append(round((time.perf_counter() - t0) * 1e3, 2))
print("open ms", min(times), times)
wl.close()
ctx.close()
