import sys, time
sys.path.insert(0, '.')
from jolt_amd import ffi
from jolt_amd.workload import DeviceWorkload
ctx = ffi.Context(0)
wl = DeviceWorkload(ctx, int(sys.argv[1]) if len(sys.argv) > 1 else 20)
for _ in range(3): wl.prove(label=1)
acc = {}
N = 10
for it in range(N):
    for stage, idxs in sorted(wl.stages.items()):
        ms = [wl.members[i] for i in idxs]
        deg = max(m.degree for m in ms)
        ctx.synchronize(); t0 = time.perf_counter()
        ctx.prove_batch(ms, [wl.claims[i] for i in idxs], [wl.batch_coeffs[i] for i in idxs], [0] * len(ms), wl.n_vars, deg, label=5 + stage, use_round_group=True)
        ctx.synchronize(); acc[stage] = acc.get(stage, 0) + time.perf_counter() - t0
    for m in wl.members: m.reset()
print({k: round(v / N * 1e3, 3) for k, v in acc.items()}, 'sum', round(sum(acc.values()) / N * 1e3, 3))
print({st: [wl.members_spec[i].name for i in idxs] for st, idxs in wl.stages.items()})
