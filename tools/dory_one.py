"""One jolt_dory_commit_rows call (2048 x 2048 u64) and one jolt_dory_commit_onehot call -- the rocprofv3 target behind
profiles/r01_dory_tier1.txt."""
import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from jolt_amd import ffi
from tools.bench_msm import rand_fr
ctx = ffi.Context(0)
g = np.zeros(12, dtype=np.uint64)
g[0:4] = [0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f]
g[4:8] = [0xa6ba871b8b1e1b3a, 0x14f1d651eb8e167b, 0xccdd46def0f28c58, 0x1c14ef83340fbe5e]
g[8:12] = g[0:4]
width, count = 2048, 1 << 22
srs = ctx.srs_setup_from_secret(rand_fr(1, 1)[0], width, g)
rng = np.random.default_rng(7)
ints = ctx.ints(rng.integers(0, 2**64, size=count, dtype=np.uint64))
ctx.dory_commit_rows(srs, ints, width)
ctx.dory_commit_rows(srs, ints, width)
idx = rng.integers(0, 16, size=(1, count)).astype(np.uint8)
oh = ctx.onehot(idx, 16)
ctx.dory_commit_onehot(srs, oh, 0, width)
ctx.close()
