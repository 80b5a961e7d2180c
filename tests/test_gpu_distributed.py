"""The hypercube-sharded bench workload on real device members: two ranks (both on GPU 0, gloo for the tiny
collectives since the box has one GPU) must reproduce the transcript of the CPU oracle proving the GLOBAL tables in one
process.  Covers the device path of jolt_amd/distributed.py: aligned eq blocks, shard-scaled split-eq tables, grouped
final values, the gathered tail arena and its members."""
import os
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, tmpdir, n_local):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, ".."))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jolt_amd import distributed as D
    from jolt_amd import ffi
    ctx = ffi.Context(0)
    coll = D.Collective(dist, world, None)
    wl = D.ShardedWorkload(ctx, n_local, rank, world, dist, seed=31, coll=coll)
    outs = [wl.prove(label=50), wl.prove(label=50)]  # second pass reuses the cached tail members
    if rank == 0:
        np.savez(os.path.join(tmpdir, "got.npz"), **{f"{p}_{st}_{k}": v for p, o in enumerate(outs) for st, d in o.items() for k, v in d.items()})
    ctx.close()
    dist.destroy_process_group()


def test_sharded_device_workload_matches_global_oracle():
    import torch.multiprocessing as mp
    import oracle_lib as O
    from jolt_amd import distributed as D
    from jolt_amd import workload as W
    world, n_local = 2, 6
    port = 29800 + os.getpid() % 1000
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(world, port, tmp, n_local), nprocs=world, join=True)
        got = np.load(os.path.join(tmp, "got.npz"))
    # ---- global reference on the CPU oracle
    specs = [D.build_sharded_spec(n_local, r, world, seed=31) for r in range(world)]
    n_total = specs[0]["n_total"]
    one = O.to_mont([1])[0]
    mul = lambda a, b: O.fr_mul(np.asarray(a).reshape(1, 4), np.asarray(b).reshape(1, 4))[0]
    neg = lambda a: O.fr_neg(np.asarray(a).reshape(1, 4))[0]
    res = W.Resolver(specs[0]["gammas"], one, mul, neg)
    tables = {}
    for name, t in specs[0]["tables"].items():
        if t["kind"] == "eqblock":
            tables[name] = O.eq_evals(t["point"])
        else:
            conv = O.fr_from_u64 if t["kind"] == "u64" else O.fr_from_i64
            tables[name] = np.concatenate([conv(specs[r]["tables"][name]["data"]) for r in range(world)], axis=0)
    stages = {}
    for k, ms in enumerate(specs[0]["members"]):
        stages.setdefault(ms.stage, []).append(k)
    for stage, idxs in sorted(stages.items()):
        members = []
        for k in idxs:
            ms = specs[0]["members"][k]
            tabs = [tables[t] for t in ms.tables]
            if ms.split_eq is not None:
                a, b, _ = ms.split_eq
                members.append(O.Member.gruen_product(tabs[a], tabs[b], specs[0]["split_points"][k]))
            else:
                members.append(O.Member.expr(tabs, W.expand_to_flat_terms(res.groups(ms.groups), mul, one), ms.degree))
        claims = [m.input_claim() for m in members]
        deg = max(m.degree for m in members)
        want = O.prove_batch(members, claims, [specs[0]["batch_coeffs"][k] for k in idxs], [0] * len(idxs), n_total, deg, label=50 + stage)
        for p in (0, 1):
            assert np.array_equal(got[f"{p}_{stage}_polys"], want["polys"]), (p, stage)
            assert np.array_equal(got[f"{p}_{stage}_challenges"], want["challenges"]), (p, stage)
            assert np.array_equal(got[f"{p}_{stage}_final_claim"], want["final_claim"]), (p, stage)
