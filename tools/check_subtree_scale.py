"""Two ranks on ONE GPU (gloo rendezvous), 2^n_local cycles each: the subtree-sharded opening (jolt_host_hyperkzg_open_subtree) at a
size with window tables, deep recursive scans and many Horner workgroups per segment.  The proof is checked on rank 0 against the
GLOBAL polynomial through the beta-known identities (tests/kzg_check.py: folding relation, every level commitment = P_j(beta) G,
witness commitments, the combined commitment), and both ranks must return the same bytes.

    python tools/check_subtree_scale.py [n_local = 18]

Not part of the pytest suite: rank 0's check downloads the GLOBAL joint polynomial (2^(n_local + 5) coefficients), folds it and runs
every Horner pass on the oracle while the other rank waits.  The suite covers the same code against the oracle's full proof up to
2^16 coefficients (tests/test_gpu_distributed.py); tools/collect_round.sh runs this script and keeps its output under profiles/."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
from util import free_port  # noqa: E402


def _pcs_scale_worker(rank, world, port, tmpdir, n_local):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, ".."))
    from util import init_gloo
    dist = init_gloo(rank, world, port)
    import oracle_lib as O
    from jolt_amd import distributed as D
    from jolt_amd import ffi
    from kzg_check import check_opening, same_point
    ctx = ffi.Context(0)
    coll = D.Collective(dist, world, None)
    rng = np.random.default_rng(78)  # the GLOBAL raw columns, identical on every rank
    T = world << n_local
    ram = rng.integers(0, 16, size=(3, T), dtype=np.uint8)
    ram[rng.random((3, T)) < 0.4] = 0xFF
    ins = rng.integers(0, 16, size=(5, T), dtype=np.uint8)
    dense = [rng.integers(0, 2**64, size=T, dtype=np.uint64), rng.integers(-2**62, 2**62, size=T, dtype=np.int64)]
    gp, gfn, guser = D.make_point_gather(coll, world)
    pcs = D.ShardedPcs(ctx, rank, world, n_local, [ram, ins], dense, gp, gfn, guser, seed=6, fixed_base=True, subtree=True)
    out = pcs.step(label=11)
    np.savez(os.path.join(tmpdir, f"scale{rank}.npz"), com=out["open"]["com"], w=out["open"]["w"], v=out["open"]["v"], ch=out["open"]["challenges"],
             dense=out["commit"]["dense"], onehot=out["commit"]["onehot"])
    if rank == 0:  # the proof against the GLOBAL polynomial, through the identities a verifier holding beta can check
        tables = [ctx.table_from_ints(d) for d in pcs.dense_ints]
        joint = ctx.grid_joint_polynomial(pcs.sources, pcs.rlc_onehot, tables, pcs.rlc_dense, pcs.log_k)
        claimed = ctx.evaluate(joint, pcs.open_point)
        j_beta = check_opening(ctx, joint, pcs.open_point, out["open"], pcs.beta, claimed, max_download_log=pcs.grid_vars)  # every level from the oracle
        combined = O.g1_identity()
        for p in range(out["commit"]["onehot"].shape[0]):
            combined = O.g1_add(combined, O.g1_scalar_mul(out["commit"]["onehot"][p], pcs.rlc_onehot[p]))
        for d in range(out["commit"]["dense"].shape[0]):
            combined = O.g1_add(combined, O.g1_scalar_mul(out["commit"]["dense"][d], pcs.rlc_dense[d]))
        assert same_point(combined, O.g1_scalar_mul(O.g1_generator(), j_beta))
        open(os.path.join(tmpdir, "scale_ok.txt"), "w").write("ok")
    ctx.close()
    dist.destroy_process_group()


def main(n_local):
    import torch.multiprocessing as mp
    port = free_port()
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_pcs_scale_worker, args=(2, port, tmp, n_local), nprocs=2, join=True)
        assert open(os.path.join(tmp, "scale_ok.txt")).read() == "ok"
        a, b = np.load(os.path.join(tmp, "scale0.npz")), np.load(os.path.join(tmp, "scale1.npz"))
        for key in ("com", "w", "v", "ch", "dense", "onehot"):
            assert np.array_equal(a[key], b[key]), key
    print("subtree-sharded opening at 2 x 2^%d cycles: ok" % n_local)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 18)
