"""N > 1 path on CPU ranks (gloo, world_size 2 and 4): the hypercube-sharded batched sumcheck -- partial round sums per
shard, all-gather + modular sum, shard-local rounds then the gathered tail rounds -- must reproduce the transcript of
the single-process prover over the global tables bit for bit.  The local compute is the CPU oracle here (there is no GPU
in this container); the sharding / collective / round-loop logic under test is the product code of
jolt_amd/distributed.py and jolt_amd/csrc/batch.hip."""
import os
import sys
import tempfile

import numpy as np
import pytest

from util import free_port

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, tmpdir, tail_log):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, ".."))
    from util import init_gloo
    dist = init_gloo(rank, world, port)
    import oracle_lib as O
    from jolt_amd import distributed as D
    from util import rand_fr

    log_g = world.bit_length() - 1
    n_total, n_local = 5, 5 - log_g
    N, NL = 1 << n_total, 1 << n_local
    one = O.to_mont([1])[0]
    # global inputs (same seed on every rank)
    eq_pt = rand_fr(n_total, 1)
    eq = O.eq_evals(eq_pt)
    a, b, c = rand_fr(N, 2), rand_fr(N, 3), rand_fr(N, 4)
    g = rand_fr(1, 5)[0]
    w = rand_fr(n_total, 6)
    flat = [(one, [0, 1]), (g, [0, 2])]           # eq * (a + g b), degree 2
    cubic = [(one, [0, 1, 2])]                    # a * b * c, degree 3
    coeffs = list(rand_fr(3, 7))

    lo, hi = rank * NL, (rank + 1) * NL

    class OracleShard:
        def __init__(self, members, scales):
            self.members, self.scales = members, scales

        def round(self, idx, binds):
            out = []
            for i, bnd in zip(idx, binds):
                m = self.members[i]
                s = m.round_sums(bnd, self.scales[i])
                out.append(s if m.gruen else np.concatenate([s[:1], s[2:]]))  # skipped s(1) form
            return np.concatenate(out, axis=0)

        def flush(self, binds):
            for m, bnd in zip(self.members, binds):
                if bnd is not None:
                    m.finish_rounds(bnd)

        def make_tail(self, coll, scalars):
            # every rank's partially bound tables (2^tail_log entries each), all-gathered; rank = top variable(s)
            mine = np.concatenate([m.current_tables() for m in self.members], axis=0)  # (n_tables_total, E, 4)
            assert mine.shape[1] == 1 << tail_log
            gathered = coll.all_gather_u64(mine).reshape(world, mine.shape[0], mine.shape[1], 4)
            tabs = [np.ascontiguousarray(gathered[:, k]).reshape(-1, 4) for k in range(mine.shape[0])]
            m0 = O.Member.expr(tabs[0:3], flat, 2)
            m1 = O.Member.expr(tabs[3:6], cubic, 3)
            m2 = O.Member.gruen_product(tabs[6], tabs[7], w[: log_g + tail_log], scalars[2])
            return OracleShard([m0, m1, m2], [None, None, None])

        def close(self):
            pass

    shard_scale = one.reshape(1, 4)
    for k in range(log_g):
        bit = (rank >> (log_g - 1 - k)) & 1
        f = w[k].reshape(1, 4) if bit else O.fr_sub(one.reshape(1, 4), w[k].reshape(1, 4))
        shard_scale = O.fr_mul(shard_scale, f)
    def make_local():
        return OracleShard([O.Member.expr([eq[lo:hi], a[lo:hi], b[lo:hi]], flat, 2), O.Member.expr([a[lo:hi], b[lo:hi], c[lo:hi]], cubic, 3),
                            O.Member.gruen_product(a[lo:hi], c[lo:hi], w[log_g:])], [None, None, shard_scale[0]])
    local, local_again = make_local(), make_local()
    # global reference run (every rank computes it; cheap at 2^5)
    gm = [O.Member.expr([eq, a, b], flat, 2), O.Member.expr([a, b, c], cubic, 3), O.Member.gruen_product(a, c, w)]
    claims = [m.input_claim() for m in gm]
    want = O.prove_batch(gm, claims, coeffs, [0, 0, 0], n_total, 3, label=11)
    infos = [D.MemberInfo(D.KIND_EXPR_SKIP, 2, n_total, 3), D.MemberInfo(D.KIND_EXPR_SKIP, 3, n_total, 3),
             D.MemberInfo(D.KIND_SPLIT_EQ, 3, n_total, 2, w=w)]
    coll = D.Collective(dist, world, None)
    got = D.prove_batch_sharded(None, infos, claims, coeffs, n_total, n_local, 3, world, coll, local, label=11, tail_log=tail_log)
    # the same proof with the round sums exchanged through shared memory (the default between the ranks of a node)
    shm = D.make_shm_exchange(dist, rank, world)
    assert shm is not None
    got_shm = D.prove_batch_sharded(None, infos, claims, coeffs, n_total, n_local, 3, world, coll, local_again, label=11, tail_log=tail_log,
                                    round_exchange=shm)
    shm.close()
    same = all(np.array_equal(got[k], got_shm[k]) for k in ("polys", "challenges", "final_claim", "member_claims"))
    ok = same and (np.array_equal(got["polys"], want["polys"]) and np.array_equal(got["challenges"], want["challenges"])
          and np.array_equal(got["final_claim"], want["final_claim"]) and np.array_equal(got["member_claims"], want["member_claims"]))
    open(os.path.join(tmpdir, f"rank{rank}.txt"), "w").write("ok" if ok else "MISMATCH")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,tail_log", [(2, 0), (2, 2), (4, 0), (4, 1), (4, 3), (8, 0), (8, 2)])  # 8 ranks: three hypercube variables select the rank, two stay local
def test_sharded_prove_batch_matches_single_process(world, tail_log):
    """tail_log = 0: the shards run all their local rounds and hand over single entries; tail_log > 0: early hand-over of
    2^tail_log-entry tables (fewer exchanges); tail_log = n_local (world 4, 3): everything runs in the redundant tail."""
    import util as mp  # util.spawn (standard library): the same launcher the GPU tests use
    port = free_port()
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(world, port, tmp, tail_log), nprocs=world, join=True)
        for r in range(world):
            assert open(os.path.join(tmp, f"rank{r}.txt")).read() == "ok", f"rank {r}"


def _msm_worker(rank, world, port, tmpdir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, ".."))
    from util import init_gloo
    dist = init_gloo(rank, world, port)
    import oracle_lib as O
    from jolt_amd import distributed as D
    from util import rand_fr
    n = 96
    srs = O.srs_setup_from_secret(rand_fr(1, 50)[0], n)
    scalars = rand_fr(n, 51)
    scalars[5] = 0
    lo, hi = rank * n // world, (rank + 1) * n // world
    local = O.g1_msm_pippenger(srs[lo:hi], scalars[lo:hi])  # the rank's full Pippenger over its term range (oracle on CPU ranks)
    got = D.msm_sharded(D.Collective(dist, world, None), local)
    want = O.g1_msm_pippenger(srs, scalars)
    ok = O.g1_eq(got, want) and O.g1_serialize_compressed(got) == O.g1_serialize_compressed(want)
    open(os.path.join(tmpdir, f"rank{rank}.txt"), "w").write("ok" if ok else "MISMATCH")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_msm_combines_to_the_single_process_point(world):
    """Term-range sharded MSM: per-rank partial sums, one all-gather of `world` Jacobian points, world-1 additions."""
    import util as mp  # util.spawn (standard library): the same launcher the GPU tests use
    port = free_port()
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_msm_worker, args=(world, port, tmp), nprocs=world, join=True)
        for r in range(world):
            assert open(os.path.join(tmp, f"rank{r}.txt")).read() == "ok", f"rank {r}"


def _shm_worker(rank, world, name, tmpdir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, ".."))
    from jolt_amd import distributed as D
    shm = D.ShmExchange(name, rank, world, max_bytes=4096)
    rng = np.random.default_rng(1000 + rank)
    ok = True
    for it in range(3000):  # back-to-back exchanges of varying size: the two slots per rank must never be overwritten early
        n = 1 + (it * 7) % 500
        mine = (np.arange(n, dtype=np.uint64) * np.uint64(it + 1)) ^ np.uint64(rank << 40)
        got = shm.all_gather_u64(mine)
        for r in range(world):
            ok = ok and np.array_equal(got[r], (np.arange(n, dtype=np.uint64) * np.uint64(it + 1)) ^ np.uint64(r << 40))
        if rank == it % world and it % 97 == 0:
            import time
            time.sleep(0.002 * rng.random())  # a straggler now and then
    shm.close()
    open(os.path.join(tmpdir, f"shm{rank}.txt"), "w").write("ok" if ok else "MISMATCH")


@pytest.mark.parametrize("world", [2, 4, 8])
def test_shared_memory_round_exchange(world):
    """jolt_shm_* (csrc/shm_exchange.hip): the per-round exchange between the ranks of one node, host memory only -- 3000 exchanges of
    varying size between `world` processes, stragglers included, every rank sees every rank's payload of the same exchange; an
    oversized payload and a second creator are refused."""
    import util as mp  # util.spawn (standard library): the same launcher the GPU tests use
    from jolt_amd import distributed as D
    from jolt_amd import ffi
    name = f"/jolt_test_{os.getpid()}_{world}"
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_shm_worker, args=(world, name, tmp), nprocs=world, join=True)
        for r in range(world):
            assert open(os.path.join(tmp, f"shm{r}.txt")).read() == "ok", f"rank {r}"
    solo = D.ShmExchange(name + "_solo", 0, 1, max_bytes=64)
    assert np.array_equal(solo.all_gather_u64(np.arange(8, dtype=np.uint64))[0], np.arange(8, dtype=np.uint64))
    with pytest.raises(ffi.JoltError) as e:
        solo.all_gather_u64(np.arange(9, dtype=np.uint64))  # 72 bytes > max_bytes
    assert e.value.status == 5
    solo.close()


def test_shared_memory_exchange_ignores_a_stale_segment():
    """A crashed run leaves its segment behind under the same name: magic, world and max_bytes all match.  With the per-run nonce an
    attaching rank that gets there BEFORE this run's rank 0 keeps looking until rank 0 has replaced the segment, instead of mapping
    the orphan (jolt_shm_create_nonce)."""
    import threading
    import time
    from jolt_amd import distributed as D
    name = f"/jolt_stale_{os.getpid()}"
    stale = D.ShmExchange(name, 0, 2, max_bytes=256, nonce=0x1111)  # "previous run": never closed, so the file stays in /dev/shm
    result = {}

    def late_rank_one():
        shm = D.ShmExchange(name, 1, 2, max_bytes=256, nonce=0x2222)
        result["got"] = shm.all_gather_u64(np.array([7, 8], dtype=np.uint64))
        shm.close()

    t = threading.Thread(target=late_rank_one)
    t.start()
    time.sleep(0.3)  # rank 1 is already polling and has seen the stale segment
    assert t.is_alive()
    fresh = D.ShmExchange(name, 0, 2, max_bytes=256, nonce=0x2222)
    mine = fresh.all_gather_u64(np.array([1, 2], dtype=np.uint64))
    t.join(timeout=30)
    assert not t.is_alive()
    assert np.array_equal(mine, np.array([[1, 2], [7, 8]], dtype=np.uint64)) and np.array_equal(result["got"], mine)
    fresh.close()
    stale.h = None  # its file was unlinked by the fresh run's rank 0; drop the handle without touching the new segment


def _gather_sum_worker(rank, world, port, tmpdir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, ".."))
    from util import init_gloo
    dist = init_gloo(rank, world, port)
    import oracle_lib as O
    from jolt_amd import distributed as D
    from util import rand_fr
    coll = D.Collective(dist, world)
    # every rank can rebuild every rank's share: the expected total is the oracle's sum over the ranks
    minus_one = O.fr_neg(O.to_mont([1]))[0]
    shares = []
    for r in range(world):
        v = rand_fr(37, 900 + r).copy()
        v[0] = minus_one                      # world * (r - 1): wraps world - 1 times
        v[1] = 0                              # zeros stay zero
        v[2] = O.to_mont([r + 1])[0]
        shares.append(v)
    want = shares[0].copy()
    for r in range(1, world):
        want = O.fr_add(want, shares[r])
    got = D.gather_sum(coll, shares[rank])
    ok = np.array_equal(got, want) and got.shape == want.shape
    # the shape of the caller's array comes back (the stage drivers pass (k, 3, 4) blocks of round sums)
    block = np.stack([shares[rank][:12].reshape(4, 3, 4)])[0]
    want_block = want[:12].reshape(4, 3, 4)
    ok = ok and np.array_equal(D.gather_sum(coll, block), want_block)
    with open(os.path.join(tmpdir, f"gs_{rank}.txt"), "w") as f:
        f.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_gather_sum_is_the_modular_sum_over_the_ranks(world):
    """distributed.gather_sum (what the cross-rank stage operators use for every additive T-scale quantity: uni-skip sums, pushforward masses, read-RAF scan sums):
    ONE all-gather + fr_add_vec on every rank == the oracle's field sum of the ranks' shares, wrap-arounds included, for flat and blocked arrays"""
    import util as mp  # util.spawn (standard library): the same launcher the GPU tests use
    port = 29500 + (os.getpid() % 2000) + 17 * world
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_gather_sum_worker, args=(world, port, tmp), nprocs=world, join=True)
        for r in range(world):
            assert open(os.path.join(tmp, f"gs_{r}.txt")).read() == "ok", r


def test_fr_add_vec_matches_the_oracle_on_corner_operands():
    import oracle_lib as O
    from jolt_amd import distributed as D
    from util import rand_fr
    minus_one, one, zero = O.fr_neg(O.to_mont([1]))[0], O.to_mont([1])[0], np.zeros(4, dtype=np.uint64)
    a = np.stack([minus_one, minus_one, zero, one, minus_one] + list(rand_fr(200, 77)))
    b = np.stack([minus_one, one, zero, minus_one, zero] + list(rand_fr(200, 78)))
    assert np.array_equal(D.fr_add_vec(a, b), O.fr_add(a, b))
