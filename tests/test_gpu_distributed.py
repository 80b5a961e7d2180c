"""The hypercube-sharded bench workload on real device members: two ranks (both on GPU 0, gloo for the tiny
collectives since the box has one GPU) must reproduce the transcript of the CPU oracle proving the GLOBAL tables in one
process.  Covers the device path of jolt_amd/distributed.py: aligned eq blocks, shard-scaled split-eq tables, grouped
final values, the gathered tail arena and its members."""
import os
import sys
import tempfile

import numpy as np
import pytest

from util import free_port

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, tmpdir, n_local, tail_log):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, ".."))
    from util import init_gloo
    dist = init_gloo(rank, world, port)
    from jolt_amd import distributed as D
    from jolt_amd import ffi
    ctx = ffi.Context(0)
    coll = D.Collective(dist, world, None)
    wl = D.ShardedWorkload(ctx, n_local, rank, world, dist, seed=31, coll=coll, tail_log=tail_log)
    outs = [wl.prove(label=50), wl.prove(label=50)]  # second pass reuses the cached tail members
    if rank == 0:
        np.savez(os.path.join(tmpdir, "got.npz"), **{f"{p}_{st}_{k}": v for p, o in enumerate(outs) for st, d in o.items() for k, v in d.items()})
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_local,tail_log", [(2, 6, 0), (2, 6, 3), (2, 6, 6), (4, 5, 2), (8, 5, 1)])
def test_sharded_device_workload_matches_global_oracle(world, n_local, tail_log):
    """tail_log 0: all local rounds sharded, single entries handed over; 3: early hand-over of 8-entry tables (packed,
    gathered, interleaved on the device); 6 = n_local: everything in the redundant tail.  World 4 and 8 (all ranks on GPU 0): two and
    three hypercube variables select the rank -- more tail rounds, the 8-rank slots of the shared-memory round exchange."""
    import util as mp  # util.spawn: torch.multiprocessing.spawn's contract without torch in this process
    import oracle_lib as O
    from jolt_amd import distributed as D
    from jolt_amd import workload as W
    port = free_port()
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(world, port, tmp, n_local, tail_log), nprocs=world, join=True)
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
        elif t["kind"] == "onehot":  # address-folded selector column: eq(chunk point, .)[hot index], zero on cold cycles
            scale_table = O.eq_evals(t["point"])
            idx = np.concatenate([specs[r]["tables"][name]["data"] for r in range(world)])
            col = np.zeros((idx.shape[0], 4), dtype=np.uint64)
            hot = idx != 0xFF
            col[hot] = scale_table[idx[hot]]
            tables[name] = col
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
        for p in (0, 1):  # the two passes of rank 0 (the second one reuses the cached tail members); every rank draws the same transcript
            assert np.array_equal(got[f"{p}_{stage}_polys"], want["polys"]), (p, stage)
            assert np.array_equal(got[f"{p}_{stage}_challenges"], want["challenges"]), (p, stage)
            assert np.array_equal(got[f"{p}_{stage}_final_claim"], want["final_claim"]), (p, stage)


def test_native_rccl_communicator_single_rank():
    """The RCCL communicator owned by the library (dlopen'd librccl, ncclCommInitRank, ncclAllGather on the context stream)
    with world = 1 -- the only size a one-GPU box can run: host and table all-gathers are identities, and the sharded
    workload proved through it (native gather hook + device hand-over) gives the same transcript for every tail_log."""
    from jolt_amd import distributed as D
    from jolt_amd import ffi
    from util import rand_fr
    ctx = ffi.Context(0)
    coll = D.NativeCollective(ctx, None, 0, 1)
    x = rand_fr(37, 5)
    assert np.array_equal(coll.all_gather_u64(x).reshape(-1, 4), x)
    big = rand_fr(5000, 6)  # forces the staging buffers to grow
    assert np.array_equal(coll.all_gather_u64(big).reshape(-1, 4), big)
    src, dst = ctx.upload(big), ctx.alloc(5000)
    coll.all_gather_table(src, 5000, dst)
    assert np.array_equal(dst.download(), big)
    # term-range sharded MSM through the same communicator (world = 1: the partial sum is the total)
    import oracle_lib as O
    srs_host = O.srs_setup_from_secret(rand_fr(1, 60)[0], 300)
    sc = rand_fr(300, 61)
    part = ctx.msm(ctx.srs_upload(srs_host), sc)
    assert O.g1_eq(D.msm_sharded(coll, part), O.g1_msm_pippenger(srs_host, sc))
    outs = []
    for tail_log in (0, 4, 7):
        wl = D.ShardedWorkload(ctx, 7, 0, 1, None, seed=33, coll=coll, tail_log=tail_log, force_gather=True)
        outs.append(wl.prove(label=3))
    for o in outs[1:]:
        for stage in outs[0]:
            for k in ("polys", "challenges", "member_claims", "final_claim"):
                assert np.array_equal(o[stage][k], outs[0][stage][k]), (stage, k)
    coll.close()
    ctx.close()


def _msm_worker(rank, world, port, tmpdir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, ".."))
    from util import init_gloo
    dist = init_gloo(rank, world, port)
    import oracle_lib as O
    from jolt_amd import distributed as D
    from jolt_amd import ffi
    from util import rand_fr
    n = 4096
    ctx = ffi.Context(0)
    beta = rand_fr(1, 70)[0]
    full = ctx.srs_setup_from_secret(beta, n, O.g1_generator()).download()  # beta^i * G, same on every rank
    scalars = rand_fr(n, 71)
    lo, hi = rank * n // world, (rank + 1) * n // world
    part = ctx.msm(ctx.srs_upload(full[lo:hi]), ctx.upload(scalars[lo:hi]))  # device Pippenger over the rank's term range
    got = D.msm_sharded(D.Collective(dist, world, None), part)
    # size-independent identity: the commitment to p with bases beta^i G is p(beta) G
    want = O.g1_scalar_mul(O.g1_generator(), O.kzg_eval_univariate(scalars, beta))
    ok = O.g1_eq(got, want)
    open(os.path.join(tmpdir, f"msm{rank}.txt"), "w").write("ok" if ok else "MISMATCH")
    ctx.close()
    dist.destroy_process_group()


def test_sharded_device_msm_two_ranks():
    import util as mp  # util.spawn: torch.multiprocessing.spawn's contract without torch in this process
    port = free_port()
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_msm_worker, args=(2, port, tmp), nprocs=2, join=True)
        for r in range(2):
            assert open(os.path.join(tmp, f"msm{r}.txt")).read() == "ok", f"rank {r}"


def _pcs_worker(rank, world, port, tmpdir, n_local, mode):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, ".."))
    from util import init_gloo
    dist = init_gloo(rank, world, port)
    from jolt_amd import distributed as D
    from jolt_amd import ffi
    ctx = ffi.Context(0)
    coll = D.Collective(dist, world, None)
    rng = np.random.default_rng(77)  # the GLOBAL raw columns, identical on every rank
    T = world << n_local
    ram = rng.integers(0, 16, size=(3, T), dtype=np.uint8)
    ram[rng.random((3, T)) < 0.4] = 0xFF
    ins = rng.integers(0, 16, size=(5, T), dtype=np.uint8)
    dense = [rng.integers(0, 2**64, size=T, dtype=np.uint64), rng.integers(-2**62, 2**62, size=T, dtype=np.int64)]
    gp, gfn, guser = D.make_point_gather(coll, world)
    pcs = D.ShardedPcs(ctx, rank, world, n_local, [ram, ins], dense, gp, gfn, guser, seed=5, fixed_base=(n_local >= 6), block_cyclic=(mode != "range"),
                       subtree=(mode == "subtree"))
    out = pcs.step(label=9)
    again = pcs.step(label=9)
    assert np.array_equal(out["open"]["v"], again["open"]["v"])
    np.savez(os.path.join(tmpdir, f"pcs{rank}.npz"), dense=out["commit"]["dense"], onehot=out["commit"]["onehot"], com=out["open"]["com"], w=out["open"]["w"],
             v=out["open"]["v"], ch=out["open"]["challenges"], beta=pcs.beta, rlc_onehot=pcs.rlc_onehot, rlc_dense=pcs.rlc_dense, point=pcs.open_point,
             ram=ram, ins=ins, d0=dense[0], d1=dense[1])
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_local,mode", [(2, 4, "subtree"), (4, 4, "subtree"), (2, 7, "subtree"), (2, 11, "subtree"), (4, 9, "subtree"), (8, 8, "subtree"),
                                                (2, 4, "cyclic"), (4, 9, "cyclic"), (8, 5, "cyclic"), (2, 4, "range"), (4, 4, "range"), (2, 11, "range")])
def test_sharded_commit_and_open_equal_the_single_process_proof(world, n_local, mode):
    """Sharded PCS legs (ShardedPcs): every rank returns the commitments and the opening the ORACLE computes in one process over the
    global trace -- same transcript bytes, same points.
      subtree (the default): commitments by block-cyclic shares; the opening with the POLYNOMIAL sharded by index subtree -- folds, RLC,
        Horner passes, quotient scans and MSMs on 1 / world of the coefficients per rank, O(ell) field elements exchanged
        (jolt_host_hyperkzg_open_subtree; window tables over the rank's own bases from n_local = 8 on);
      cyclic: polynomial arithmetic replicated, every MSM split block-cyclically over per-rank compact bases;
      range: the same with contiguous term ranges against the full SRS."""
    import util as mp  # util.spawn: torch.multiprocessing.spawn's contract without torch in this process
    import oracle_lib as O
    port = free_port()
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_pcs_worker, args=(world, port, tmp, n_local, mode), nprocs=world, join=True)
        got = [np.load(os.path.join(tmp, f"pcs{r}.npz")) for r in range(world)]
    g0 = got[0]
    T, K = world << n_local, 16
    host_srs = O.srs_setup_from_secret(g0["beta"], K * T)
    one = O.to_mont([1])[0]
    joint = np.zeros((K * T, 4), dtype=np.uint64)
    cols = list(g0["ram"]) + list(g0["ins"])
    for p, col in enumerate(cols):
        emb = np.zeros((K * T, 4), dtype=np.uint64)
        hot = col != 0xFF
        emb[col[hot].astype(np.int64) * T + np.nonzero(hot)[0]] = one
        assert O.g1_eq(g0["onehot"][p], O.kzg_commit(emb, host_srs)), p
        joint = O.fr_add(joint, O.fr_mul(emb, np.repeat(g0["rlc_onehot"][p].reshape(1, 4), K * T, axis=0)))
    for d, vals in enumerate([O.fr_from_u64(g0["d0"]), O.fr_from_i64(g0["d1"])]):
        emb = np.zeros((K * T, 4), dtype=np.uint64)
        emb[:T] = vals
        assert O.g1_eq(g0["dense"][d], O.kzg_commit(emb, host_srs)), d
        joint[:T] = O.fr_add(joint[:T], O.fr_mul(vals, np.repeat(g0["rlc_dense"][d].reshape(1, 4), T, axis=0)))
    want = O.hyperkzg_open(host_srs, joint, g0["point"], label=9)
    for r in range(world):
        g = got[r]
        assert np.array_equal(g["ch"], want["challenges"]) and np.array_equal(g["v"], want["v"]), r
        for a, b in zip(list(g["com"]) + list(g["w"]), list(want["com"]) + list(want["w"])):
            assert O.g1_eq(a, b), r
        for key in ("dense", "onehot"):
            assert all(O.g1_eq(a, b) for a, b in zip(g[key], g0[key]))


def _failing_rank_worker(rank, world, port, out_path):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, ".."))
    from util import init_gloo
    dist = init_gloo(rank, world, port)
    from jolt_amd import distributed as D
    from jolt_amd import ffi
    ctx = ffi.Context(0)
    coll = D.Collective(dist, world, None)
    rng = np.random.default_rng(79)
    n_local = 4
    T = world << n_local
    ins = rng.integers(0, 16, size=(3, T), dtype=np.uint8)
    dense = [rng.integers(0, 2**64, size=T, dtype=np.uint64)]
    gp, gfn, guser = D.make_point_gather(coll, world)
    pcs = D.ShardedPcs(ctx, rank, world, n_local, [ins], dense, gp, gfn, guser, seed=5, fixed_base=False, block_cyclic=True, subtree=False)
    if rank == 1:  # this rank's compact SRS is too short for its share of the terms: a LOCAL failure inside the opening's first sharded MSM
        short = pcs.srs.download(0, len(pcs.srs) // 2)
        pcs.srs = ctx.srs_upload(short)
    outcome = "returned"
    try:
        pcs.open(label=3)
    except ffi.JoltError as e:
        outcome = f"JoltError {e.status}"
    with open(f"{out_path}.{rank}", "w") as f:
        f.write(outcome)
    ctx.close()
    dist.destroy_process_group()


def test_a_local_failure_in_the_sharded_opening_reaches_every_rank():
    """One rank fails inside the opening (its SRS is too short); the status word every exchange carries makes BOTH ranks return that error instead of
    the healthy rank blocking in the collective (which the shared-memory exchange would time out of and RCCL would hang in)."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import test_gpu_distributed as t, util as mp; "
            "mp.spawn(t._failing_rank_worker, args=(2, %d, %r), nprocs=2, join=True)")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "outcome")
        try:
            r = subprocess.run([sys.executable, "-c", code % (HERE, free_port(), out)], capture_output=True, text=True, timeout=180)
        except subprocess.TimeoutExpired:
            pytest.fail("the healthy rank hung in a collective the failing rank never entered")
        outcomes = [open(f"{out}.{k}").read() if os.path.exists(f"{out}.{k}") else f"no outcome (rc {r.returncode}): {r.stderr[-400:]}" for k in range(2)]
    assert outcomes == ["JoltError 9", "JoltError 9"], outcomes  # JOLT_ERR_SRS_TOO_SMALL on both


@pytest.mark.parametrize("world,scale,steps", [(2, 10, 2), (4, 10, 2), (8, 12, 1), (8, 14, 1)])  # 8 x 2^14: configs[3]'s own shape (`--gpus 8`), as large as 8 ranks sharing one GPU finish in seconds
def test_bench_multi_rank_path_runs_end_to_end_on_one_gpu(world, scale, steps):
    """`python bench.py --gpus N` with NO launcher around it (the driver's bare command shape): the script re-executes itself under
    torch.distributed.run with N ranks and rank 0 prints one JSON line with n_gpus = N.  JOLT_BENCH_SHARE_GPU=1 puts every rank on device 0
    with gloo as the rendezvous backend (this box has one GPU; RCCL needs one device per rank): the N > 1 code path of bench.py -- sharded
    prepare + commit + prove + open per step, both round exchanges timed, max-over-ranks timing -- minus RCCL itself."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["JOLT_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--scale", str(scale), "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]  # torchrun merges the ranks' stdout: nobody but rank 0 may print a result line
    # gloo's own connection banner ("[Gloo] Rank k is connected to N peer ranks. Expected number of connected peer ranks is : N") goes to stdout too, and the ranks'
    # banners interleave mid-line under torchrun: every line that is not the result must be made of that banner's characters (a stray "3", half a sentence) --
    # what this guards against is a rank printing something of its own
    banner = set("[Gloo] Rank is connected to peer ranks. Expected number of connected peer ranks is : 0123456789")
    for l in r.stdout.splitlines():
        assert l.startswith("{") or set(l) <= banner, l[:200]
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["steps"] == steps and line["scaling"] == "weak"
    assert line["config"]["trace_length_total"] == world << scale
    assert line["value"] == pytest.approx(world * (1 << scale) / (line["ms_per_step"] * 1e-3), rel=1e-3)
    cfg = line["config"]
    assert cfg["trace_length_per_gpu"] == 1 << scale and "configs[2] sharded" in cfg["workload"] and "every step rebuilds" in cfg["workload"]
    assert cfg["round_exchange"].startswith(("rccl", "torch")) and f"{world} rank(s)" in cfg["communicator"] and "pcs" in cfg
    ab = cfg["round_exchange_ab"]  # both exchanges of the per-round sums timed in this one run; `value` is the first key's
    assert set(ab) == {"rccl", "shm"} and list(ab)[0] == "rccl" and ab["rccl"] == pytest.approx(line["ms_per_step"], rel=1e-6)
    assert line["roofline"]["frac"] > 0 and line["roofline"]["traffic_source"]


def _rccl_same_device_worker(rank, world, port, out_path):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, ".."))
    from util import init_gloo
    dist = init_gloo(rank, world, port)
    from jolt_amd import distributed as D
    from jolt_amd import ffi
    ctx = ffi.Context(0)
    outcome = "created"
    try:
        coll = D.NativeCollective(ctx, dist, rank, world, None)
        got = coll.all_gather_u64(np.arange(4, dtype=np.uint64) + np.uint64(10 * rank))
        outcome = "all-gather ok" if np.array_equal(got, np.stack([np.arange(4, dtype=np.uint64) + np.uint64(10 * r) for r in range(world)])) else "all-gather WRONG"
        coll.close()
    except Exception as e:  # noqa: BLE001 -- the outcome is the test's subject
        outcome = f"refused: {type(e).__name__}: {e}"
    with open(f"{out_path}.{rank}", "w") as f:
        f.write(outcome)
    ctx.close()


def test_native_rccl_with_two_ranks_on_one_device_is_refused_or_works_but_never_hangs():
    """RCCL with N > 1 ranks needs one device per rank; this box has one.  Two ranks on device 0 try jolt_comm_create: RCCL is expected to
    refuse duplicate devices (then ShardedWorkload's collective decision falls back on every rank), it must not hang, and if it ever
    accepts them the all-gather must be right.  The outcome is printed and kept under gpurun_out/ for DESIGN.md section 6."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import test_gpu_distributed as t, util as mp; "
            "mp.spawn(t._rccl_same_device_worker, args=(2, %d, %r), nprocs=2, join=True)")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "outcome")
        port = free_port()
        try:
            r = subprocess.run([sys.executable, "-c", code % (HERE, port, out)], capture_output=True, text=True, timeout=120)
        except subprocess.TimeoutExpired:
            pytest.fail("jolt_comm_create with two ranks on one device hung")
        outcomes = [open(f"{out}.{k}").read() if os.path.exists(f"{out}.{k}") else f"no outcome (rc {r.returncode}): {r.stderr[-300:]}" for k in range(2)]
    print("RCCL, two ranks on device 0:", outcomes)
    keep = os.path.join(HERE, "..", "gpurun_out")
    os.makedirs(keep, exist_ok=True)
    with open(os.path.join(keep, "rccl_two_ranks_one_device.txt"), "w") as f:
        f.write("\n".join(outcomes) + "\n")
    assert all(o.startswith(("refused", "all-gather ok")) for o in outcomes), outcomes


def _extended_worker(rank, world, port, tmpdir, n_local, kw):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, ".."))
    import pickle
    from util import init_gloo, stack_dump_later
    stack_dump_later(f"extended_w{world}_r{rank}", 240)  # a hang at N ranks leaves every rank's stack under gpurun_out/stacks/ instead of a silent timeout
    dist = init_gloo(rank, world, port, seconds=300)
    from jolt_amd import distributed as D
    from jolt_amd import ffi
    from jolt_amd.stages_sharded import ShardedExtended
    ctx = ffi.Context(0)
    coll = D.Collective(dist, world, None)
    ext = ShardedExtended(ctx, n_local, rank, world, coll, seed=77, **kw)
    outs = [ext.prove(label=40) for _ in range(1 if world >= 8 else 2)]  # a second proof over the resident inputs: same bytes (8 ranks time-slicing one GPU: one proof, the
    # two-proof run took 134 s, profiles/r05_pytest_8rank_stage_operators.txt)
    with open(os.path.join(tmpdir, f"got{rank}.pkl"), "wb") as f:
        pickle.dump(dict(outs=outs, claims=ext.claims), f)
    ext.close()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_local,kw", [(2, 6, dict(n_tables=8, log_k=5, log_kb=5)), (4, 5, dict(n_tables=6, log_k=4, log_kb=4)), (2, 10, dict(n_tables=12, log_k=8)),
                                              (1, 6, dict(n_tables=5, log_k=4)), (8, 4, dict(n_tables=5, log_k=4, log_kb=4))])
def test_sharded_stage_operators_prove_one_trace(world, n_local, kw):
    """The stage 1 / 2 / 4 / 5 / 6a / 6b / 7 operators over ONE trace of world * 2^n_local cycles dealt to `world` ranks (jolt_amd/stages_sharded.py; all ranks on GPU 0,
    gloo standing in for RCCL): every rank's messages -- uni-skip sums, every round polynomial of every operator (the sparse matrices' local, merged-cycle and address
    rounds, the 128 read-RAF address rounds, the sharded cycle phases), challenges, claims, final values -- equal the single-process oracle twin of the GLOBAL trace."""
    import pickle
    import util as mp  # util.spawn: torch.multiprocessing.spawn's contract without torch in this process
    from jolt_amd import stages as S
    from test_gpu_extended import same
    from workload_oracle import OracleExtended
    port = free_port()
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_extended_worker, args=(world, port, tmp, n_local, kw), nprocs=world, join=True)
        got = [pickle.load(open(os.path.join(tmp, f"got{r}.pkl"), "rb")) for r in range(world)]
    n_total = n_local + world.bit_length() - 1
    want = OracleExtended(n_total, description=S.build_extended(n_total, seed=77, n_blocks=world, **kw)).prove(label=40)
    want.pop("booleanity_cycle")  # single-rank operator (jolt_stage_booleanity_cycle_create): not in the sharded driver
    address_domain = {"bytecode_read_raf", "ram_raf_evaluation", "ram_output_check", "hamming_weight"}
    claim_key = {"spartan_outer": "outer", "spartan_product": "product", "ram_read_write": "ram", "registers_read_write": "registers", "instruction_read_raf": "lookup"}
    for r in range(world):
        for p, out in enumerate(got[r]["outs"]):
            assert set(out) == set(want)
            for name in out:
                if name in claim_key:
                    assert np.array_equal(got[r]["claims"][claim_key[name]], want[name]["claim"]), (r, name)
                same(out[name], {k: v for k, v in want[name].items() if k != "claim" or name in address_domain}, f"rank {r} proof {p} {name}")
