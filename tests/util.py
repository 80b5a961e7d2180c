"""Shared helpers for the parity tests (seeded inputs, shapes of real sumcheck challenges)."""
import numpy as np

P_TOP = 0x30644E72E131A029  # top u64 limb of r: any limb vector with a smaller top limb is a canonical element


def rand_fr(n, seed):
    """n uniformly random canonical Montgomery-form elements (every 256-bit value < r is a valid Montgomery form)."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    a[:, 3] = a[:, 3] % np.uint64(P_TOP)
    return a


def rand_challenge(seed, shifted=True):
    """A challenge with the reference's 125-bit shape (two low u64 Montgomery limbs zero,
    crates/jolt-field/src/bn254/mod.rs:172-184) or a full-width element."""
    c = rand_fr(1, seed)[0]
    if shifted:
        c[0] = 0
        c[1] = 0
        c[3] &= np.uint64((1 << 61) - 1)
    return c


def small_fr(vals, oracle):
    return oracle.fr_from_u64(np.asarray(vals, dtype=np.uint64))


def free_port():
    """a TCP port nobody listens on right now (the multi-process tests used to derive ports from the pid: a collision hangs the rendezvous)"""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def init_gloo(rank, world, port, seconds=120):
    """gloo process group on 127.0.0.1 with a bounded rendezvous / collective timeout: a rank that died must fail the test, not hang it"""
    import datetime
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=seconds))
    return dist


def stack_dump_later(tag, seconds):
    """faulthandler watchdog for the multi-process tests: if this process is still alive after `seconds`, every thread's Python stack goes to
    gpurun_out/stacks/<tag>.txt (kept by gpurun) -- tells a hang in an exchange from plain slowness.  The process is not killed; the test's own timeouts do that."""
    import faulthandler
    import os
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "stacks")
    os.makedirs(d, exist_ok=True)
    f = open(os.path.join(d, f"{tag}.txt"), "w")
    faulthandler.dump_traceback_later(seconds, repeat=False, file=f, exit=False)
    return f


def _spawn_entry(fn, rank, args):
    fn(rank, *args)


def spawn(fn, args=(), nprocs=1, join=True, timeout=None):  # timeout None: 1800 s
    """torch.multiprocessing.spawn's contract -- fn(rank, *args) in `nprocs` fresh interpreters, an exception if any of them fails -- on the standard library, so that the
    pytest process itself never imports torch: a GPU test process that holds torch (its bundled HIP runtime, RCCL and rocm_smi) NEXT TO libjolt_hip.so's system ones
    aborted at exit in round 4 (profiles/r05_teardown_abort_backtrace.txt).  The workers import torch first (init_gloo) and libjolt_hip.so after it: one runtime each."""
    import multiprocessing
    ctx = multiprocessing.get_context("spawn")
    procs = [ctx.Process(target=_spawn_entry, args=(fn, r, tuple(args)), daemon=False) for r in range(nprocs)]
    for p in procs:
        p.start()
    if not join:
        return procs
    # poll all ranks together: the first rank that exits non-zero (or the deadline) ends the others at once -- a sibling blocked in a collective would otherwise
    # sit out the gloo / shared-memory timeout (120 - 300 s) before the test failed
    import time
    deadline = time.monotonic() + (timeout if timeout is not None else 1800.0)
    failed = []
    while True:
        alive = [p for p in procs if p.is_alive()]
        failed = [(r, p.exitcode) for r, p in enumerate(procs) if not p.is_alive() and p.exitcode != 0]
        if failed or not alive:
            break
        if time.monotonic() > deadline:
            failed = [(r, "timeout") for r, p in enumerate(procs) if p.is_alive()]
            break
        time.sleep(0.05)
    if failed:
        for p in procs:
            if p.is_alive():
                p.kill()
        for p in procs:
            p.join()
        raise RuntimeError(f"spawned rank(s) failed: {failed}")
    for p in procs:
        p.join()
    return None
