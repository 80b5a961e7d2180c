"""Hypercube-sharded batched sumcheck: one process per GPU, rank g owns block g of every table (DESIGN.md section 6,
SURVEY.md section 8e).

A batch over n_total = n_local + log2(G) variables is proved in two phases through the resumable round loop of
jolt_amd/csrc/batch.hip:
  phase A  rounds 0 .. n_local-1: every rank computes the round sums of its block (LowToHigh binds pair (2y, 2y+1), so
           they stay local); ONE all-gather of a few hundred bytes per batch round, modular sum on every rank, identical
           transcripts -> identical challenges;
  phase B  once the local tables are down to 2^tail_log entries they are all-gathered once, each rank rebuilds the
           G * 2^tail_log-entry tables (rank = top variables) and finishes the last tail_log + log2(G) rounds
           redundantly, without communication.  The per-round exchange is latency-bound (a few hundred bytes), and the
           late rounds are tiny, so handing over early halves the number of exchanges; tail_log is chosen so that the
           tail still fits the single-launch tail kernel (G * 2^tail_log <= 8192 entries).
The collectives are RCCL over xGMI: called natively on the context's stream (jolt_amd/csrc/comm.hip, `NativeCollective`,
rendezvous through torch.distributed) on the GPU box, or through torch.distributed itself (`Collective`: "nccl", or "gloo"
in the CPU tests).  RCCL has no mod-r reduction, hence all-gather + local modular sum.

The local compute is pluggable so that the N>1 logic can be tested on CPU ranks: `DeviceShard` drives device members
through the C ABI; tests substitute an oracle-backed shard.
"""
import ctypes as C
import os

import numpy as np

from . import ffi
from . import workload as W

KIND_EXPR, KIND_EXPR_SKIP, KIND_SPLIT_EQ, KIND_SPLIT_EQ_UNIFORM = 0, 1, 2, 3

LOCAL_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_size_t), C.c_size_t, C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t)
GATHER_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class ShmExchange:
    """The per-round exchange of partial round sums between the ranks of one node through POSIX shared memory (jolt_shm_*,
    csrc/shm_exchange.hip): a memcpy and a sequence number per rank instead of an RCCL all-gather on the critical path of every
    sharded round.  Host memory only, so it also runs (and is tested) without a GPU.  `name` must be the same on every rank."""
    MAX_BYTES = 64 * 1024

    def __init__(self, name, rank, world, max_bytes=MAX_BYTES, nonce=0):
        self.rank, self.world = rank, world
        h = C.c_void_p()
        st = ffi.lib().jolt_shm_create_nonce(name.encode(), C.c_uint64(nonce), C.c_int32(rank), C.c_int32(world), C.c_size_t(max_bytes), C.byref(h))
        if st:
            raise ffi.JoltError(st, "jolt_shm_create_nonce")
        self.h = h

    def all_gather_u64(self, arr):
        flat = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1)
        out = np.empty(self.world * flat.size, dtype=np.uint64)
        st = ffi.lib().jolt_shm_all_gather(self.h, _p(flat), C.c_size_t(flat.nbytes), _p(out))
        if st:
            raise ffi.JoltError(st, "jolt_shm_all_gather")
        return out.reshape(self.world, -1)

    def close(self):
        if self.h:
            ffi.lib().jolt_shm_destroy(self.h)
            self.h = None


def make_shm_exchange(dist, rank, world):
    """Collective: every rank attaches to one segment (name drawn by rank 0, carried by torch.distributed) and the ranks agree --
    all or none -- that it works (a probe exchange included).  Returns a ShmExchange or None (then the RCCL exchange stays)."""
    if os.environ.get("JOLT_ROUND_EXCHANGE", "shm") != "shm":
        return None
    name = [f"/jolt_{os.getpid()}_{int.from_bytes(os.urandom(4), 'little'):08x}", int.from_bytes(os.urandom(8), "little") | 1]  # name, per-run nonce
    if world > 1:
        dist.broadcast_object_list(name, src=0)
    shm = None
    try:
        shm = ShmExchange(name[0], rank, world, nonce=name[1])
        probe = np.arange(4, dtype=np.uint64) + np.uint64(100 * rank)
        got = shm.all_gather_u64(probe)
        ok = all(np.array_equal(got[r], np.arange(4, dtype=np.uint64) + np.uint64(100 * r)) for r in range(world))
    except Exception:  # noqa: BLE001 -- decided collectively below
        ok = False
    if world > 1:
        flags = [None] * world
        dist.all_gather_object(flags, bool(ok))
        ok = all(flags)
    if not ok:
        if shm is not None:
            shm.close()
        return None
    return shm


class Collective:
    """all_gather of small uint64 arrays over torch.distributed (device tensors for nccl, host tensors for gloo)."""

    def __init__(self, dist, world, device=None):
        self.dist, self.world, self.device = dist, world, device

    def all_gather_u64(self, arr):
        import torch
        flat = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1)
        t = torch.from_numpy(flat.view(np.int64).copy())
        if self.device is not None:
            t = t.to(self.device)
        out = torch.empty(self.world * t.numel(), dtype=torch.int64, device=t.device)
        self.dist.all_gather_into_tensor(out, t)
        return out.cpu().numpy().view(np.uint64).reshape(self.world, -1)


def _system_rccl_path():
    """The librccl the native communicator opens when the process holds none yet (comm.hip: rccl_api): the system one next to the HIP runtime libjolt_hip.so uses.

    Round 5: an RCCL that is ALREADY in the process wins -- in a process that imported torch first (bench.py --gpus N) both libjolt_hip.so and torch resolve
    libamdhip64.so.7 to torch's bundled copy (same SONAME), torch's librccl is loaded with it, and loading the system librccl beside it would bring a second rocm_smi
    whose globals interpose with the first's (the round-4 teardown abort, profiles/r05_teardown_abort_backtrace.txt).  A process with TWO HIP runtimes (torch imported
    after libjolt_hip.so was loaded) gets no native communicator at all.  JOLT_RCCL_PATH overrides the path of the fallback."""
    env = os.environ.get("JOLT_RCCL_PATH")
    if env:
        return env
    hip_dir = None
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.split()[-1] if line.strip() else ""
                if "libamdhip64" in path and "/torch/" not in path:
                    hip_dir = os.path.dirname(path)
                    break
    except OSError:
        pass
    for d in ([hip_dir] if hip_dir else []) + [os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")]:
        cand = os.path.join(d, "librccl.so.1")
        if os.path.exists(cand):
            return cand
    return None


class NativeCollective:
    """RCCL communicator owned by libjolt_hip.so (jolt_comm_*): rank 0 draws the ncclUniqueId, torch.distributed
    broadcasts it, every rank joins.  The round loop then calls RCCL without going through Python.

    Which RCCL: the one the process already holds (torch's bundled copy in a process that imported torch), else the system one (`_system_rccl_path`); comm.hip decides.
    ShardedWorkload falls back to the torch.distributed collective (collectively, on every rank) when the native communicator
    cannot be created."""

    def __init__(self, ctx, dist, rank, world, device=None):
        lib = ffi.lib()
        self.ctx, self.rank, self.world = ctx, rank, world
        path = _system_rccl_path()
        self._path = path.encode() if path else None
        uid = (C.c_uint8 * 128)()
        if rank == 0:
            st = lib.jolt_comm_unique_id(self._path, uid)
            if st:
                raise ffi.JoltError(st, "jolt_comm_unique_id")
        if world > 1:  # torch.distributed carries the 128 bytes to the other ranks (and is already imported by the launcher)
            import torch
            t = torch.tensor(list(bytes(uid)), dtype=torch.uint8)
            if device is not None:
                t = t.to(device)
            dist.broadcast(t, src=0)
            uid = (C.c_uint8 * 128)(*bytes(t.cpu().tolist()))
        h = C.c_void_p()
        # RCCL prints its version banner on C stdout at communicator creation; bench.py's stdout carries exactly one JSON
        # line, so fd 1 points at stderr while the communicator is built (and the C stdio buffer is flushed there)
        import os
        import sys
        sys.stdout.flush()
        libc = C.CDLL(None)
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            st = lib.jolt_comm_create(ctx.h, self._path, uid, C.c_int32(rank), C.c_int32(world), C.byref(h))
        finally:
            libc.fflush(None)
            os.dup2(saved, 1)
            os.close(saved)
        if st:
            ffi._ck(st, "jolt_comm_create", ctx)
        self.h = h

    def all_gather_u64(self, arr):
        flat = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1)
        out = np.empty(self.world * flat.size, dtype=np.uint64)
        st = ffi.lib().jolt_comm_all_gather_host(self.h, _p(flat), C.c_size_t(flat.nbytes), _p(out))
        if st:
            ffi._ck(st, "jolt_comm_all_gather_host", self.ctx)
        return out.reshape(self.world, -1)

    def all_gather_table(self, local, n, gathered):
        st = ffi.lib().jolt_comm_all_gather_table(self.h, local.h, C.c_size_t(n), gathered.h)
        if st:
            ffi._ck(st, "jolt_comm_all_gather_table", self.ctx)

    def close(self):
        if self.h:
            ffi.lib().jolt_comm_destroy(self.h)
            self.h = None


def tail_log_for(n_local, world, max_tail_entries=8192):
    """Local table size (log2) at which a shard hands over to the redundant tail: the largest k <= n_local with
    world * 2^k <= max_tail_entries (the tail kernel's single-launch range), never below 0."""
    log_g = world.bit_length() - 1
    k = max_tail_entries.bit_length() - 1 - log_g
    return max(0, min(n_local, k))


def msm_sharded(coll, local_point):
    """Term-range sharded G1 MSM: every rank runs the full bucket method over its own slice of
    the scalars against its resident slice of the bases (jolt_msm_g1_table) and passes the partial sum here; ONE
    all-gather of `world` Jacobian points (96 bytes each) and world - 1 point additions give the total on every rank.
    Same POINT as the single-process MSM (the Jacobian representative is free)."""
    pts = coll.all_gather_u64(np.ascontiguousarray(local_point, dtype=np.uint64).reshape(-1)).reshape(-1, 12)
    acc = pts[0].copy()
    for r in range(1, pts.shape[0]):
        acc = ffi.host_g1_add(acc, pts[r])
    return acc


# wall-clock split of prove_batch_sharded, accumulated over calls (seconds); bench.py reports it per step
TIMINGS = {"sharded_rounds": 0.0, "hand_over": 0.0, "tail_rounds": 0.0, "setup_and_end": 0.0}


class MemberInfo:
    def __init__(self, kind, degree, rounds, n_tables, w=None, scale=None):
        self.kind, self.degree, self.rounds, self.n_tables, self.w, self.scale = kind, degree, rounds, n_tables, w, scale

    @property
    def n_evals(self):
        if self.kind == KIND_SPLIT_EQ_UNIFORM:
            return self.degree - 1
        return 2 if self.kind == KIND_SPLIT_EQ else (self.degree if self.kind == KIND_EXPR_SKIP else self.degree + 1)


def prove_batch_sharded(ctx_handle, infos, claims, coeffs, n_total, n_local, max_degree, world, coll, shard, label=0,
                        challenge_mode=0, tail_log=0, force_gather=False, round_exchange=None):
    """infos: [MemberInfo] (global rounds = n_total for every member here); shard: local backend with
         round(active_idx, binds) -> np (total,4);  flush(binds);
         make_tail(coll, split_eq_scalars) -> tail backend over the gathered world * 2^tail_log-entry tables
       A backend exposing `handles` (device members, C array indexed like infos) is driven natively by the C++ round loop;
       with a NativeCollective the per-round all-gather is native too, so no Python runs inside the rounds.
       Returns dict(polys, challenges, member_claims, final_claim)."""
    import time
    t_begin = time.perf_counter()
    lib = ffi.lib()
    n = len(infos)
    ic = np.ascontiguousarray(np.stack(claims), dtype=np.uint64)
    co = np.ascontiguousarray(np.stack(coeffs), dtype=np.uint64)
    rounds = (C.c_size_t * n)(*[i.rounds for i in infos])
    offsets = (C.c_size_t * n)(*([0] * n))
    kinds = (C.c_int32 * n)(*[i.kind for i in infos])
    degs = (C.c_uint32 * n)(*[i.degree for i in infos])
    w_store = [np.ascontiguousarray(i.w, dtype=np.uint64) if i.w is not None else None for i in infos]
    w_ptrs = (C.c_void_p * n)(*[None if w is None else w.ctypes.data for w in w_store])
    one = ffi.host_fr_from_u64(1)
    scales = np.ascontiguousarray(np.stack([one if i.scale is None else np.asarray(i.scale, dtype=np.uint64) for i in infos]))
    batch = C.c_void_p()
    st = lib.jolt_host_batch_begin(ctx_handle, C.c_size_t(n), _p(ic), _p(co), rounds, offsets, kinds, degs, w_ptrs, _p(scales),
                                   C.c_size_t(n_total), C.c_size_t(max_degree), C.c_uint64(label), C.c_int32(challenge_mode), C.byref(batch))
    if st:
        raise ffi.JoltError(st, "jolt_host_batch_begin")
    state = {"backend": shard, "err": None}

    def local_cb(user, active, n_active, binds, evals_out, count):
        try:
            idx = [active[k] for k in range(n_active)]
            bs = []
            for k in range(n_active):
                if binds[k]:
                    bs.append(np.ctypeslib.as_array(C.cast(binds[k], C.POINTER(C.c_uint64)), shape=(4,)).copy())
                else:
                    bs.append(None)
            ev = np.ascontiguousarray(state["backend"].round(idx, bs), dtype=np.uint64).reshape(-1, 4)
            assert ev.shape[0] == count, (ev.shape, count)
            C.memmove(evals_out, ev.ctypes.data, count * 32)
            return 0
        except Exception as e:  # never unwind through the C frame
            state["err"] = e
            return 4

    def gather_cb(user, local, count, gathered):
        try:
            loc = np.ctypeslib.as_array(C.cast(local, C.POINTER(C.c_uint64)), shape=(count * 4,)).copy()
            g = np.ascontiguousarray(coll.all_gather_u64(loc))
            C.memmove(gathered, g.ctypes.data, g.nbytes)
            return 0
        except Exception as e:
            state["err"] = e
            return 4

    lcb, gcb = LOCAL_FN(local_cb), GATHER_FN(gather_cb)
    native_gather = C.cast(lib.jolt_comm_gather_round_sums, GATHER_FN) if isinstance(coll, NativeCollective) else None
    native_user = coll.h if native_gather is not None else None
    if round_exchange is not None:  # the round sums go through shared memory; `coll` keeps the table hand-over
        native_gather, native_user = C.cast(lib.jolt_shm_gather_round_sums, GATHER_FN), round_exchange.h

    def run(n_rounds, w, exchange=True):
        if n_rounds == 0:
            return
        handles = getattr(state["backend"], "handles", None)
        gather, user = (None, None)
        if w > 1 or (force_gather and exchange):  # force_gather: run the exchange even with a single rank
            gather, user = (native_gather, native_user) if native_gather is not None else (gcb, None)
        st = lib.jolt_host_batch_run(batch, handles, C.c_size_t(n_rounds), C.c_int32(w), gather, None if handles is not None else lcb, user)
        if st:
            if state["err"] is not None:
                raise state["err"]
            raise ffi.JoltError(st, "jolt_host_batch_run")

    def flush():
        handles = getattr(state["backend"], "handles", None)
        if handles is not None:  # the C++ loop applies the pending binds to the device members itself
            st = lib.jolt_host_batch_flush_binds(batch, handles, None, None)
            if st:
                raise ffi.JoltError(st, "jolt_host_batch_flush_binds")
            return
        binds = ffi.fr_array(n)
        has = (C.c_int32 * n)()
        st = lib.jolt_host_batch_flush_binds(batch, None, _p(binds), has)
        if st:
            raise ffi.JoltError(st, "jolt_host_batch_flush_binds")
        state["backend"].flush([binds[i] if has[i] else None for i in range(n)])

    log_g = n_total - n_local
    assert 0 <= tail_log <= n_local
    t0 = time.perf_counter()
    run(n_local - tail_log, world)
    flush()
    t1 = time.perf_counter()
    t2 = t1
    if log_g + tail_log > 0:
        scalars = []
        for i, info in enumerate(infos):
            if info.kind in (KIND_SPLIT_EQ, KIND_SPLIT_EQ_UNIFORM):
                o = ffi.fr_array(1)
                lib.jolt_host_batch_split_eq_scalar(batch, C.c_size_t(i), _p(o))
                scalars.append(o[0])
            else:
                scalars.append(None)
        tail = shard.make_tail(coll, scalars)
        t2 = time.perf_counter()
        state["backend"] = tail
        run(log_g + tail_log, 1, exchange=False)
        flush()
        tail.close()
    t3 = time.perf_counter()
    TIMINGS["sharded_rounds"] += t1 - t0
    TIMINGS["hand_over"] += t2 - t1
    TIMINGS["tail_rounds"] += t3 - t2
    polys, chal = ffi.fr_array(n_total * (max_degree + 1)), ffi.fr_array(n_total)
    mclaims, final = ffi.fr_array(n), ffi.fr_array(1)
    st = lib.jolt_host_batch_end(batch, _p(polys), _p(chal), _p(mclaims), _p(final))
    if st:
        raise ffi.JoltError(st, "jolt_host_batch_end")
    TIMINGS["setup_and_end"] += (time.perf_counter() - t3) + (t0 - t_begin)
    return dict(polys=polys.reshape(n_total, max_degree + 1, 4), challenges=chal, member_claims=mclaims, final_claim=final[0])


# ---------------------------------------------------------------------------------------------------------------------
# device backend
# ---------------------------------------------------------------------------------------------------------------------
class DeviceShard:
    """Local backend over device members: the C++ round loop drives them directly through `handles`
    (jolt_round_group_prove / jolt_round_group_finish); round()/flush() are the same operations for callers that step
    the loop from Python."""

    def __init__(self, ctx, members, rebuild=None):
        self.ctx, self.members, self.rebuild = ctx, members, rebuild
        self.handles = (C.c_void_p * len(members))(*[m.h for m in members])

    def round(self, idx, binds):
        res = self.ctx.round_group_prove([self.members[i] for i in idx], binds)
        return np.concatenate(res, axis=0)

    def flush(self, binds):
        ms = [m for m, b in zip(self.members, binds) if b is not None]
        bs = [b for b in binds if b is not None]
        if not ms:
            return
        hs = (C.c_void_p * len(ms))(*[m.h for m in ms])
        store = [ffi.fr(b) for b in bs]
        bp = (C.c_void_p * len(ms))(*[b.ctypes.data for b in store])
        st = ffi.lib().jolt_round_group_finish(self.ctx.h, hs, C.c_size_t(len(ms)), bp)
        if st:
            raise ffi.JoltError(st, "jolt_round_group_finish")

    def make_tail(self, coll, scalars):
        return self.rebuild(coll, scalars)

    def close(self):
        pass


# ---------------------------------------------------------------------------------------------------------------------
# a batch over ANY device members (the stage operators of jolt_amd/stages_sharded.py): phase A on the rank's blocks, one hand-over, the redundant tail
# ---------------------------------------------------------------------------------------------------------------------
_P_LIMBS = np.array([(0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001 >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)


def fr_add_vec(a, b):
    """(n, 4) + (n, 4) canonical field elements (Montgomery limbs add like plain residues), vectorised: what a rank does with the other ranks' partial sums"""
    a, b = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4), np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    out = np.empty_like(a)
    carry = np.zeros(a.shape[0], dtype=np.uint64)
    for i in range(4):
        s = a[:, i] + b[:, i]
        c1 = s < a[:, i]
        s2 = s + carry
        c2 = s2 < s
        out[:, i] = s2
        carry = (c1 | c2).astype(np.uint64)
    d = np.empty_like(out)  # out < 2 r < 2^255: one conditional subtraction
    borrow = np.zeros(a.shape[0], dtype=np.uint64)
    for i in range(4):
        t = out[:, i] - _P_LIMBS[i]
        b1 = out[:, i] < _P_LIMBS[i]
        t2 = t - borrow
        b2 = t < borrow
        d[:, i] = t2
        borrow = (b1 | b2).astype(np.uint64)
    take = borrow == 0
    out[take] = d[take]
    return out


def gather_sum(coll, values):
    """sum over the ranks of an (n, 4) array of partial field sums: ONE all-gather + a local modular sum (RCCL has no mod-r reduction)"""
    v = np.ascontiguousarray(values, dtype=np.uint64).reshape(-1, 4)
    allv = np.ascontiguousarray(coll.all_gather_u64(v.reshape(-1)))
    allv = allv.reshape(allv.shape[0], -1, 4)
    acc = allv[0].copy()
    for r in range(1, allv.shape[0]):
        acc = fr_add_vec(acc, allv[r])
    return acc.reshape(np.asarray(values).shape)


class _TailShard(DeviceShard):
    """DeviceShard whose hand-over builds throw-away tail members through each member's `_tail(tables, scalar, w_rem)` closure and keeps them (their final
    values are the batch's); buffers and members are dropped by drop()."""

    def __init__(self, ctx, members, infos, world, tail_log):
        super().__init__(ctx, members)
        self.infos, self.world, self.tail_log = infos, world, tail_log
        self.tail_members, self.keep = None, []

    def make_tail(self, coll, scalars):
        ctx, world, lib = self.ctx, self.world, ffi.lib()
        log_g = world.bit_length() - 1
        E = 1 << self.tail_log
        n_tab = sum(m.n_tables for m in self.members)
        rem = log_g + self.tail_log
        pack, gath, arena = ctx.alloc(n_tab * E), ctx.alloc(world * n_tab * E), ctx.alloc(world * n_tab * E)
        self.keep += [pack, gath, arena]
        hs = (C.c_void_p * len(self.members))(*[m.h for m in self.members])
        ffi._ck(lib.jolt_round_group_pack_tables(ctx.h, hs, C.c_size_t(len(self.members)), C.c_size_t(E), pack.h), "jolt_round_group_pack_tables", ctx)
        if isinstance(coll, NativeCollective):
            coll.all_gather_table(pack, n_tab * E, gath)
        else:  # torch.distributed: through the host
            g = coll.all_gather_u64(pack.download())
            ffi._ck(lib.jolt_table_write(ctx.h, gath.h, C.c_size_t(0), _p(np.ascontiguousarray(g.reshape(-1, 4))), C.c_size_t(world * n_tab * E)), "jolt_table_write", ctx)
        ffi._ck(lib.jolt_tail_interleave(ctx.h, gath.h, C.c_size_t(world), C.c_size_t(n_tab), C.c_size_t(E), arena.h), "jolt_tail_interleave", ctx)
        tails, pos = [], 0
        for k, m in enumerate(self.members):
            tabs = []
            for _ in range(m.n_tables):
                h = C.c_void_p()
                ffi._ck(lib.jolt_table_slice(ctx.h, arena.h, C.c_size_t(pos * world * E), C.c_size_t(world * E), C.byref(h)), "jolt_table_slice", ctx)
                tabs.append(ffi.Table(ctx, h))
                pos += 1
            self.keep += tabs
            w = self.infos[k].w
            tails.append(m._tail(tabs, scalars[k], None if w is None else np.asarray(w)[:rem]))
        self.tail_members = tails
        return DeviceShard(ctx, tails)

    def drop(self):
        for m in self.tail_members or []:
            m.destroy()
        for t in self.keep:
            t.free()
        self.tail_members, self.keep = None, []


def prove_members_sharded(ctx, coll, world, members, infos, claims, coeffs, n_total, n_local, max_degree, label=0, tail_log=None, round_exchange=None, force_gather=False):
    """One batched sumcheck over device members that hold this rank's block of their tables (every member: n_total rounds, a `_tail` closure for the hand-over).
    -> (the transcript of prove_batch_sharded, per member its final values after ALL n_total rounds: tables..., and the eq scalar of a split-eq member)."""
    if tail_log is None:
        tail_log = tail_log_for(n_local, world)
    tail_log = min(tail_log, n_local)
    shard = _TailShard(ctx, members, infos, world, tail_log)
    try:
        out = prove_batch_sharded(ctx.h, infos, claims, coeffs, n_total, n_local, max_degree, world, coll, shard, label=label, tail_log=tail_log, force_gather=force_gather,
                                  round_exchange=round_exchange)
        finals = [m.final_values() for m in (shard.tail_members if shard.tail_members is not None else members)]
    finally:
        shard.drop()
    return out, finals


def build_sharded_spec(n_local, rank, world, seed=2026):
    """Pure description of rank `rank`'s shard of the bench workload: witness columns are seeded per rank; every derived
    leaf (eq / eq+1 / LT in the single-GPU workload) is the rank's aligned block of eq(point, .) for a GLOBAL random point
    of n_local + log2(world) coordinates; gammas, split-eq points and batching coefficients are global (same on all ranks)."""
    log_g = world.bit_length() - 1
    n_total = n_local + log_g
    tables_spec, members_spec, gammas = W.build(n_local, seed + 7919 * rank)
    g_rng = np.random.default_rng(seed)
    gam = [W.rand_fr(1, g_rng)[0] for _ in gammas]
    tables = {}
    for name, spec in tables_spec.items():
        if spec.kind in ("u64", "i64"):
            tables[name] = {"kind": spec.kind, "data": spec.data}
        elif spec.kind == "onehot":  # hot indices are witness data (seeded per rank); the chunk point is global
            tables[name] = {"kind": "onehot", "data": spec.data, "point": W.rand_fr(len(spec.point), g_rng)}
        else:
            tables[name] = {"kind": "eqblock", "point": W.rand_fr(n_total, g_rng)}
    split_points = {k: W.rand_fr(n_total, g_rng) for k, ms in enumerate(members_spec) if ms.split_eq is not None}
    batch_coeffs = [W.rand_fr(1, g_rng)[0] for _ in members_spec]
    return {"tables": tables, "members": members_spec, "gammas": gam, "split_points": split_points, "batch_coeffs": batch_coeffs,
            "n_total": n_total}


class ShardedWorkload:
    """The bench workload sharded over `world` GPUs (weak scaling: every rank holds T = 2^n_local cycles of a trace of
    world * T cycles).  Tables are synthetic per rank: witness columns are seeded per rank, eq tables are the aligned block
    of a global point (EqPolynomial::evals_for_aligned_block); LT / eq+1 leaves are replaced by eq blocks of the same
    size (the arithmetic per entry is identical, their global structure is exercised by the single-GPU tests)."""

    def __init__(self, ctx, n_local, rank, world, dist, seed=2026, coll=None, tail_log=None, force_gather=False):
        self.ctx, self.n_local, self.rank, self.world = ctx, n_local, rank, world
        log_g = world.bit_length() - 1
        assert (1 << log_g) == world, "world size must be a power of two"
        self.n_total = n_local + log_g
        if tail_log is None and os.environ.get("JOLT_TAIL_LOG"):  # measurement knob
            tail_log = min(n_local, int(os.environ["JOLT_TAIL_LOG"]))
        self.tail_log = tail_log_for(n_local, world) if tail_log is None else tail_log
        self.force_gather = force_gather
        if coll is None:
            import torch
            dev = torch.device("cuda", torch.cuda.current_device())
            # RCCL called natively from the C++ round loop; torch.distributed only carries the rendezvous.  All ranks must
            # agree on the choice: if the native communicator fails anywhere, everybody falls back to torch.distributed.
            # With more than one rank the native communicator is OPT-IN (JOLT_NATIVE_RCCL=1): it has only ever run with one rank (the
            # development box has one GPU and RCCL refuses two ranks per device), and a process that imports torch already holds torch's
            # own RCCL.  The default for N > 1 is therefore torch.distributed's nccl backend -- which IS RCCL over xGMI -- for the table
            # hand-over and the partial-point gathers; the per-round sums go through shared memory either way.
            native, err = None, None
            want_native = world == 1 or os.environ.get("JOLT_NATIVE_RCCL") == "1"
            try:
                if want_native:
                    native = NativeCollective(ctx, dist, rank, world, dev)
                else:
                    err = "not requested (JOLT_NATIVE_RCCL=1 enables it)"
            except Exception as e:  # noqa: BLE001 -- reported below, decision taken collectively
                err = e
            ok = torch.tensor([1 if native is not None else 0], dtype=torch.int32, device=dev)
            if world > 1:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1 and world > 1:  # probe: the native all-gather must agree with torch.distributed's
                probe = np.arange(8, dtype=np.uint64) + np.uint64(1000 * rank)
                same = np.array_equal(native.all_gather_u64(probe), Collective(dist, world, dev).all_gather_u64(probe))
                ok = torch.tensor([1 if same else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if not same:
                    err = "probe all-gather mismatch"
            self.communicator_note = "native RCCL communicator (jolt_comm_*: the librccl the process already holds -- torch's when torch is imported -- else the system one; %d rank(s))" % world
            if int(ok.item()) == 1:
                coll = native
            else:
                self.communicator_note = (f"torch.distributed nccl backend (= RCCL), {world} ranks" if not want_native
                                          else f"torch.distributed fallback: native RCCL communicator unavailable ({err})")
                if native is not None:
                    native.close()
                if want_native:
                    import sys
                    print(f"[jolt_amd] rank {rank}: native RCCL communicator unavailable ({err}); using torch.distributed", file=sys.stderr)
                coll = Collective(dist, world, dev)
        if not hasattr(self, "communicator_note"):
            self.communicator_note = f"caller-supplied {type(coll).__name__}"
        self.coll = coll
        # round sums: shared memory between the ranks of the node when every rank can map it, else the collective above
        self.round_exchange = make_shm_exchange(dist, rank, world) if (world > 1 or force_gather) else None
        spec = build_sharded_spec(n_local, rank, world, seed)
        # the raw committed columns of this rank's block (what ShardedPcs commits to): hot indices per lazy member, the two increments
        self.committed_onehot = [np.stack([spec["tables"][t]["data"] for t in ms.tables[1:]]) for ms in spec["members"]
                                 if ms.uniform is not None and all(spec["tables"][t]["kind"] == "onehot" for t in ms.tables[1:])]
        self.committed_dense = [spec["tables"]["s6.ram_inc"]["data"], spec["tables"]["s6.rd_inc"]["data"]]
        self.members_spec = spec["members"]
        one = ffi.host_fr_from_u64(1)
        zero = np.zeros(4, dtype=np.uint64)
        self.resolver = W.Resolver(spec["gammas"], one, ffi.host_fr_mul, lambda x: ffi.host_fr_sub(zero, x))
        self.spec = spec
        skip = {ms.tables[0] for ms in self.members_spec if ms.uniform is not None or ms.eq_inner is not None}
        skip -= {t for ms in self.members_spec for t in (ms.tables if (ms.uniform is None and ms.eq_inner is None) else ms.tables[1:])}
        # one-hot selector columns of uniform members stay index-encoded on every rank (lazily bound members, as on one GPU)
        # (only when the fourth bind -- the one that writes them dense -- comes before the hand-over to the tail)
        lazy = lambda ms: (ms.uniform is not None and n_local - self.tail_log >= 4
                           and all(spec["tables"][t]["kind"] == "onehot" for t in ms.tables[1:]))
        skip |= {t for ms in self.members_spec if lazy(ms) for t in ms.tables[1:]}
        self._skip, self._lazy_ms = skip, lazy
        # ---- resident inputs (the witness of this rank's block): integer columns, hot indices of the lazy members
        self.ints, self.sources, self.scale_tables = {}, {}, {}
        for name, t in spec["tables"].items():
            if name not in skip and t["kind"] in ("u64", "i64"):
                self.ints[name] = ctx.ints(t["data"].astype(np.uint64 if t["kind"] == "u64" else np.int64))
        for k, ms in enumerate(self.members_spec):
            if lazy(ms):
                specs = [spec["tables"][t] for t in ms.tables[1:]]
                self.sources[k] = ctx.onehot(np.stack([sp["data"] for sp in specs]), 1 << len(specs[0]["point"]))
        self.batch_coeffs = spec["batch_coeffs"]
        self.n_tables = sum(len(ms.tables) for ms in self.members_spec)
        self.tables, self.members, self.infos, self.stages, self.prepared = {}, [], [], {}, False
        self._tails = {}
        self.prepare()
        # global input claims = sum over ranks of the local claims (one all-gather at setup; in the real prover they are
        # the previous stage's output claims)
        local_claims = np.stack([self._local_claim(i) for i in range(len(self.members))])
        allc = self.coll.all_gather_u64(local_claims).reshape(world, len(self.members), 4)
        zero = np.zeros(4, dtype=np.uint64)
        self.claims = []
        for i in range(len(self.members)):
            acc = zero
            for r in range(world):
                acc = ffi.host_fr_add(acc, allc[r, i])
            self.claims.append(acc)

    def release(self):
        for m in self.members:
            m.destroy()
        for t in self.tables.values():
            t.free()
        self.tables, self.members, self.infos, self.stages, self.prepared = {}, [], [], {}, False

    def prepare(self):
        """Everything a proof builds before its first round, on every rank (the N = 1 step's `prepare`, DeviceWorkload.prepare): the rank's
        block of every derived table, the promoted witness columns, the linear-leaf fusions and the members over them."""
        if self.prepared:
            self.release()
        ctx, spec, rank, world, n_local = self.ctx, self.spec, self.rank, self.world, self.n_local
        log_g = world.bit_length() - 1
        one = ffi.host_fr_from_u64(1)
        skip, lazy = self._skip, self._lazy_ms
        for name, t in spec["tables"].items():
            if name in skip:
                continue
            if t["kind"] == "onehot":
                src = ctx.onehot(t["data"].reshape(1, -1), 1 << len(t["point"]))
                st = ctx.eq_evals(t["point"])
                self.tables[name] = src.materialize(0, st)
                ctx.synchronize()
                st.free()
                src.free()
                continue
            if t["kind"] in ("u64", "i64"):
                self.tables[name] = ctx.table_from_ints(self.ints[name])
            else:  # the aligned block of eq(point, .) owned by this rank (EqPolynomial::evals_for_aligned_block)
                self.tables[name] = ctx.eq_evals_aligned_block(t["point"], rank << n_local, 1 << n_local)
        def shard_scale_of(w):
            sc = one
            for j in range(log_g):  # eq(w_hi, rank), big-endian
                bit = (rank >> (log_g - 1 - j)) & 1
                sc = ffi.host_fr_mul(sc, w[j] if bit else ffi.host_fr_sub(one, w[j]))
            return sc

        for k, ms in enumerate(self.members_spec):
            tabs = [self.tables.get(t) for t in ms.tables]
            if ms.uniform is not None:
                V, F, csyms = ms.uniform
                w = spec["tables"][ms.tables[0]]["point"]  # the eq leaf's global point
                coeffs = [self.resolver.coeff(c) for c in csyms]
                if lazy(ms):
                    specs = [spec["tables"][t] for t in ms.tables[1:]]
                    src = self.sources[k]
                    scale_tables = np.stack([ffi.host_eq_evals(sp["point"]) for sp in specs])  # eq(r_chunk, .) over K = 16 entries, on the host
                    m = ctx.member_lazy_ra_uniform(src, scale_tables, V, F, coeffs, w[log_g:], shard_scale=shard_scale_of(w))
                    # the tables this member hands over to the tail carry c_v folded into the first factor of product v
                    m._uniform = (V, F, [one] * V)
                    m._lazy = (src, scale_tables, coeffs)
                else:
                    m = ctx.member_split_eq_uniform(tabs[1:], V, F, coeffs, w[log_g:], shard_scale=shard_scale_of(w), borrow=True)
                    m._uniform = (V, F, coeffs)
                self.infos.append(MemberInfo(KIND_SPLIT_EQ_UNIFORM, F + 1, self.n_total, V * F, w=w))
                self.members.append(m)
                self.stages.setdefault(ms.stage, []).append(len(self.members) - 1)
                continue
            if ms.eq_inner is not None:  # eq(w, j) * q(j), the eq weight factored out on every rank (shard scale = eq(w_hi, rank))
                dq, inner = ms.eq_inner
                w = spec["tables"][ms.tables[0]]["point"]
                groups = self.resolver.groups(inner)
                m = ctx.member_lc(tabs[1:], groups, dq, borrow=True, eq_point=w[log_g:], shard_scale=shard_scale_of(w))
                m._groups, m._dq = groups, dq
                self.infos.append(MemberInfo(KIND_SPLIT_EQ_UNIFORM, dq + 1, self.n_total, len(tabs) - 1, w=w))
                self.members.append(m)
                self.stages.setdefault(ms.stage, []).append(len(self.members) - 1)
                continue
            if ms.fused is not None:  # linear-leaf fusion on every rank: the fused leaf is the same combination of the rank's blocks
                parts, names, fgroups = ms.fused
                for fname, entries in parts:
                    srcs = [self.tables[ms.tables[ti]] for _, ti in entries]
                    self.tables[fname] = ctx.rlc(srcs, np.stack([self.resolver.coeff(c) for c, _ in entries]))
                ctx.synchronize()
                groups = self.resolver.groups(fgroups)
                ftabs = [self.tables[t] for t in names]
                m = ctx.member_lc(ftabs, groups, ms.degree, borrow=True, skip_one=True)
                m._groups = groups
                self.infos.append(MemberInfo(KIND_EXPR_SKIP, ms.degree, self.n_total, len(ftabs)))
                self.members.append(m)
                self.stages.setdefault(ms.stage, []).append(len(self.members) - 1)
                continue
            if ms.split_eq is not None:
                a, b, _ = ms.split_eq
                w = spec["split_points"][k]
                shard_scale = one
                for j in range(log_g):  # eq(w_hi, rank), big-endian
                    bit = (rank >> (log_g - 1 - j)) & 1
                    shard_scale = ffi.host_fr_mul(shard_scale, w[j] if bit else ffi.host_fr_sub(one, w[j]))
                h = C.c_void_p()
                wl = np.ascontiguousarray(w[log_g:])
                st = ffi.lib().jolt_member_create_split_eq_product_sharded(ctx.h, tabs[a].h, tabs[b].h, _p(wl), C.c_size_t(n_local), None,
                                                                        _p(ffi.fr(shard_scale)), C.byref(h))
                if st:
                    raise ffi.JoltError(st, "jolt_member_create_split_eq_product_sharded")
                m = ffi.Member(ctx, h, 3, 2, True, False)
                m._keepalive = [tabs[a], tabs[b]]
                self.infos.append(MemberInfo(KIND_SPLIT_EQ, 3, self.n_total, 2, w=w))
            else:
                groups = self.resolver.groups(ms.groups)
                m = ctx.member_lc(tabs, groups, ms.degree, borrow=True, skip_one=True)
                m._groups = groups
                self.infos.append(MemberInfo(KIND_EXPR_SKIP, ms.degree, self.n_total, len(tabs)))
            self.members.append(m)
            self.stages.setdefault(ms.stage, []).append(len(self.members) - 1)
        self.prepared = True

    def _local_claim(self, i):
        """This rank's share of member i's input claim.  Split-eq members are summed against the rank's aligned block of the
        dense eq table (the members' own claim helper does not know the shard scale)."""
        m, ms = self.members[i], self.members_spec[i]
        if not m.split_eq:
            return m.input_claim()
        one = ffi.host_fr_from_u64(1)
        eq = self.ctx.eq_evals_aligned_block(self.infos[i].w, self.rank << self.n_local, 1 << self.n_local)
        temp = []
        if ms.eq_inner is not None:  # eq block as one more factor of every inner group
            dq, _ = ms.eq_inner
            cols = [self.tables[t] for t in ms.tables[1:]]
            groups = [[(None, [(one, 0)])] + [(c, [(cf, 1 + ti) for cf, ti in ents]) for c, ents in g] for g in m._groups]
            tmp = self.ctx.member_lc([eq] + cols, groups, dq + 1, borrow=True)
        elif ms.uniform is not None:
            V, F, coeffs = m._uniform
            if getattr(m, "_lazy", None) is not None:  # index-encoded columns: gathered dense for this setup-time helper only
                src, scale_tables, coeffs = m._lazy
                for p in range(V * F):
                    st = self.ctx.upload(scale_tables[p])
                    temp.append(src.materialize(p, st))
                    self.ctx.synchronize()
                    st.free()
                cols = temp
            else:
                cols = [self.tables[t] for t in ms.tables[1:]]
            tabs = [eq] + cols
            groups = [[(None, [(coeffs[v], 0)])] + [(None, [(one, 1 + v * F + k)]) for k in range(F)] for v in range(V)]
            tmp = self.ctx.member_lc(tabs, groups, F + 1, borrow=True)
        else:
            a, b, _ = ms.split_eq
            tabs = [eq, self.tables[ms.tables[a]], self.tables[ms.tables[b]]]
            tmp = self.ctx.member_lc(tabs, [[(None, [(one, 0)]), (None, [(one, 1)]), (None, [(one, 2)])]], 3, borrow=True)
        c = tmp.input_claim()
        tmp.destroy()
        eq.free()
        for t in temp:
            t.free()
        return c

    def _tail_backend(self, stage, idxs, coll, scalars):
        """Hand-over to the redundant tail (phase B): pack this rank's 2^tail_log-entry tables, all-gather them, rebuild the
        world * 2^tail_log-entry tables with the rank as the top variables.  Buffers, table views and tail members are created
        once per stage and only refilled afterwards (split-eq members are rebuilt: their scalar depends on the challenges)."""
        ctx, world, lib = self.ctx, self.world, ffi.lib()
        log_g = world.bit_length() - 1
        E = 1 << self.tail_log
        n_tab = sum(self.members[i].n_tables for i in idxs)
        rem = log_g + self.tail_log  # variables left: the top coordinates of every global point
        if stage not in self._tails:
            self._tails[stage] = {"pack": ctx.alloc(n_tab * E), "gath": ctx.alloc(world * n_tab * E), "arena": ctx.alloc(world * n_tab * E),
                                  "slices": None, "members": None}
        T = self._tails[stage]
        hs = (C.c_void_p * len(idxs))(*[self.members[i].h for i in idxs])
        st = lib.jolt_round_group_pack_tables(ctx.h, hs, C.c_size_t(len(idxs)), C.c_size_t(E), T["pack"].h)
        if st:
            ffi._ck(st, "jolt_round_group_pack_tables", ctx)
        if isinstance(coll, NativeCollective):
            coll.all_gather_table(T["pack"], n_tab * E, T["gath"])
        else:  # torch.distributed fallback: through the host
            g = coll.all_gather_u64(T["pack"].download())
            st = lib.jolt_table_write(ctx.h, T["gath"].h, C.c_size_t(0), _p(np.ascontiguousarray(g.reshape(-1, 4))), C.c_size_t(world * n_tab * E))
            if st:
                ffi._ck(st, "jolt_table_write", ctx)
        st = lib.jolt_tail_interleave(ctx.h, T["gath"].h, C.c_size_t(world), C.c_size_t(n_tab), C.c_size_t(E), T["arena"].h)
        if st:
            ffi._ck(st, "jolt_tail_interleave", ctx)

        def build_split(i, tabs, scalar):
            m = self.members[i]
            if getattr(m, "eq_weighted", False):
                tm = ctx.member_lc(tabs, m._groups, m._dq, borrow=True, eq_point=self.infos[i].w[:rem], eq_scale=scalar)
                return tm
            if getattr(m, "uniform", False):
                V, F, coeffs = m._uniform
                return ctx.member_split_eq_uniform(tabs, V, F, coeffs, self.infos[i].w[:rem], scale=scalar, borrow=True)
            return ctx.member_split_eq_product(tabs[0], tabs[1], self.infos[i].w[:rem], scale=scalar, borrow=True)

        if T["members"] is None:
            slices, members, pos = [], [], 0
            for k, i in enumerate(idxs):
                m = self.members[i]
                tabs = []
                for _ in range(m.n_tables):
                    h = C.c_void_p()
                    st = lib.jolt_table_slice(ctx.h, T["arena"].h, C.c_size_t(pos * world * E), C.c_size_t(world * E), C.byref(h))
                    if st:
                        ffi._ck(st, "jolt_table_slice", ctx)
                    tabs.append(ffi.Table(ctx, h))
                    pos += 1
                slices.append(tabs)
                if m.split_eq:
                    members.append(build_split(i, tabs, scalars[k]))
                else:
                    members.append(ctx.member_lc(tabs, m._groups, m.degree, borrow=True, skip_one=True))
            T["slices"], T["members"] = slices, members
        else:
            for k, i in enumerate(idxs):
                T["members"][k].reset()
                if self.members[i].split_eq:  # same point, new eq scalar (the product of this proof's shard-local challenges)
                    T["members"][k].set_scale(scalars[k])
        return DeviceShard(ctx, T["members"])

    def prove(self, label=0):
        outs = {}
        for stage, idxs in sorted(self.stages.items()):
            ms = [self.members[i] for i in idxs]
            infos = [self.infos[i] for i in idxs]
            deg = max(m.degree for m in ms)
            shard = DeviceShard(self.ctx, ms, rebuild=lambda c, s, stage=stage, idxs=idxs: self._tail_backend(stage, idxs, c, s))
            outs[stage] = prove_batch_sharded(self.ctx.h, infos, [self.claims[i] for i in idxs], [self.batch_coeffs[i] for i in idxs],
                                              self.n_total, self.n_local, deg, self.world, self.coll, shard, label=label + stage,
                                              tail_log=self.tail_log, force_gather=self.force_gather, round_exchange=self.round_exchange)
        for m in self.members:
            m.reset()
        return outs


# ---------------------------------------------------------------------------------------------------------------------------------
# PCS legs of the sharded step (BASELINE configs[2] at N > 1): commitments and ONE HyperKZG opening over the global commitment grid
# ---------------------------------------------------------------------------------------------------------------------------------
class ShardedPcs:
    """The committed columns of a trace of world * T cycles over the shared 2^(log_k) x (world * T) commitment grid, with the PCS work
    sharded over the ranks (north_star: "the MSM scalar set shards naturally across the GPUs, bucket sums reduced with RCCL"; DESIGN.md
    section 6):

      resident on every rank (inputs, before the timed region): the raw committed columns of the WHOLE trace -- 36 hot-index bytes
        and two 64-bit increments per cycle (52 B per cycle, 1.7 GB at 8 x 2^22 cycles) -- and the rank's compact SRS objects with
        their window tables;
      commit(): rank g multiplies / sums the bases of ITS block of cycles for every column; one all-gather of 38 partial points;
      open():   default (subtree): every rank builds ITS 1 / world of the joint polynomial and runs folds, Horner passes, RLC, quotient
        scans and MSMs on it (jolt_host_hyperkzg_open_subtree); five small exchanges per opening.  subtree=False: the polynomial
        arithmetic replicated, every MSM split over the ranks (block-cyclically, or by term range with block_cyclic=False).
    Every rank ends up with the same commitments and the same proof as a single process over the global trace (tests/
    test_gpu_distributed.py).  `gather(points)` all-gathers (count, 12) uint64 arrays; `gather_fn / gather_user` are the C-level
    jolt_gather_fn the opening calls."""

    def __init__(self, ctx, rank, world, n_local, onehot_global, dense_global, gather_points, gather_fn, gather_user, seed=2026, log_k=4, fixed_base=False,
                 block_cyclic=True, subtree=None):
        from .workload import G1_GENERATOR, rand_fr
        self.ctx, self.rank, self.world, self.n_local, self.log_k = ctx, rank, world, n_local, log_k
        self.T_local, self.T_global = 1 << n_local, world << n_local
        self.grid_vars = log_k + n_local + (world.bit_length() - 1)
        self.gather_points, self.gather_fn, self.gather_user = gather_points, gather_fn, gather_user
        self.sources = [ctx.onehot(idx, 1 << log_k) for idx in onehot_global]  # (polys, world * T) hot indices, cycle-major blocks by rank
        self.dense_ints = [ctx.ints(np.ascontiguousarray(d)) for d in dense_global]
        prng = np.random.default_rng(seed + 2)
        self.beta = rand_fr(1, prng)[0]
        # Block-cyclic term assignment: term i of every MSM belongs to rank (i / T_local) % world.  The grid
        # position of (address k, cycle g * T_local + j) is k * T_global + g * T_local + j, i.e. block k * world + g: a rank's
        # terms are the grid of ITS cycles, its compact SRS is that grid's bases in the single-GPU layout k * T_local + j, and the
        # window tables over it (one set of 2^(log_k + n_local) points whatever the world size) serve the commitments and every
        # level of the opening.  block_cyclic=False: contiguous term ranges against the full SRS (tables only for world <= 2).
        self.block = self.T_local if block_cyclic else 0
        if self.block:
            lo = rank * self.T_local
            self.local_sources = [ctx.onehot(np.ascontiguousarray(np.asarray(idx)[:, lo: lo + self.T_local]), 1 << log_k) for idx in onehot_global]
            self.srs = ctx.srs_setup_from_secret_blocks(self.beta, 1 << self.grid_vars, G1_GENERATOR, self.block, rank, world)
        else:
            self.srs = ctx.srs_setup_from_secret(self.beta, 1 << self.grid_vars, G1_GENERATOR)
        ctx.synchronize()
        if fixed_base and log_k + n_local >= 12:
            ctx.srs_precompute_windows(self.srs)
        # Subtree assignment for the OPENING (tests/subtree_model.py): the polynomial itself is sharded -- every rank
        # builds, folds, combines, divides and commits 1 / world of the joint polynomial against a second compact SRS (the bases of the
        # indices it owns under that assignment) with its own window tables.  The commitments keep the block-cyclic assignment above
        # (a rank commits its own cycles).  Default from 2 ranks on when the world size is a power of two.
        self.subtree = (block_cyclic and world >= 2 and world & (world - 1) == 0) if subtree is None else bool(subtree)
        self.srs_open = None
        if self.subtree:
            self.srs_open = ctx.srs_setup_from_secret_subtree(self.beta, 1 << self.grid_vars, G1_GENERATOR, rank, world)
            ctx.synchronize()
            if fixed_base and log_k + n_local >= 12:
                ctx.srs_precompute_windows(self.srs_open)
        prng = np.random.default_rng(seed + 3)
        self.rlc_onehot = rand_fr(sum(s.n_polys for s in self.sources), prng)
        self.rlc_dense = rand_fr(len(self.dense_ints), prng)
        self.open_point = rand_fr(self.grid_vars, prng)
        self.dense_tables = []

    def _promote_dense(self):
        """The global dense columns as field tables (commit's MSM scalars and a term of the joint polynomial); freed by open()."""
        if not self.dense_tables:
            self.dense_tables = [self.ctx.table_from_ints(d) for d in self.dense_ints]

    def commit(self):
        ctx, lo = self.ctx, self.rank * self.T_local
        self._promote_dense()
        parts = []
        base_lo = 0 if self.block else lo  # compact SRS: the rank's cycles of address row 0 are its first T_local bases
        for t in self.dense_tables:
            out = ffi.g1_array(1)
            ffi._ck(ffi.lib().jolt_msm_g1_table_range(ctx.h, self.srs.h, C.c_size_t(base_lo), t.h, C.c_size_t(lo), C.c_size_t(self.T_local), ffi._p(out)),
                    "jolt_msm_g1_table_range", ctx)
            parts.append(out)
        if self.block:
            for s in self.local_sources:  # the single-GPU commitment of the rank's own cycles over its compact SRS
                out = ffi.g1_array(s.n_polys)
                ffi._ck(ffi.lib().jolt_grid_commit_onehot(ctx.h, self.srs.h, s.h, ffi._p(out)), "jolt_grid_commit_onehot", ctx)
                parts.append(out)
        else:
            for s in self.sources:
                out = ffi.g1_array(s.n_polys)
                ffi._ck(ffi.lib().jolt_grid_commit_onehot_range(ctx.h, self.srs.h, s.h, C.c_size_t(lo), C.c_size_t(lo + self.T_local), ffi._p(out)),
                        "jolt_grid_commit_onehot_range", ctx)
                parts.append(out)
        local = np.concatenate(parts)
        allp = self.gather_points(local)  # (world, count, 12)
        total = allp[0].copy()
        for r in range(1, self.world):
            for i in range(total.shape[0]):
                total[i] = ffi.host_g1_add(total[i], allp[r][i])
        nd = len(self.dense_tables)
        return dict(dense=total[:nd], onehot=total[nd:])

    def open(self, label=0):
        ctx = self.ctx
        self._promote_dense()  # open() without a preceding commit(), or twice in a row: the columns are promoted again
        if self.subtree:
            joint = ctx.grid_joint_polynomial_subtree(self.sources, self.rlc_onehot, self.dense_tables, self.rlc_dense, self.log_k, self.rank, self.world)
            out = ctx.hyperkzg_open_subtree(self.srs_open, joint, self.open_point, label, self.rank, self.world, self.gather_fn, self.gather_user)
            joint.free()
            for t in self.dense_tables:
                t.free()
            self.dense_tables = []
            return out
        joint = ctx.grid_joint_polynomial(self.sources, self.rlc_onehot, self.dense_tables, self.rlc_dense, self.log_k)
        p = ffi.fr(self.open_point).reshape(-1, 4)
        ell = p.shape[0]
        com, w, v, ch = ffi.g1_array(max(ell - 1, 1)), ffi.g1_array(3), ffi.fr_array(3 * ell), ffi.fr_array(3)
        if self.block:
            ffi._ck(ffi.lib().jolt_host_hyperkzg_open_sharded_blocks(ctx.h, self.srs.h, joint.h, ffi._p(p), C.c_size_t(ell), C.c_uint64(label), C.c_int32(self.rank),
                                                                      C.c_int32(self.world), C.c_size_t(self.block), self.gather_fn, self.gather_user, ffi._p(com),
                                                                      ffi._p(w), ffi._p(v), ffi._p(ch)), "jolt_host_hyperkzg_open_sharded_blocks", ctx)
        else:
            ffi._ck(ffi.lib().jolt_host_hyperkzg_open_sharded(ctx.h, self.srs.h, joint.h, ffi._p(p), C.c_size_t(ell), C.c_uint64(label), C.c_int32(self.rank),
                                                               C.c_int32(self.world), self.gather_fn, self.gather_user, ffi._p(com), ffi._p(w), ffi._p(v), ffi._p(ch)),
                    "jolt_host_hyperkzg_open_sharded", ctx)
        joint.free()
        for t in self.dense_tables:
            t.free()
        self.dense_tables = []
        return dict(com=com[: ell - 1], w=w, v=v.reshape(3, ell, 4), challenges=ch)

    def step(self, label=0):
        return dict(commit=self.commit(), open=self.open(label))


def make_point_gather(coll, world):
    """(gather_points, gather_fn, gather_user) over a Collective / NativeCollective: Python-level all-gather of (count, 12) point
    arrays and the C-level jolt_gather_fn for the opening."""
    lib = ffi.lib()

    def gather_points(local):
        flat = np.ascontiguousarray(local, dtype=np.uint64).reshape(-1)
        return np.ascontiguousarray(coll.all_gather_u64(flat)).reshape(world, -1, 12)

    if isinstance(coll, NativeCollective):
        return gather_points, C.cast(lib.jolt_comm_gather_round_sums, GATHER_FN), coll.h

    def gather_cb(user, local, count, gathered):
        try:
            loc = np.ctypeslib.as_array(C.cast(local, C.POINTER(C.c_uint64)), shape=(count * 4,)).copy()
            g = np.ascontiguousarray(coll.all_gather_u64(loc))
            C.memmove(gathered, g.ctypes.data, g.nbytes)
            return 0
        except Exception:  # noqa: BLE001 -- never unwind through the C frame
            return 4

    cb = GATHER_FN(gather_cb)
    return gather_points, cb, None
