"""Hypercube-sharded batched sumcheck: one process per GPU, rank g owns block g of every table (DESIGN.md section 6,
SURVEY.md section 8e).

A batch over n_total = n_local + log2(G) variables is proved in two phases through the resumable round loop of
jolt_amd/csrc/batch.hip:
  phase A  rounds 0 .. n_local-1: every rank computes the round sums of its block (LowToHigh binds pair (2y, 2y+1), so
           they stay local); ONE all-gather of a few hundred bytes per batch round, modular sum on every rank, identical
           transcripts -> identical challenges;
  phase B  the single remaining entry of every local table is all-gathered once, each rank rebuilds the G-entry tables
           and finishes the last log2(G) rounds redundantly (no communication).
The collective is torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests); the
payload is latency-bound (RCCL has no mod-r reduction, hence all-gather + local sum).

The local compute is pluggable so that the N>1 logic can be tested on CPU ranks: `DeviceShard` drives device members
through the C ABI; tests substitute an oracle-backed shard.
"""
import ctypes as C

import numpy as np

from . import ffi
from . import workload as W

KIND_EXPR, KIND_EXPR_SKIP, KIND_SPLIT_EQ, KIND_SPLIT_EQ_UNIFORM = 0, 1, 2, 3

LOCAL_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_size_t), C.c_size_t, C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t)
GATHER_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


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


class MemberInfo:
    def __init__(self, kind, degree, rounds, n_tables, w=None, scale=None):
        self.kind, self.degree, self.rounds, self.n_tables, self.w, self.scale = kind, degree, rounds, n_tables, w, scale

    @property
    def n_evals(self):
        if self.kind == KIND_SPLIT_EQ_UNIFORM:
            return self.degree - 1
        return 2 if self.kind == KIND_SPLIT_EQ else (self.degree if self.kind == KIND_EXPR_SKIP else self.degree + 1)


def prove_batch_sharded(ctx_handle, infos, claims, coeffs, n_total, n_local, max_degree, world, coll, shard, label=0,
                        challenge_mode=0):
    """infos: [MemberInfo] (global rounds = n_total for every member here); shard: local backend with
         round(active_idx, binds) -> np (total,4);  flush(binds) ; finals() -> np (sum n_tables, 4);
         make_tail(gathered_finals (world, sum n_tables, 4), split_eq_scalars) -> tail backend with .round()/.flush()/.close()
       Returns dict(polys, challenges, member_claims, final_claim)."""
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

    def run(n_rounds, w):
        st = lib.jolt_host_batch_run(batch, None, C.c_size_t(n_rounds), C.c_int32(w), gcb if w > 1 else None, lcb, None)
        if st:
            if state["err"] is not None:
                raise state["err"]
            raise ffi.JoltError(st, "jolt_host_batch_run")

    def flush():
        binds = ffi.fr_array(n)
        has = (C.c_int32 * n)()
        st = lib.jolt_host_batch_flush_binds(batch, None, _p(binds), has)
        if st:
            raise ffi.JoltError(st, "jolt_host_batch_flush_binds")
        state["backend"].flush([binds[i] if has[i] else None for i in range(n)])

    log_g = n_total - n_local
    run(n_local, world)
    flush()
    if log_g > 0:
        finals = np.ascontiguousarray(shard.finals(), dtype=np.uint64).reshape(-1, 4)
        gathered = coll.all_gather_u64(finals).reshape(world, -1, 4)
        scalars = []
        for i, info in enumerate(infos):
            if info.kind in (KIND_SPLIT_EQ, KIND_SPLIT_EQ_UNIFORM):
                o = ffi.fr_array(1)
                lib.jolt_host_batch_split_eq_scalar(batch, C.c_size_t(i), _p(o))
                scalars.append(o[0])
            else:
                scalars.append(None)
        tail = shard.make_tail(gathered, scalars)
        state["backend"] = tail
        run(log_g, 1)
        flush()
        tail.close()
    polys, chal = ffi.fr_array(n_total * (max_degree + 1)), ffi.fr_array(n_total)
    mclaims, final = ffi.fr_array(n), ffi.fr_array(1)
    st = lib.jolt_host_batch_end(batch, _p(polys), _p(chal), _p(mclaims), _p(final))
    if st:
        raise ffi.JoltError(st, "jolt_host_batch_end")
    return dict(polys=polys.reshape(n_total, max_degree + 1, 4), challenges=chal, member_claims=mclaims, final_claim=final[0])


# ---------------------------------------------------------------------------------------------------------------------
# device backend
# ---------------------------------------------------------------------------------------------------------------------
class DeviceShard:
    """Local backend over device members (jolt_round_group_prove / jolt_round_group_finish / ..._final_values)."""

    def __init__(self, ctx, members, rebuild=None):
        self.ctx, self.members, self.rebuild = ctx, members, rebuild

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

    def finals(self):
        total = sum(m.n_tables for m in self.members)
        out = ffi.fr_array(total)
        hs = (C.c_void_p * len(self.members))(*[m.h for m in self.members])
        st = ffi.lib().jolt_round_group_final_values(self.ctx.h, hs, C.c_size_t(len(self.members)), _p(out), C.c_size_t(total))
        if st:
            raise ffi.JoltError(st, "jolt_round_group_final_values")
        return out

    def make_tail(self, gathered, scalars):
        return self.rebuild(gathered, scalars)

    def close(self):
        pass


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

    def __init__(self, ctx, n_local, rank, world, dist, seed=2026, coll=None):
        self.ctx, self.n_local, self.rank, self.world = ctx, n_local, rank, world
        log_g = world.bit_length() - 1
        assert (1 << log_g) == world, "world size must be a power of two"
        self.n_total = n_local + log_g
        if coll is None:
            import torch
            coll = Collective(dist, world, torch.device("cuda", torch.cuda.current_device()))
        self.coll = coll
        spec = build_sharded_spec(n_local, rank, world, seed)
        self.members_spec = spec["members"]
        one = ffi.host_fr_from_u64(1)
        zero = np.zeros(4, dtype=np.uint64)
        self.resolver = W.Resolver(spec["gammas"], one, ffi.host_fr_mul, lambda x: ffi.host_fr_sub(zero, x))
        self.tables = {}
        skip = {ms.tables[0] for ms in self.members_spec if ms.uniform is not None}
        skip -= {t for ms in self.members_spec for t in (ms.tables if ms.uniform is None else ms.tables[1:])}
        for name, t in spec["tables"].items():
            if name in skip:
                continue
            if t["kind"] == "u64":
                self.tables[name] = ctx.from_u64(t["data"])
            elif t["kind"] == "i64":
                self.tables[name] = ctx.from_i64(t["data"])
            else:  # the aligned block of eq(point, .) owned by this rank (EqPolynomial::evals_for_aligned_block)
                self.tables[name] = ctx.eq_evals_aligned_block(t["point"], rank << n_local, 1 << n_local)
        self.members, self.infos, self.stages = [], [], {}
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
                m = ctx.member_split_eq_uniform(tabs[1:], V, F, coeffs, w[log_g:], shard_scale=shard_scale_of(w), borrow=True)
                m._uniform = (V, F, coeffs)
                self.infos.append(MemberInfo(KIND_SPLIT_EQ_UNIFORM, F + 1, self.n_total, V * F, w=w))
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
        ctx.synchronize()
        # global input claims = sum over ranks of the local claims (one all-gather at setup; in the real prover they are
        # the previous stage's output claims)
        local_claims = np.stack([self._local_claim(i) for i in range(len(self.members))])
        allc = self.coll.all_gather_u64(local_claims).reshape(world, len(self.members), 4)
        self.claims = []
        for i in range(len(self.members)):
            acc = zero
            for r in range(world):
                acc = ffi.host_fr_add(acc, allc[r, i])
            self.claims.append(acc)
        self.batch_coeffs = spec["batch_coeffs"]
        self.n_tables = sum(len(ms.tables) for ms in self.members_spec)
        self._tails = {}

    def _local_claim(self, i):
        """This rank's share of member i's input claim.  Split-eq members are summed against the rank's aligned block of the
        dense eq table (the members' own claim helper does not know the shard scale)."""
        m, ms = self.members[i], self.members_spec[i]
        if not m.split_eq:
            return m.input_claim()
        one = ffi.host_fr_from_u64(1)
        eq = self.ctx.eq_evals_aligned_block(self.infos[i].w, self.rank << self.n_local, 1 << self.n_local)
        if ms.uniform is not None:
            V, F, coeffs = m._uniform
            tabs = [eq] + [self.tables[t] for t in ms.tables[1:]]
            groups = [[(None, [(coeffs[v], 0)])] + [(None, [(one, 1 + v * F + k)]) for k in range(F)] for v in range(V)]
            tmp = self.ctx.member_lc(tabs, groups, F + 1, borrow=True)
        else:
            a, b, _ = ms.split_eq
            tabs = [eq, self.tables[ms.tables[a]], self.tables[ms.tables[b]]]
            tmp = self.ctx.member_lc(tabs, [[(None, [(one, 0)]), (None, [(one, 1)]), (None, [(one, 2)])]], 3, borrow=True)
        c = tmp.input_claim()
        tmp.destroy()
        eq.free()
        return c

    def _tail_backend(self, stage, idxs, gathered, scalars):
        """G-entry tables rebuilt from the gathered finals; the arena and the tail members are created once per stage and
        only refilled afterwards."""
        ctx, world = self.ctx, self.world
        log_g = world.bit_length() - 1
        key = stage
        order = []  # (member idx, table k) in finals order
        for i in idxs:
            for k in range(self.members[i].n_tables):
                order.append((i, k))
        flat = np.ascontiguousarray(gathered.transpose(1, 0, 2))  # (n_tables_total, world, 4): table-major, rank = top variables
        if key not in self._tails:
            arena = ctx.upload(flat.reshape(-1, 4))
            slices, members, pos = [], [], 0
            for i in idxs:
                m = self.members[i]
                tabs = []
                for _ in range(m.n_tables):
                    h = C.c_void_p()
                    st = ffi.lib().jolt_table_slice(ctx.h, arena.h, C.c_size_t(pos * world), C.c_size_t(world), C.byref(h))
                    if st:
                        raise ffi.JoltError(st, "jolt_table_slice")
                    tabs.append(ffi.Table(ctx, h))
                    pos += 1
                slices.extend(tabs)
                if getattr(m, "uniform", False):
                    V, F, coeffs = m._uniform
                    tm = ctx.member_split_eq_uniform(tabs, V, F, coeffs, self.infos[i].w[:log_g], scale=scalars[idxs.index(i)], borrow=True)
                elif m.split_eq:
                    tm = ctx.member_split_eq_product(tabs[0], tabs[1], self.infos[i].w[:log_g], scale=scalars[idxs.index(i)], borrow=True)
                else:
                    tm = ctx.member_lc(tabs, m._groups, m.degree, borrow=True, skip_one=True)
                members.append(tm)
            self._tails[key] = (arena, slices, members)
        else:
            arena, slices, members = self._tails[key]
            st = ffi.lib().jolt_table_write(ctx.h, arena.h, C.c_size_t(0), _p(flat.reshape(-1, 4)), C.c_size_t(flat.shape[0] * world))
            if st:
                raise ffi.JoltError(st, "jolt_table_write")
            for k, i in enumerate(idxs):
                if self.members[i].split_eq:  # its initial scalar depends on this proof's challenges: rebuild that member
                    members[k].destroy()
                    base = sum(self.members[j].n_tables for j in idxs[:k])
                    if getattr(self.members[i], "uniform", False):
                        V, F, coeffs = self.members[i]._uniform
                        members[k] = ctx.member_split_eq_uniform(slices[base:base + V * F], V, F, coeffs, self.infos[i].w[:log_g], scale=scalars[k], borrow=True)
                    else:
                        members[k] = ctx.member_split_eq_product(slices[base], slices[base + 1], self.infos[i].w[:log_g], scale=scalars[k], borrow=True)
                else:
                    members[k].reset()
        return DeviceShard(ctx, members)

    def prove(self, label=0):
        outs = {}
        for stage, idxs in sorted(self.stages.items()):
            ms = [self.members[i] for i in idxs]
            infos = [self.infos[i] for i in idxs]
            deg = max(m.degree for m in ms)
            shard = DeviceShard(self.ctx, ms, rebuild=lambda g, s, stage=stage, idxs=idxs: self._tail_backend(stage, idxs, g, s))
            outs[stage] = prove_batch_sharded(self.ctx.h, infos, [self.claims[i] for i in idxs], [self.batch_coeffs[i] for i in idxs],
                                              self.n_total, self.n_local, deg, self.world, self.coll, shard, label=label + stage)
        for m in self.members:
            m.reset()
        return outs
