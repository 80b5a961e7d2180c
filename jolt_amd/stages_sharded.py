"""The stage 1 / 2 / 4 / 5 / 6a / 6b / 7 operators of jolt_amd/stages.py proved by G ranks over ONE trace (one process per GPU).

The trace of G * T cycles is dealt to the ranks in contiguous blocks (rank g holds cycles [g T, (g + 1) T): jolt_amd.stages.extended_block), every point has
log2(G T) coordinates, and each operator runs in the form the hypercube sharding gives it -- the reference has no multi-GPU code, so the forms follow from the
relations themselves (north_star: "the 2^n boolean hypercube shards naturally across the GPUs, partial round-polynomial evaluations reduced with RCCL"):

  T-scale sums that are ADDITIVE over cycles (uni-skip sums, the read-RAF phase scans, every pushforward, evaluations of integer columns, input claims): each rank
      sums its block against ITS aligned block of the eq table (EqPolynomial::evals_for_aligned_block, crates/jolt-poly/src/eq.rs:238-263), ONE all-gather of the
      partial sums and a local modular sum (`gather_sum`: RCCL has no mod-r reduction) give every rank the global sums; all ranks run the same transcript on them.
  cycle-domain sumchecks (Spartan remainders, the read-RAF and bytecode cycle phases): phase A / hand-over / redundant tail of jolt_amd/distributed.py
      (prove_members_sharded): LowToHigh binds keep the first log T rounds local, one exchange of the round sums per round.
  K-sized rounds (the 128 read-RAF address rounds, booleanity / Hamming-weight address rounds, the address phases over the bytecode and RAM domains): replicated on
      every rank from the summed tables -- 16 .. 2^16 entries, the same on every rank, no communication.
  sparse read-write matrices (RAM, registers): the first log T cycle rounds on the rank's LOCAL matrix (round sums scaled by eq(w_hi, g), added over the ranks), then
      the ranks' single rows are exchanged once and stacked into the matrix of the remaining log G cycle variables (jolt_rw_matrix_create_merged), on which every rank
      finishes the cycle rounds and runs the address rounds.

tests/test_gpu_distributed.py proves a 2- and a 4-rank trace this way (all ranks on GPU 0, gloo) and compares every message with the single-process oracle twin of the
GLOBAL trace (tests/workload_oracle.py: OracleExtended over build_extended(n, n_blocks = G)).
"""
import numpy as np

from . import distributed as D
from . import ffi
from . import stages as S
from .distributed import KIND_EXPR, KIND_SPLIT_EQ, KIND_SPLIT_EQ_UNIFORM, MemberInfo, gather_sum, prove_members_sharded


class ShardedOps(S.DeviceOps):
    """The adapter of the address-domain drivers (stages.bytecode_read_raf, ram_raf_evaluation, ram_output_check) on one rank of a sharded prover: cycle-domain
    tables are the rank's blocks, pushforwards are summed over the ranks, K-sized members run replicated, the one T-sized member goes through the sharded batch."""

    def __init__(self, ext, indexes, chunk_source):
        super().__init__(ext.ctx, ffi, indexes, chunk_source)
        self.x = ext

    def eq(self, point):
        x = self.x
        return x.ctx.eq_evals_aligned_block(point, x.rank << x.n_local, 1 << x.n_local)

    def pushforward(self, which, tables):
        parts = self.indexes[which].pushforward(tables)
        out = []
        for t in parts:
            total = gather_sum(self.x.coll, t.download())
            t.free()
            out.append(self.ctx.upload(total))
        return out

    def last_value(self, which, init):
        """the word every address holds after its last access ANYWHERE in the trace: a rank only sees its block, so it reports, per address, whether it touched it (two
        passes over differing initial tables: untouched addresses echo the initial table) and its last word; the highest rank that touched an address wins"""
        x = self.x
        index, post = self.indexes[which], self.indexes[which + "_post"]
        base = init.download()
        a = index.last_value(post, init)
        va = a.download()
        a.free()
        other = self.ctx.upload(D.fr_add_vec(base, np.repeat(np.asarray(self.one).reshape(1, 4), base.shape[0], axis=0)))
        b = index.last_value(post, other)
        vb = b.download()
        b.free()
        other.free()
        touched = np.all(va == vb, axis=1)
        allv = np.ascontiguousarray(x.coll.all_gather_u64(va.reshape(-1))).reshape(x.world, -1, 4)
        allt = np.ascontiguousarray(x.coll.all_gather_u64(touched.astype(np.uint64))).reshape(x.world, -1)
        out = base.copy()
        for g in range(x.world):  # later blocks overwrite earlier ones
            m = allt[g] != 0
            out[m] = allv[g][m]
        return self.ctx.upload(out)

    def cycle_product(self, tables, n_vars, label):
        x = self.x
        n_f = len(tables)
        one = self.one
        groups = [[(None, [(one, i)]) for i in range(n_f)]]
        m = self.ctx.member_lc(tables, groups, n_f)
        m._tail = lambda tabs, scalar, w_rem: self.ctx.member_lc(tabs, groups, n_f, borrow=True)
        claim = gather_sum(x.coll, m.input_claim().reshape(1, 4))[0]
        out, fin = prove_members_sharded(self.ctx, x.coll, x.world, [m], [MemberInfo(KIND_EXPR, n_f, x.n_total, n_f)], [claim], [one], x.n_total, x.n_local, n_f,
                                         label=label, tail_log=x.tail_log, round_exchange=x.round_exchange)
        m.destroy()
        return dict(polys=out["polys"], challenges=out["challenges"], final_claim=out["final_claim"]), claim, list(fin[0])


class ShardedExtended:
    def __init__(self, ctx, n_local, rank, world, coll, seed=2026, tail_log=None, round_exchange=None, **kw):
        log_g = world.bit_length() - 1
        assert (1 << log_g) == world
        self.ctx, self.rank, self.world, self.coll = ctx, rank, world, coll
        self.n_local, self.log_g, self.n_total = n_local, log_g, n_local + log_g
        self.n_vars = self.n_total
        self.tail_log, self.round_exchange = tail_log, round_exchange
        self.p = p = S.extended_params(self.n_total, seed, **kw)
        # the machine state this rank's block starts from: replay the RAM / register generators of the blocks before it (no other column is generated)
        ram, reg, first = None, None, None
        for g in range(rank):
            st = S.extended_block(p, n_local, g, seed, ram_init=ram, reg_init=reg, only_state=True)
            if g == 0:
                first = st["ram"]["val_init"]
            ram, reg = st["ram"]["val_final"], st["registers"]["reg_final"]
        self.b = b = S.extended_block(p, n_local, rank, seed, ram_init=ram, reg_init=reg)
        self.val_init_global = b["ram"]["val_init"] if rank == 0 else first  # the memory the TRACE starts from
        self.one = ffi.host_fr_from_u64(1)
        zero = np.zeros(4, dtype=np.uint64)
        # ---- resident inputs of this rank's block
        self.outer_ints = [ctx.ints(c) for c in b["outer_cols"]]
        r = b["product_rows"]
        self.product_ints = [ctx.ints(r["left_input"]), ctx.ints(r["lookup_output"]), ctx.ints(r["jump"].astype(np.uint64)), ctx.ints(r["right_input"], "i128"),
                             ctx.ints(r["branch"].astype(np.uint64)), ctx.ints(r["next_is_noop"].astype(np.uint64))]
        self.product_ia, self.product_ib = S.product_integer_weights()
        self.product_fa, self.product_fb = S.product_field_weights(p["product_w"], lambda v: ffi.host_fr_sub(zero, v))
        ramb = b["ram"]
        self.ram_inc = ctx.ints(ramb["inc"])
        self.ram_val_init = ctx.ints(self.val_init_global)
        self.ram_cols = [ctx.ints(ramb[k]) for k in ("addresses", "pre", "post")]
        regb = b["registers"]
        self.reg_idx = ctx.onehot(np.stack([regb["rs1"], regb["rs2"], regb["rd"]]), 1 << regb["log_k"])
        self.reg_cols = [ctx.ints(regb[k]) for k in ("rs1_val", "rs2_val", "rd_pre", "rd_post")]
        lo = (regb["rd_post"] - regb["rd_pre"]).astype(np.uint64)
        hi = np.where(regb["rd_post"] < regb["rd_pre"], np.uint64(0xFFFFFFFFFFFFFFFF), np.uint64(0)).astype(np.uint64)
        self.reg_inc = ctx.ints(np.stack([lo, hi], axis=1), "i128")
        self.bool_cols = ctx.onehot(b["bool_cols"], 1 << p["log_kc"])
        lk = b["lookup"]
        self.read_raf = ctx.read_raf(lk["idx"], lk["table"], lk["raf"], S.N_LOOKUP_TABLES)
        self.lookup_lists = ffi.lookup_suffix_lists()
        tab = lk["table"].astype(np.int32)
        claim_cols = [np.where((tab >= 16 * a) & (tab < 16 * a + 16), tab - 16 * a, 0xFF).astype(np.uint8) for a in range(3)]
        self.lookup_claim_columns = ctx.onehot(np.stack(claim_cols + [np.where(lk["raf"] != 0, 0, 0xFF).astype(np.uint8)]), 16)
        bc = b["bytecode"]
        self.pc_ints = ctx.ints(bc["push_pc"])
        self.pc_chunks = ctx.onehot(bc["chunk_cols"], 1 << p["bytecode"]["chunk_bits"])
        self.first_pc = int(np.ascontiguousarray(coll.all_gather_u64(np.array([bc["push_pc"][0]], dtype=np.uint64))).reshape(world, -1)[0, 0])
        # ---- input claims (the previous stage's output claims in a real proof): every rank's share against its eq block, summed over the ranks once, untimed
        self.claims = {}
        T = 1 << n_local
        eq_block = lambda point, streams_log=0: ctx.eq_evals_aligned_block(point, rank << (n_local + streams_log), T << streams_log)
        for name, cols, fa, fb, tau, kernel, streams in (("outer", self.outer_ints, p["outer_wa"], p["outer_wb"], p["outer_tau"], p["outer_kernel"], 2),
                                                         ("product", self.product_ints, self.product_fa, self.product_fb, p["product_tau"], p["product_kernel"], 1)):
            az, bz = ctx.r1cs_materialize_small(cols, fa, fb, streams=streams)
            eq = eq_block(tau, streams - 1)
            m = ctx.member_lc([eq, az, bz], [[(None, [(kernel, 0)]), (None, [(self.one, 1)]), (None, [(self.one, 2)])]], 3, borrow=True)
            self.claims[name] = self.gsum1(m.input_claim())
            m.destroy()
            for t in (eq, az, bz):
                t.free()
        acc = ramb["addresses"] != S.NO_ACCESS
        eq = eq_block(p["ram_tau"])
        pre, post = ctx.from_u64(np.where(acc, ramb["pre"], 0).astype(np.uint64)), ctx.from_u64(np.where(acc, ramb["post"], 0).astype(np.uint64))
        m = ctx.member_lc([eq, pre, post], [[(None, [(self.one, 0)]), (None, [(self.one, 1), (p["ram_gamma"], 2)])]], 2, borrow=True)
        self.claims["ram"] = self.gsum1(m.input_claim())
        m.destroy()
        for t in (eq, pre, post):
            t.free()
        eq = eq_block(p["registers_r_cycle"])
        t_post, t_rs1, t_rs2 = (ctx.table_from_ints(self.reg_cols[k]) for k in (3, 0, 1))
        g = p["registers_gamma"]
        m = ctx.member_lc([eq, t_post, t_rs1, t_rs2], [[(None, [(self.one, 0)]), (None, [(self.one, 1), (g, 2), (ffi.host_fr_mul(g, g), 3)])]], 2, borrow=True)
        self.claims["registers"] = self.gsum1(m.input_claim())
        m.destroy()
        for t in (eq, t_post, t_rs1, t_rs2):
            t.free()
        self.claims["lookup"] = None
        ctx.synchronize()

    # ---- helpers ----------------------------------------------------------------------------------------------------------------
    def gsum(self, values):
        return gather_sum(self.coll, values)

    def gsum1(self, value):
        return gather_sum(self.coll, np.asarray(value, dtype=np.uint64).reshape(1, 4))[0]

    def eq_block(self, point):
        return self.ctx.eq_evals_aligned_block(point, self.rank << self.n_local, 1 << self.n_local)

    def shard_scale(self, w):
        """eq(w_hi, rank): the weight of this rank's block under the top log G coordinates of a point (big-endian)"""
        sc = self.one
        for j in range(self.log_g):
            bit = (self.rank >> (self.log_g - 1 - j)) & 1
            sc = ffi.host_fr_mul(sc, w[j] if bit else ffi.host_fr_sub(self.one, w[j]))
        return sc

    def scaled(self, values, scalar):
        v = np.asarray(values, dtype=np.uint64).reshape(-1, 4)
        return np.stack([ffi.host_fr_mul(x, scalar) for x in v])

    def prove1(self, member, info, claim, n_total, n_local, degree, label):
        out, fin = prove_members_sharded(self.ctx, self.coll, self.world, [member], [info], [claim], [self.one], n_total, n_local, degree, label=label, tail_log=self.tail_log,
                                         round_exchange=self.round_exchange)
        return out, fin[0]

    # ---- the operators ---------------------------------------------------------------------------------------------------------
    def spartan(self, cols, iwa, iwb, fa, fb, tau, kernel, claim, streams, label):
        ctx, log_s, log_g = self.ctx, streams - 1, self.log_g
        n_loc, n_tot = self.n_local + log_s, self.n_total + log_s
        eq = ctx.eq_evals_aligned_block(tau, self.rank << n_loc, 1 << n_loc)
        sums = self.gsum(ctx.r1cs_uniskip_sums_small(cols, eq, iwa, iwb, streams=streams))  # additive over cycles
        eq.free()
        tr = ffi.HostTranscript(label)
        tr.append(sums)
        r0 = tr.challenge()
        tr.close()
        az, bz = ctx.r1cs_materialize_small(cols, fa, fb, streams=streams)
        w_local = np.ascontiguousarray(tau[log_g:])
        member = ctx.member_split_eq_product_sharded(az, bz, w_local, shard_scale=self.shard_scale(tau))
        member._tail = lambda tabs, scalar, w_rem: ctx.member_split_eq_product(tabs[0], tabs[1], w_rem, scale=scalar, borrow=True)
        out, _ = self.prove1(member, MemberInfo(KIND_SPLIT_EQ, 3, n_tot, 2, w=tau, scale=kernel), claim, n_tot, n_loc, 3, label + 1)
        member.destroy()
        az.free()
        bz.free()
        point = out["challenges"][n_tot - self.n_total:][::-1]  # the cycle coordinates, most significant first
        local = ctx.ints_evaluate(cols, np.ascontiguousarray(point[log_g:]))
        values = self.gsum(self.scaled(local, self.shard_scale(point)))
        return dict(sums=sums, r0=r0, polys=out["polys"], challenges=out["challenges"], final_claim=out["final_claim"], values=values)

    def _rw(self, local, registers, log_k, w, gamma, claim, label, val_init):
        """both sparse matrices: n_local cycle rounds on the rank's local matrix, the exchange of the rows, the merged matrix for everything after"""
        ctx, n_local, log_g, world = self.ctx, self.n_local, self.log_g, self.world
        if log_g == 0:  # one rank: the matrix is the whole trace's, nothing to merge
            tr = ffi.HostTranscript(label)
            S._sub = ffi.host_fr_sub
            out = S.rw_rounds(lambda rnd, bind: local.prove_round(bind), local.finish, local.final_values, n_local, log_k, claim, tr, ffi.host_gruen_poly_deg_3,
                              ffi.host_univariate_from_evals, ffi.host_univariate_evaluate, four_point_address=registers)
            tr.close()
            local.free()
            return out
        ffi.rw_hold_row(local)
        tr = ffi.HostTranscript(label)
        scale = self.shard_scale(w)
        sub = ffi.host_fr_sub
        polys, chal, bind = [], [], None
        for rnd in range(n_local):
            evals, aux = local.prove_round(bind)
            e = self.gsum(self.scaled(evals[:2], scale))
            poly = ffi.host_gruen_poly_deg_3(aux[0], aux[1], e[0], e[1], claim)
            tr.append(poly)
            bind = tr.challenge()
            claim = ffi.host_univariate_evaluate(poly, bind)
            polys.append(poly)
            chal.append(bind)
        ffi.rw_bind(local, bind)
        row = ffi.rw_export_row(local, registers)
        local.free()
        # the ranks' rows, stacked in rank order: counts first, then the padded arrays
        counts = np.ascontiguousarray(self.coll.all_gather_u64(np.array([row["cols"].shape[0]], dtype=np.uint64))).reshape(world).astype(np.int64)
        cap = int(max(1, counts.max()))

        def gather(a, width):
            pad = np.zeros((cap, width), dtype=np.uint64)
            a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, width)
            pad[: a.shape[0]] = a
            allv = np.ascontiguousarray(self.coll.all_gather_u64(pad.reshape(-1))).reshape(world, cap, width)
            return np.concatenate([allv[g, : counts[g]] for g in range(world)])

        cols, prev, nxt = gather(row["cols"], 1)[:, 0], gather(row["prev"], 1)[:, 0], gather(row["next"], 1)[:, 0]
        val, ra = gather(row["val"], 4), gather(row["ra"], 4)
        wa = gather(row["wa"], 4) if registers else None
        rows = np.concatenate([np.full(counts[g], g, dtype=np.uint64) for g in range(world)])
        inc = np.ascontiguousarray(self.coll.all_gather_u64(np.asarray(row["inc"], dtype=np.uint64).reshape(-1))).reshape(world, 4)
        merged = ffi.MergedRw(ctx, registers, log_g, log_k, rows, cols, prev, nxt, val, ra, wa, inc, val_init, np.ascontiguousarray(w[:log_g]), row["scalar"], gamma)
        bind = None
        for rnd in range(log_g + log_k):
            evals, aux = merged.prove_round(bind)
            if rnd < log_g:
                poly = ffi.host_gruen_poly_deg_3(aux[0], aux[1], evals[0], evals[1], claim)
            elif registers:  # every point sampled (registers_read_write/mod.rs:217-252)
                poly = ffi.host_univariate_from_evals(np.stack([evals[k] for k in range(4)]))
            else:
                poly = ffi.host_univariate_from_evals(np.stack([evals[0], sub(claim, evals[0]), evals[1]]))
            tr.append(poly)
            bind = tr.challenge()
            claim = ffi.host_univariate_evaluate(poly, bind)
            polys.append(poly)
            chal.append(bind)
        merged.finish(bind)
        out = dict(polys=polys, challenges=np.stack(chal), final_claim=claim, final_values=merged.final_values())
        merged.free()
        tr.close()
        return out

    def ram_read_write(self, label):
        ctx, p, ram = self.ctx, self.p, self.b["ram"]
        inc, val_init = ctx.table_from_ints(self.ram_inc), ctx.table_from_ints(self.ram_val_init)
        tau = p["ram_tau"]
        local = ctx.rw_matrix(self.ram_cols[0], self.ram_cols[1], self.ram_cols[2], inc, val_init, np.ascontiguousarray(tau[self.log_g:]), p["ram_gamma"])
        inc.free()
        out = self._rw(local, False, ram["log_k"], tau, p["ram_gamma"], self.claims["ram"], label, val_init)
        val_init.free()
        return out

    def registers_read_write(self, label):
        ctx, p, reg = self.ctx, self.p, self.b["registers"]
        n_total, log_k, log_g = self.n_total, reg["log_k"], self.log_g
        inc = ctx.table_from_ints(self.reg_inc)
        w = p["registers_r_cycle"]
        local = ctx.registers_rw(self.reg_idx, *self.reg_cols, inc, np.ascontiguousarray(w[log_g:]), p["registers_gamma"])
        inc.free()
        out = self._rw(local, True, log_k, w, p["registers_gamma"], self.claims["registers"], label, None)
        # the operand claims: the index columns at (r_address, r_cycle) -- each rank evaluates its block at the low coordinates, weighted by eq(r_cycle_hi, rank)
        r_cycle = out["challenges"][:n_total][::-1]
        eq_adr = ctx.upload(ffi.host_eq_evals(out["challenges"][n_total:][::-1]))
        scale = self.shard_scale(r_cycle)
        claims = []
        for q in (0, 1):
            col = self.reg_idx.materialize(q, eq_adr)
            claims.append(ffi.host_fr_mul(ctx.evaluate(col, np.ascontiguousarray(r_cycle[log_g:])), scale))
            col.free()
        eq_adr.free()
        out["operand_claims"] = self.gsum(np.stack(claims))
        return out

    def _masses(self, point):
        eq = self.eq_block(point)
        g = self.bool_cols.pushforward(eq)
        eq.free()
        masses = self.gsum(g.download()).reshape(self.b["bool_cols"].shape[0], 1 << self.p["log_kc"], 4)
        g.free()
        return masses

    def booleanity_address(self, label):
        bo = self.p["booleanity"]
        masses = self._masses(bo["reference_cycle"])
        tr = ffi.HostTranscript(label)
        out = S.booleanity_address_rounds(ffi.HostBooleanityAddress(masses, bo["gamma"], bo["reference_address"]), bo["log_k"], tr, ffi.host_univariate_from_evals,
                                          ffi.host_univariate_evaluate)
        tr.close()
        out["masses"] = masses
        return out

    def hamming_weight(self, label):
        bo, hw = self.p["booleanity"], self.p["hamming"]
        masses = self._masses(hw["r_cycle"])
        tr = ffi.HostTranscript(label)
        out = S.hamming_weight_rounds(ffi.HostHammingWeight(masses, hw["gamma"], hw["r_address"], hw["virtualization_points"]), bo["log_k"], tr, ffi.host_univariate_from_evals,
                                      ffi.host_univariate_evaluate, ffi.host_fr_sub)
        tr.close()
        out["masses"] = masses
        return out

    def instruction_read_raf(self, label):
        """rows shard by cycles: the phase scans are additive over the ranks (one exchange of 100 x 256 sums per phase), the 128 address rounds run replicated on every
        rank from the summed scans (same transcript, same challenges), condensation and cycle columns stay local, the cycle rounds are a sharded batch"""
        ctx, p, rr = self.ctx, self.p, self.read_raf
        tr = ffi.HostTranscript(label)
        w = p["lookup_reduction"]
        u = self.eq_block(w)
        present = np.zeros(S.N_LOOKUP_TABLES, dtype=np.uint8)
        present[p["lookup_present"]] = 1
        state = ffi.HostReadRafAddress(p["lookup_gamma"], present)
        claim = self.claims["lookup"]
        v_tables, scans, messages, challenges = [], [], [], []
        for phase in range(S.PHASES):
            suffix_len = S.ADDRESS_BITS - 8 * (phase + 1)
            if phase:
                rr.condense(u, v_tables[-1], suffix_len + 8)
            raf, suf = rr.phase_scan(u, suffix_len, S.ADDRESS_BITS, self.lookup_lists)
            both = self.gsum(np.concatenate([raf.reshape(-1, 4), suf.reshape(-1, 4)]))
            raf, suf = both[: 6 * 256].reshape(6, 256, 4), both[6 * 256:].reshape(-1, 256, 4)
            state.init_phase(phase, raf, suf)
            if claim is None:
                e = state.message()
                claim = self.claims["lookup"] = ffi.host_fr_add(e[0], e[1])
            claim, coeffs, chal = state.prove_phase(claim, tr)
            scans.append((raf, suf))
            messages.append(coeffs)
            challenges.append(chal)
            v_tables.append(state.v_table(phase))
        u.free()
        vt = np.stack(v_tables)
        table_values, raf_interleaved, raf_identity = state.finish()
        state.close()
        combined, ra = rr.cycle_tables(table_values, raf_interleaved, raf_identity, vt, S.ADDRESS_BITS, p["ra_count"])
        n_f = 1 + p["ra_count"]
        groups = [[(None, [(self.one, i)]) for i in range(n_f)]]
        member = ctx.member_lc([combined] + ra, groups, n_f, eq_point=np.ascontiguousarray(w[self.log_g:]), shard_scale=self.shard_scale(w))
        member._tail = lambda tabs, scalar, w_rem: ctx.member_lc(tabs, groups, n_f, borrow=True, eq_point=w_rem, eq_scale=scalar)
        out, fin = self.prove1(member, MemberInfo(KIND_SPLIT_EQ_UNIFORM, n_f + 1, self.n_total, n_f, w=w), claim, self.n_total, self.n_local, n_f + 1, label + 1)
        member.destroy()
        instruction_ra = np.stack(list(fin)[1:n_f])
        eq_cycle = self.eq_block(out["challenges"][::-1])
        flags = self.lookup_claim_columns.pushforward(eq_cycle)
        flag_claims = self.gsum(flags.download()).reshape(4, 16, 4)
        flags.free()
        eq_cycle.free()
        tr.close()
        pres = p["lookup_present"]
        return dict(lookup_table_flags=flag_claims[:3].reshape(48, 4)[pres], instruction_raf_flag=flag_claims[3][0], instruction_ra=instruction_ra, scans=scans,
                    address_polys=np.concatenate(messages), address_challenges=np.concatenate(challenges), v_tables=vt, table_values=table_values[pres],
                    raf_values=np.stack([raf_interleaved, raf_identity]), cycle_claim=claim, polys=out["polys"], challenges=out["challenges"], final_claim=out["final_claim"])

    def address_domain(self, label):
        ctx, p, b = self.ctx, self.p, self.b
        ram = dict(log_k=p["ram_log_k"], val_init=self.val_init_global)
        bc = dict(p["bytecode"], first_pc=self.first_pc, push_pc=b["bytecode"]["push_pc"])
        indexes = {"pc": ctx.key_index(self.pc_ints, 1 << bc["log_k"]), "ram": ctx.key_index(self.ram_cols[0], 1 << ram["log_k"]), "ram_post": self.ram_cols[2]}
        ops = ShardedOps(self, indexes, self.pc_chunks)
        out = {"bytecode_read_raf": S.bytecode_read_raf(ops, bc, self.n_total, label),
               "ram_raf_evaluation": S.ram_raf_evaluation(ops, ram, p["ram_raf"], label + 10),
               "ram_output_check": S.ram_output_check(ops, ram, p["ram_output"], label + 20)}
        indexes["pc"].free()
        indexes["ram"].free()
        return out

    def prove(self, label=0):
        p = self.p
        return {
            **self.address_domain(label + 500),
            "spartan_outer": self.spartan(self.outer_ints, p["outer_iwa"], p["outer_iwb"], p["outer_wa"], p["outer_wb"], p["outer_tau"], p["outer_kernel"], self.claims["outer"], 2,
                                          label + 100),
            "spartan_product": self.spartan(self.product_ints, self.product_ia, self.product_ib, self.product_fa, self.product_fb, p["product_tau"], p["product_kernel"],
                                            self.claims["product"], 1, label + 200),
            "ram_read_write": self.ram_read_write(label + 300),
            "registers_read_write": self.registers_read_write(label + 350),
            "instruction_read_raf": self.instruction_read_raf(label + 400),
            "booleanity_address": self.booleanity_address(label + 450),
            "hamming_weight": self.hamming_weight(label + 470),
        }

    def close(self):
        for c in self.outer_ints + self.product_ints + self.ram_cols + self.reg_cols + [self.ram_inc, self.ram_val_init, self.reg_inc, self.reg_idx, self.bool_cols, self.pc_ints,
                                                                                        self.pc_chunks]:
            c.free()
        self.read_raf.free()
        self.lookup_claim_columns.free()
