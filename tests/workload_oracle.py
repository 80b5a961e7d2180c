"""CPU-oracle instantiation of jolt_amd.workload (TEST INFRASTRUCTURE: tests and bench.py's cpu_baseline only)."""
import numpy as np

import oracle_lib as O
from jolt_amd import workload as W


def make_table(spec):
    if spec.kind == "u64":
        return O.fr_from_u64(spec.data)
    if spec.kind == "i64":
        return O.fr_from_i64(spec.data)
    if spec.kind == "eq":
        return O.eq_evals(spec.point)
    if spec.kind == "lt":
        return O.lt_evals(spec.point)
    if spec.kind == "eq1":
        return O.eq_plus_one_evals(spec.point)[1]
    if spec.kind == "onehot":  # the dense address-folded column the lazy member never materialises at this size
        table = O.eq_evals(spec.point)
        out = np.zeros((spec.data.shape[0], 4), dtype=np.uint64)
        hot = spec.data != 0xFF
        out[hot] = table[spec.data[hot]]
        return out
    raise ValueError(spec.kind)


def resolver(gammas):
    one = O.to_mont([1])[0]
    mul = lambda a, b: O.fr_mul(np.asarray(a).reshape(1, 4), np.asarray(b).reshape(1, 4))[0]
    neg = lambda a: O.fr_neg(np.asarray(a).reshape(1, 4))[0]
    return W.Resolver(gammas, one, mul, neg), one, mul


class OracleWorkload:
    def __init__(self, n_vars, seed=2026, only_stages=None, **kw):
        """only_stages: materialise (and prove) just these stages -- the T = 2^22 parity test checks a stage, not the catalogue"""
        self.n_vars = n_vars
        self.tables_spec, self.members_spec, gammas = W.build(n_vars, seed, **kw)
        self.res, self.one, self.mul = resolver(gammas)
        wanted = {t for ms in self.members_spec if only_stages is None or ms.stage in only_stages for t in ms.tables}
        self.tables = {name: make_table(spec) for name, spec in self.tables_spec.items() if name in wanted}
        rng = np.random.default_rng(seed + 1)
        self.batch_coeffs = [W.rand_fr(1, rng)[0] for _ in self.members_spec]
        self.stages = {}
        for i, ms in enumerate(self.members_spec):
            if only_stages is None or ms.stage in only_stages:
                self.stages.setdefault(ms.stage, []).append(i)

    def member(self, i):
        ms = self.members_spec[i]
        tabs = [self.tables[t] for t in ms.tables]
        if ms.split_eq is not None:
            a, b, w = ms.split_eq
            return O.Member.gruen_product(tabs[a], tabs[b], w)
        terms = W.expand_to_flat_terms(self.res.groups(ms.groups), self.mul, self.one)
        return O.Member.expr(tabs, terms, ms.degree)

    def prove(self, label=0):
        outs = {}
        for stage, idxs in sorted(self.stages.items()):
            ms = [self.member(i) for i in idxs]
            claims = [m.input_claim() for m in ms]
            deg = max(m.degree for m in ms)
            outs[stage] = O.prove_batch(ms, claims, [self.batch_coeffs[i] for i in idxs], [0] * len(ms), self.n_vars, deg,
                                        label=label + stage)
            outs[stage]["claims"] = claims
        return outs


class OracleOps:
    """jolt_amd.stages.DeviceOps on the CPU oracle: tables are numpy (n, 4) arrays, the pushforwards are the restatements of oracle/address_ops.c (the bytecode one in the
    reference's split-eq two-table form when the weights are eq tables of known points, fold_cycles otherwise), members the dense ones of oracle/sumcheck.c."""

    def __init__(self, indexes, chunk_cols, k_chunk):
        self.indexes, self.chunk_cols, self.k_chunk = indexes, chunk_cols, k_chunk
        self.one = O.to_mont([1])[0]
        v = lambda a: np.asarray(a, dtype=np.uint64).reshape(1, 4)
        self.mul = lambda a, b: O.fr_mul(v(a), v(b))[0]
        self.add = lambda a, b: O.fr_add(v(a), v(b))[0]
        self.sub = lambda a, b: O.fr_sub(v(a), v(b))[0]
        self.host_eq = O.eq_evals
        self.points = {}

    def eq(self, point):
        t = O.eq_evals(point) if len(point) else O.to_mont([1])
        self.points[id(t)] = np.asarray(point)
        return t

    def upload(self, values):
        return np.ascontiguousarray(values, dtype=np.uint64).reshape(-1, 4).copy()

    def u64_table(self, values):
        return O.fr_from_u64(np.ascontiguousarray(values, dtype=np.uint64))

    def pushforward(self, which, tables):
        keys, k = self.indexes[which]
        pts = [self.points.get(id(t)) for t in tables]
        if which == "pc" and all(p is not None and len(p) for p in pts):  # stage_pushforwards: all stages in one walk over the split eq tables
            return list(O.stage_pushforwards(np.stack(pts), keys, k))
        return [O.fold_cycles(keys, k, t) for t in tables]

    def last_value(self, which, init):
        keys, k = self.indexes[which]
        return O.last_value(keys, self.indexes[which + "_post"], k, init)

    def materialize_chunk(self, i, eq_chunk):
        return O.onehot_values(eq_chunk, 1, self.k_chunk, self.chunk_cols[i], self.chunk_cols.shape[1])

    def rlc(self, tables, scalars):
        acc = np.zeros_like(tables[0])
        for t, c in zip(tables, scalars):
            acc = O.fr_add(acc, O.fr_mul(t, np.repeat(np.asarray(c).reshape(1, 4), t.shape[0], axis=0)))
        return acc

    def member_expr(self, tables, terms, degree):
        return O.Member.expr(tables, terms, degree)

    def member_gruen_product(self, a, b, w):
        return O.Member.gruen_product(a, b, w)

    def cycle_product(self, tables, n_vars, label):
        n_f = len(tables)
        member = self.member_expr(tables, [(self.one, list(range(n_f)))], n_f)
        claim = member.input_claim()
        out = self.prove(member, claim, n_vars, n_f, label)
        fin = self.final_values(member)
        self.destroy(member)
        return out, claim, fin

    def prove(self, member, claim, n_vars, degree, label):
        out = O.prove_batch([member], [claim], [self.one], [0], n_vars, degree, label=label)
        return dict(polys=out["polys"], challenges=out["challenges"], final_claim=out["final_claim"])

    def final_values(self, member):
        return list(member.final_values())

    def evaluate(self, table, point):
        return O.poly_evaluate(table, point) if len(point) else table[0]

    def destroy(self, member):
        member.close()

    def free(self, table):
        self.points.pop(id(table), None)


class OracleExtended:
    """jolt_amd.stages.DeviceExtended on the CPU oracle: the same description (build_extended), the same operator drivers, every T-scale
    quantity from oracle/r1cs.c, rw_matrix.c, read_raf.c and the dense members of oracle/sumcheck.c."""

    def __init__(self, n_vars, seed=2026, description=None, **kw):
        from jolt_amd import stages as S
        self.S, self.n_vars = S, n_vars
        self.d = description if description is not None else S.build_extended(n_vars, seed, **kw)
        self.one = O.to_mont([1])[0]
        self.neg = lambda a: O.fr_neg(np.asarray(a).reshape(1, 4))[0]

    def _spartan(self, inputs, eq_sums, tables, tau, kernel, label):
        sums = eq_sums()
        tr = O.MockTranscript(label)
        for v in sums:
            tr.append_fr(v)
        r0 = tr.challenge()
        az, bz = tables()
        member = O.Member.gruen_product(az, bz, tau, scale=kernel)
        claim = member.input_claim()
        rounds = len(tau)
        out = O.prove_batch([member], [claim], [self.one], [0], rounds, 3, label=label + 1)
        point = out["challenges"][rounds - self.n_vars:][::-1]
        values = np.stack([O.poly_evaluate(z, point) for z in inputs]) if self.n_vars else np.stack([z[0] for z in inputs])
        return dict(sums=sums, r0=r0, polys=out["polys"], challenges=out["challenges"], final_claim=out["final_claim"], values=values, claim=claim)

    def spartan_outer(self, label):
        d = self.d
        inputs = [O.fr_from_u64(c) for c in d["outer_cols"]]
        n = len(inputs)
        shape = d["outer_iwa"].shape
        wa_f = O.fr_from_i64(d["outer_iwa"].reshape(-1)).reshape(shape + (4,))
        wb_f = O.fr_from_i64(d["outer_iwb"].reshape(-1)).reshape(shape + (4,))
        eq = O.eq_evals(d["outer_tau"])
        return self._spartan(inputs, lambda: O.r1cs_uniskip_sums(inputs, eq, wa_f, wb_f), lambda: O.r1cs_materialize(inputs, d["outer_wa"], d["outer_wb"]),
                             d["outer_tau"], d["outer_kernel"], label)

    def spartan_product(self, label):
        d = self.d
        rows = d["product_rows"]
        two64 = O.to_mont([1 << 64])[0].reshape(1, 4)
        hi = O.fr_from_i64(rows["right_input"][:, 1].copy().view(np.int64))
        right = O.fr_add(O.fr_mul(hi, np.repeat(two64, hi.shape[0], axis=0)), O.fr_from_u64(rows["right_input"][:, 0].copy()))
        lanes = [O.fr_from_u64(rows["left_input"]), O.fr_from_u64(rows["lookup_output"]), O.fr_from_u64(rows["jump"].astype(np.uint64)), right,
                 O.fr_from_u64(rows["branch"].astype(np.uint64)), O.fr_from_u64(rows["next_is_noop"].astype(np.uint64))]
        eq = O.eq_evals(d["product_tau"]) if self.n_vars else O.to_mont([1])
        return self._spartan(lanes, lambda: O.spartan_product_t1(rows, eq), lambda: O.spartan_product_tables(rows, d["product_w"]), d["product_tau"], d["product_kernel"], label)

    def ram_read_write(self, label):
        S, d = self.S, self.d
        ram = d["ram"]
        log_t, log_k = ram["log_t"], ram["log_k"]
        gamma = d["ram_gamma"]
        state = dict(inc=O.fr_from_i64(ram["inc"]), vi=O.fr_from_u64(ram["val_init"]))
        orc = O.RwMatrix(ram["addresses"], ram["pre"], ram["post"])
        eq_state = O.SplitEqState(d["ram_tau"])
        # the input claim from the dense definition: sum_j eq(tau, j) * [access_j] * (pre_j + gamma * post_j)
        acc = ram["addresses"] != S.NO_ACCESS
        pre, post = O.fr_from_u64(np.where(acc, ram["pre"], 0).astype(np.uint64)), O.fr_from_u64(np.where(acc, ram["post"], 0).astype(np.uint64))
        g = np.repeat(np.asarray(gamma).reshape(1, 4), pre.shape[0], axis=0)
        claim = O.Member.expr([O.eq_evals(d["ram_tau"]), O.fr_add(pre, O.fr_mul(g, post))], [(self.one, [0, 1])], 2).input_claim()

        def ingest(rnd_bound, bind):
            if rnd_bound < log_t:
                orc.cycle_bind(bind)
                eq_state.bind(bind)
                state["inc"] = O.bind_low_to_high(state["inc"], bind)
                if rnd_bound == log_t - 1:
                    orc.into_address_major()
            else:
                state["vi"] = orc.address_bind(bind, state["vi"])

        def matrix_round(rnd, bind):
            if bind is not None:
                ingest(rnd - 1, bind)
            if rnd < log_t:
                e_out, e_in, in_bits = eq_state.tables()
                return orc.cycle_round(e_out, e_in, in_bits, state["inc"], gamma), (eq_state.scalar, eq_state.point())
            return orc.address_round(state["vi"], state["inc"], eq_state.scalar.reshape(1, 4), gamma), None

        def final_values():
            ra_f, val_f = orc.final_values(state["vi"])
            return np.stack([ra_f, val_f, state["inc"][0], eq_state.scalar])

        class Tr:
            def __init__(self, label):
                self.t = O.MockTranscript(label)

            def append(self, values):
                for v in np.asarray(values).reshape(-1, 4):
                    self.t.append_fr(v)

            def challenge(self):
                return self.t.challenge()

        S._sub = lambda a, b: O.fr_sub(np.asarray(a).reshape(1, 4), np.asarray(b).reshape(1, 4))[0]
        out = S.rw_rounds(matrix_round, lambda bind: ingest(log_t + log_k - 1, bind), final_values, log_t, log_k, claim, Tr(label), O.gruen_poly_deg_3,
                          O.univariate_from_evals, O.univariate_evaluate)
        out["claim"] = claim
        orc.close()
        return out

    def registers_read_write(self, label):
        from registers_fixture import inc_table
        S, d = self.S, self.d
        reg = d["registers"]
        log_t, log_k = reg["log_t"], reg["log_k"]
        K = 1 << log_k
        gamma, r_cycle = d["registers_gamma"], d["registers_r_cycle"]
        g2 = O.fr_mul(gamma.reshape(1, 4), gamma.reshape(1, 4))[0]
        rep = lambda v, n: np.repeat(np.asarray(v).reshape(1, 4), n, axis=0)
        T = 1 << log_t
        weighted = O.fr_add(O.fr_from_u64(reg["rd_post"]), O.fr_add(O.fr_mul(rep(gamma, T), O.fr_from_u64(reg["rs1_val"])), O.fr_mul(rep(g2, T), O.fr_from_u64(reg["rs2_val"]))))
        claim = O.Member.expr([O.eq_evals(r_cycle), weighted], [(self.one, [0, 1])], 2).input_claim()
        state = dict(inc=inc_table(reg, O), dense=None)
        orc = O.RegMatrix(reg["rs1"], reg["rs1_val"], reg["rs2"], reg["rs2_val"], reg["rd"], reg["rd_pre"], reg["rd_post"], gamma)
        eq_state = O.SplitEqState(r_cycle)

        def ingest(bound, bind):
            if bound < log_t:
                orc.cycle_bind(bind)
                eq_state.bind(bind)
                state["inc"] = O.bind_low_to_high(state["inc"], bind)
                if bound == log_t - 1:
                    state["dense"] = list(orc.into_dense(K))
            else:
                state["dense"] = [O.bind_low_to_high(t, bind) for t in state["dense"]]

        def matrix_round(rnd, bind):
            if bind is not None:
                ingest(rnd - 1, bind)
            if rnd < log_t:
                e_out, e_in, _ = eq_state.tables()
                q = orc.cycle_round(e_out, e_in, state["inc"])
                return np.concatenate([q, np.zeros((2, 4), dtype=np.uint64)]), (eq_state.scalar, eq_state.point())
            return O.regrw_address_round(*state["dense"], state["inc"][0], eq_state.scalar), None

        def final_values():
            ra, wa, val = (t[0] for t in state["dense"])
            return np.stack([val, wa, ra, state["inc"][0], eq_state.scalar])

        orc_tr = O.MockTranscript(label)

        class Tr:
            def append(self, values):
                for v in np.asarray(values).reshape(-1, 4):
                    orc_tr.append_fr(v)

            def challenge(self):
                return orc_tr.challenge()

        S._sub = lambda a, b: O.fr_sub(np.asarray(a).reshape(1, 4), np.asarray(b).reshape(1, 4))[0]
        out = S.rw_rounds(matrix_round, lambda bind: ingest(log_t + log_k - 1, bind), final_values, log_t, log_k, claim, Tr(), O.gruen_poly_deg_3, O.univariate_from_evals,
                          O.univariate_evaluate, four_point_address=True)
        eq_adr, eq_cyc = O.eq_evals(out["challenges"][log_t:][::-1]), O.eq_evals(out["challenges"][:log_t][::-1])
        out["operand_claims"] = np.stack([O.regrw_operand_claim(reg["rs1"], eq_adr, eq_cyc), O.regrw_operand_claim(reg["rs2"], eq_adr, eq_cyc)])
        out["claim"] = claim
        orc.close()
        return out

    # above this many cycles the from-the-definition address rounds (T x 128 x 3 table evaluations) stop being a test-sized computation for ALL rounds ...
    DIRECT_ADDRESS_ROUNDS_MAX_LOG_T = 12

    # ... and a SAMPLE of them is computed from the definition instead (round-4 review, item 1): whole phases 0 and 15 plus round 62 up to T = 2^20 (17 of the 128 rounds: the first
    # phase -- no checkpoints yet --, the last, whose suffixes are empty, and one round in the middle), single rounds at the ends and in the middle above that
    # (the sample is sized so that the GPU suite stays within minutes: a from-the-definition round costs T x 3 table evaluations)
    @classmethod
    def sampled_direct_rounds(cls, n_vars):
        if n_vars <= cls.DIRECT_ADDRESS_ROUNDS_MAX_LOG_T:
            return set(range(128))
        if n_vars <= 20:
            return {8 * p + k for p in (0, 15) for k in range(8)} | {62}
        return {0, 62, 127}

    def instruction_read_raf(self, label):
        """The twin of DeviceExtended.instruction_read_raf.  The T-scale scans are the oracle's (oracle/read_raf.c) at every size and are compared sum for sum.
        The address-round polynomials are the oracle's FROM THE DEFINITION (oracle/lookup_tables.c: evaluate_mle of the row's table at the mixed point, no
        prefix / suffix machinery) for every round up to T = 2^12.  Above that only the rounds of `sampled_direct_rounds` are from the definition -- they are
        asserted equal to what the product's host state machine produces from the ORACLE's scan sums, so at those rounds the address polynomials at trace scale
        are oracle-vs-product, not product-vs-product -- and the remaining rounds' messages are the state machine's (its only inputs are the scan sums, 256 entries
        per polynomial whatever T is; tests/test_read_raf_address_cpu.py pins it to the definition round for round at <= 2^10).  For those the twin acts as the
        sumcheck VERIFIER: its end values must be the oracle's evaluate_mle of every table at r_address (asserted below).  `self.direct_checked` lists the rounds
        that were compared against the definition in the last call."""
        from jolt_amd import ffi
        S, d = self.S, self.d
        lk = d["lookup"]
        lists = ffi.lookup_suffix_lists()
        tr = O.MockTranscript(label)
        u0 = O.eq_evals(d["lookup_reduction"])
        gamma = d["lookup_gamma"]
        claim = O.read_raf_input_claim(lk["idx"], lk["table"], lk["raf"], u0, gamma)  # first principles: materialize_entry and the operands of every row
        input_claim = claim
        all_direct = self.n_vars <= self.DIRECT_ADDRESS_ROUNDS_MAX_LOG_T
        sampled = self.sampled_direct_rounds(self.n_vars)
        direct = O.ReadRafAddressDirect(lk["idx"], lk["table"], lk["raf"], u0, gamma)
        present = np.zeros(S.N_LOOKUP_TABLES, dtype=np.uint8)
        present[lk["present"]] = 1
        state = None if all_direct else ffi.HostReadRafAddress(gamma, present)
        u = u0
        v_tables, scans, messages, challenges = [], [], [], []
        self.direct_checked = []
        for phase in range(S.PHASES):
            suffix_len = S.ADDRESS_BITS - 8 * (phase + 1)
            if phase:
                u = O.read_raf_condense(lk["idx"], u, v_tables[-1], suffix_len + 8)
            raf, suf = O.read_raf_phase_scan(lk["idx"], lk["table"], lk["raf"], lk["n_tables"], u, suffix_len, S.ADDRESS_BITS, lists)
            scans.append((raf, suf))
            if state:
                state.init_phase(phase, raf, suf)
            last_sampled = max([r for r in sampled if r // 8 == phase], default=-1)
            if not all_direct and last_sampled >= 0:
                # the definition's per-row weight eq(r_reduction, j) * eq(r_{<8 phase}, k_j[<8 phase]) IS the condensed u of this phase (both are the oracle's)
                direct.restart(8 * phase, u, np.stack(challenges) if challenges else None)
            phase_challenges = []
            for rnd in range(8):
                g = 8 * phase + rnd
                if all_direct:
                    e = direct.round()
                    assert np.array_equal(O.fr_add(e[0:1], e[1:2])[0], claim), ("s(0) + s(1) != running claim", phase, rnd)
                    self.direct_checked.append(g)
                else:
                    e = state.message(claim)
                    if g in sampled:
                        want = direct.round()
                        assert np.array_equal(np.asarray(e), want), ("the state machine's address-round polynomial differs from the definition's", phase, rnd)
                        assert np.array_equal(O.fr_add(want[0:1], want[1:2])[0], claim), ("s(0) + s(1) != running claim", phase, rnd)
                        self.direct_checked.append(g)
                coeffs = O.univariate_from_evals(e)
                for c in coeffs:
                    tr.append_fr(c)
                r = tr.challenge()
                claim = O.univariate_evaluate(coeffs, r)
                if all_direct or g < last_sampled:
                    direct.bind(r)
                if state:
                    state.bind(r)
                messages.append(coeffs)
                phase_challenges.append(r)
            challenges += phase_challenges
            v_tables.append(O.eq_evals(np.stack(phase_challenges)))
        direct = direct if all_direct else None
        vt = np.stack(v_tables)
        if direct:
            table_values, (left, right, identity, _) = direct.values()
            g2 = O.fr_mul(gamma.reshape(1, 4), gamma.reshape(1, 4))[0]
            raf_interleaved = O.fr_add(O.fr_mul(gamma.reshape(1, 4), left.reshape(1, 4)), O.fr_mul(g2.reshape(1, 4), right.reshape(1, 4)))[0]
            raf_identity = O.fr_mul(g2.reshape(1, 4), identity.reshape(1, 4))[0]
        else:
            table_values, raf_interleaved, raf_identity = state.finish()
            state.close()
            # The sumcheck VERIFIER's view of those 128 rounds, from independent ends: the input claim above is first principles, every round satisfied
            # s(0) + s(1) = claim by construction, and what they must end in is fixed by the oracle's own evaluate_mle of every table (and the operand
            # polynomials) at r_address -- equal values here and an equal cycle-phase sum below leave the state machine's polynomials no room to be wrong.
            want_values, (left, right, identity, _) = O.read_raf_address_values(np.stack(challenges))
            assert np.array_equal(np.asarray(table_values)[lk["present"]], want_values[lk["present"]])
            g2 = O.fr_mul(gamma.reshape(1, 4), gamma.reshape(1, 4))
            assert np.array_equal(raf_interleaved, O.fr_add(O.fr_mul(gamma.reshape(1, 4), left.reshape(1, 4)), O.fr_mul(g2, right.reshape(1, 4)))[0])
            assert np.array_equal(raf_identity, O.fr_mul(g2, identity.reshape(1, 4))[0])
        combined, ra = O.read_raf_cycle_tables(lk["idx"], lk["table"], lk["raf"], table_values, raf_interleaved, raf_identity, vt, S.ADDRESS_BITS, d["ra_count"])
        n_f = 1 + d["ra_count"]
        orc = O.Member.expr([O.eq_evals(d["lookup_reduction"]), combined] + [ra[i] for i in range(d["ra_count"])], [(self.one, list(range(1 + n_f)))], 1 + n_f)
        assert np.array_equal(orc.input_claim(), claim), "the running claim after 128 address rounds is not the sum the cycle rounds start from"
        out = O.prove_batch([orc], [claim], [self.one], [0], self.n_vars, n_f + 1, label=label + 1)
        instruction_ra = orc.final_values()[2:1 + n_f]
        eq_cycle = O.eq_evals(np.asarray(out["challenges"])[::-1])
        flags = O.onehot_pushforward(lk["table"], 64, eq_cycle)
        raf_flag = O.onehot_pushforward(np.where(lk["raf"] != 0, 0, 0xFF).astype(np.uint8), 64, eq_cycle)[0]
        return dict(lookup_table_flags=flags[lk["present"]], instruction_raf_flag=raf_flag, instruction_ra=instruction_ra, scans=scans, address_polys=np.stack(messages), address_challenges=np.stack(challenges), v_tables=vt, table_values=np.asarray(table_values)[lk["present"]],
                    raf_values=np.stack([raf_interleaved, raf_identity]), cycle_claim=claim, polys=out["polys"], challenges=out["challenges"], final_claim=out["final_claim"],
                    claim=input_claim)

    def booleanity_address(self, label):
        S, bo = self.S, self.d["booleanity"]
        K = 1 << bo["log_k"]
        eq = O.eq_evals(bo["reference_cycle"])
        masses = np.stack([O.onehot_pushforward(bo["cols"][i], K, eq) for i in range(bo["cols"].shape[0])])
        orc_tr = O.MockTranscript(label)

        class Tr:
            def append(self, values):
                for v in np.asarray(values).reshape(-1, 4):
                    orc_tr.append_fr(v)

            def challenge(self):
                return orc_tr.challenge()

        out = S.booleanity_address_rounds(O.BooleanityAddress(masses, bo["gamma"], bo["reference_address"]), bo["log_k"], Tr(), O.univariate_from_evals, O.univariate_evaluate)
        out["masses"] = masses
        out["claim"] = np.zeros(4, dtype=np.uint64)
        return out

    def booleanity_cycle(self, label, r_address):
        """the cycle phase FROM THE DEFINITION (crates/jolt-kernels/src/reference/booleanity.rs, the naive tier): a flat Expr member over the dense eq table
        eq(reference_cycle, .) * eq(r_address, reference_address) and the dense columns H_i(j) = gamma^i eq(r_address, hot_i(j)) (0 on a cold cycle), summand
        sum_i (H_i^2 - gamma^i H_i); one-member prove_batch; the output claims are the bound columns divided by gamma^i"""
        bo = self.d["booleanity"]
        cols, K = bo["cols"], 1 << bo["log_k"]
        N, T = cols.shape
        one = O.to_mont([1])[0]
        mul = lambda a, b: O.fr_mul(np.asarray(a).reshape(1, 4), np.asarray(b).reshape(1, 4))[0]
        eq_address = O.eq_evals(np.asarray(r_address).reshape(-1, 4)) if bo["log_k"] else O.to_mont([1])
        scalar = one
        for a, c in zip(np.asarray(r_address).reshape(-1, 4), np.asarray(bo["reference_address"]).reshape(-1, 4)):  # eq_mle(r_address, reference_address)
            ac = mul(a, c)
            term = O.fr_add(O.fr_sub(O.fr_sub(one.reshape(1, 4), a.reshape(1, 4)), c.reshape(1, 4)), O.fr_add(ac.reshape(1, 4), ac.reshape(1, 4)))[0]
            scalar = mul(scalar, term)
        rho, dense = [one], []
        for i in range(N):
            table = np.concatenate([O.fr_mul(eq_address, np.repeat(rho[i].reshape(1, 4), K, axis=0)), np.zeros((1, 4), dtype=np.uint64)])  # slot K: a cold cycle
            idx = np.where(cols[i] == 0xFF, K, cols[i]).astype(np.int64)
            dense.append(np.ascontiguousarray(table[idx]))
            rho.append(mul(rho[i], bo["gamma"]))
        eq = O.eq_evals(bo["reference_cycle"], scalar) if self.n_vars else scalar.reshape(1, 4)
        neg = lambda x: O.fr_neg(np.asarray(x).reshape(1, 4))[0]
        terms = []
        for i in range(N):
            terms.append((one, [0, 1 + i, 1 + i]))
            terms.append((neg(rho[i]), [0, 1 + i]))
        member = O.Member.expr([eq] + dense, terms, 3)
        claim = member.input_claim()
        out = O.prove_batch([member], [claim], [one], [0], self.n_vars, 3, label=label)
        fv = member.final_values()
        inv = lambda x: O.fr_inv(np.asarray(x).reshape(1, 4))[0]
        return dict(polys=out["polys"], challenges=out["challenges"], final_claim=out["final_claim"], claim=claim,
                    ra_claims=np.stack([mul(fv[1 + i], inv(rho[i])) for i in range(N)]), eq_scalar=fv[0])

    def hamming_weight(self, label):
        S, bo, hw = self.S, self.d["booleanity"], self.d["hamming"]
        K = 1 << bo["log_k"]
        eq = O.eq_evals(hw["r_cycle"]) if self.n_vars else O.to_mont([1])
        masses = np.stack([O.onehot_pushforward(bo["cols"][i], K, eq) for i in range(bo["cols"].shape[0])])
        orc_tr = O.MockTranscript(label)

        class Tr:
            def append(self, values):
                for v in np.asarray(values).reshape(-1, 4):
                    orc_tr.append_fr(v)

            def challenge(self):
                return orc_tr.challenge()

        sub = lambda a, b: O.fr_sub(np.asarray(a).reshape(1, 4), np.asarray(b).reshape(1, 4))[0]
        out = S.hamming_weight_rounds(O.HammingWeight(masses, hw["gamma"], hw["r_address"], hw["virtualization_points"]), bo["log_k"], Tr(), O.univariate_from_evals,
                                      O.univariate_evaluate, sub)
        out["masses"] = masses
        return out

    def address_domain(self, label):
        S, d = self.S, self.d
        ram, bc = d["ram"], d["bytecode"]
        ops = OracleOps({"pc": (bc["push_pc"], 1 << bc["log_k"]), "ram": (ram["addresses"], 1 << ram["log_k"]), "ram_post": ram["post"]}, bc["chunk_cols"], 1 << bc["chunk_bits"])
        return {"bytecode_read_raf": S.bytecode_read_raf(ops, bc, self.n_vars, label), "ram_raf_evaluation": S.ram_raf_evaluation(ops, ram, d["ram_raf"], label + 10),
                "ram_output_check": S.ram_output_check(ops, ram, d["ram_output"], label + 20)}

    def prove(self, label=0):
        booleanity_address = self.booleanity_address(label + 450)
        return {"booleanity_cycle": self.booleanity_cycle(label + 460, booleanity_address["challenges"][::-1]),
                **self.address_domain(label + 500), "spartan_outer": self.spartan_outer(label + 100), "spartan_product": self.spartan_product(label + 200), "ram_read_write": self.ram_read_write(label + 300),
                "registers_read_write": self.registers_read_write(label + 350), "instruction_read_raf": self.instruction_read_raf(label + 400),
                "booleanity_address": booleanity_address, "hamming_weight": self.hamming_weight(label + 470)}
