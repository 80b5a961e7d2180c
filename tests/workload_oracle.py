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
