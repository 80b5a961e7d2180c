"""Compact-scalar members (jolt_member_create_lc_small, jolt_amd/csrc/small_round.hip.h): round 0 off the resident u64 columns and the bind_to_field first bind
(crates/jolt-poly/src/dense.rs:129-142; FrSmallScalarAccumulator, crates/jolt-field/src/bn254/mont.rs:343-427) against (i) the same member over promoted tables on
the device and (ii) the oracle's naive flat-Expr member over the promoted tables -- every round polynomial, challenge, final claim, final value, input claim; a second
proof after jolt_member_reset; sizes on both sides of the promote-at-creation threshold.  The catalogue-scale runs are tests/test_gpu_workload.py (T = 2^20, 2^22),
whose witness columns all go through these members."""
import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import ffi
from jolt_amd import workload as W
from util import rand_fr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = ffi.Context(0)
    yield c
    c.close()


def shapes(g):
    """(name, degree, skip_one, eq-weighted, groups over slots; slot kinds 'i' = u64 column, 'f' = field table)"""
    one = O.to_mont([1])[0]
    minus_one = O.fr_neg(one.reshape(1, 4))[0]
    return [
        ("spartan_shift", 2, True, False, "fiiiifi", [[(None, [(one, 0)]), (None, [(one, 1), (g[0], 2), (g[1], 3), (g[2], 4)])], [(None, [(g[3], 5)]), (one, [(minus_one, 6)])]]),
        ("instruction_input", 2, True, True, "iiiiiiii", [[(None, [(one, 0)]), (None, [(one, 1)])], [(None, [(one, 2)]), (None, [(one, 3)])],
                                                          [(None, [(g[0], 4)]), (None, [(one, 5)])], [(None, [(one, 6)]), (None, [(g[0], 7)])]]),
        ("claim_reduction", 1, True, True, "iiiii", [[(None, [(one, 0), (g[0], 1), (g[1], 2), (g[2], 3), (g[3], 4)])]]),
        ("ram_val_check", 3, True, False, "iff", [[(None, [(one, 0)]), (None, [(one, 1)]), (g[0], [(one, 2)])]]),
        ("all_points", 2, False, False, "iif", [[(None, [(one, 0)]), (None, [(one, 1)])], [(None, [(g[1], 2)])]]),
        ("single_column_groups", 3, True, False, "iiif", [[(None, [(g[2], 0)])], [(None, [(one, 1)]), (None, [(one, 2)]), (None, [(one, 3)])]]),
    ]


def corner_column(rng, n):
    col = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    col[rng.random(n) < 0.2] = 0
    col[rng.random(n) < 0.1] = np.uint64(2**64 - 1)
    col[rng.random(n) < 0.2] &= np.uint64(1)  # flags
    return col


@pytest.mark.parametrize("n_vars", [16, 10, 15])
@pytest.mark.parametrize("which", range(6))
def test_small_member_equals_promoted_member_and_oracle(ctx, n_vars, which):
    rng = np.random.default_rng(1000 * n_vars + which)
    gam = rand_fr(4, 50 + which)
    name, degree, skip, eqw, kinds, groups = shapes(gam)[which]
    N = 1 << n_vars
    cols = [corner_column(rng, N) if k == "i" else None for k in kinds]
    fr_tabs = [None if k == "i" else rand_fr(N, 7000 + 13 * i + which) for i, k in enumerate(kinds)]
    ints = [ctx.ints(c) if c is not None else None for c in cols]
    dev_fr = [ctx.upload(t) if t is not None else None for t in fr_tabs]
    promoted = [ctx.table_from_ints(v) if v is not None else None for v in ints]
    slots_small = [ints[i] if kinds[i] == "i" else dev_fr[i] for i in range(len(kinds))]
    slots_dense = [promoted[i] if kinds[i] == "i" else dev_fr[i] for i in range(len(kinds))]
    w = rand_fr(n_vars, 99 + which) if eqw else None
    kw = dict(borrow=True, eq_point=w) if eqw else dict(borrow=True, skip_one=skip)
    small = ctx.member_lc(slots_small, groups, degree, **kw)
    dense = ctx.member_lc(slots_dense, groups, degree, **kw)
    one = O.to_mont([1])[0]
    claim = dense.input_claim()
    assert np.array_equal(small.input_claim(), claim), "input claim off the integer columns"
    msg_degree = degree + 1 if eqw else degree
    outs = []
    for rep in range(2):
        a = ctx.prove_batch([small], [claim], [one], [0], n_vars, msg_degree, label=5 + which)
        b = ctx.prove_batch([dense], [claim], [one], [0], n_vars, msg_degree, label=5 + which)
        for key in ("polys", "challenges", "final_claim"):
            assert np.array_equal(a[key], b[key]), (name, key, rep)
        assert np.array_equal(small.final_values(), dense.final_values()), (name, rep)
        outs.append(a)
        small.reset()
        dense.reset()
    # the oracle's naive member over the promoted tables (the eq weight as one more dense factor of every group)
    host_tabs = [O.fr_from_u64(cols[i]) if kinds[i] == "i" else fr_tabs[i] for i in range(len(kinds))]
    mul = lambda x, y: O.fr_mul(np.asarray(x).reshape(1, 4), np.asarray(y).reshape(1, 4))[0]
    if eqw:
        host_tabs = host_tabs + [O.eq_evals(w)]
        groups_o = [g + [(None, [(one, len(kinds))])] for g in groups]
    else:
        groups_o = groups
    orc = O.Member.expr(host_tabs, W.expand_to_flat_terms(groups_o, mul, one), msg_degree)
    assert np.array_equal(orc.input_claim(), claim)
    want = O.prove_batch([orc], [claim], [one], [0], n_vars, msg_degree, label=5 + which)
    for key in ("polys", "challenges", "final_claim"):
        assert np.array_equal(outs[0][key], want[key]), (name, key, "oracle")
    small.destroy()
    dense.destroy()
    for t in promoted + dev_fr + ints:
        if t is not None:
            t.free()


def test_other_integer_kinds_are_refused(ctx):
    """i64 / i128 columns are promoted by the caller (jolt_table_from_ints): the constructor says JOLT_ERR_UNSUPPORTED instead of misreading them"""
    one = O.to_mont([1])[0]
    col = ctx.ints(np.arange(1 << 15, dtype=np.int64) - 5)
    with pytest.raises(ffi.JoltError):
        ctx.member_lc([col], [[(None, [(one, 0)])]], 1, borrow=True, skip_one=True)
    col.free()
