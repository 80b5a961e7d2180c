"""GPU parity of registers read/write checking (stage 4) through the C ABI: every cycle round's two sums and split-eq state,
the cells after every cycle bind, every address round's four evaluations, the final values and the two operand claims equal the oracle's
restatement of optimized/registers_read_write (oracle/registers_rw.c, itself pinned to the dense reference member in
tests/test_oracle_registers.py) -- the reference's optimized-vs-reference lock step (optimized/parity.rs:79-118)."""
import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import ffi
from jolt_amd import stages as S
from registers_fixture import inc_table
from util import rand_challenge, rand_fr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = ffi.Context(0)
    yield c
    c.close()


def device_member(ctx, tr, inc, r_cycle, gamma):
    regs = ctx.onehot(np.stack([tr["rs1"], tr["rs2"], tr["rd"]]), 1 << tr["log_k"])
    cols = [ctx.ints(tr[k]) for k in ("rs1_val", "rs2_val", "rd_pre", "rd_post")]
    inc_t = ctx.upload(inc)
    m = ctx.registers_rw(regs, *cols, inc_t, r_cycle, gamma)
    inc_t.free()
    return m, regs, cols


def lockstep(ctx, tr, seed, compare_cells=True):
    log_k, log_t = tr["log_k"], tr["log_t"]
    K = 1 << log_k
    r_cycle, gamma = rand_fr(log_t, seed), rand_fr(1, seed + 1)[0]
    inc = inc_table(tr, O)
    dev, regs, cols = device_member(ctx, tr, inc, r_cycle, gamma)
    orc = O.RegMatrix(tr["rs1"], tr["rs1_val"], tr["rs2"], tr["rs2_val"], tr["rd"], tr["rd_pre"], tr["rd_post"], gamma)
    eq_state = O.SplitEqState(r_cycle)
    inc_cur = inc.copy()
    assert len(dev) == len(orc)
    bind, dense, chal = None, None, []

    def ingest(bound, r):
        nonlocal inc_cur, dense
        if bound < log_t:
            orc.cycle_bind(r)
            eq_state.bind(r)
            inc_cur = O.bind_low_to_high(inc_cur, r)
            if bound == log_t - 1:
                dense = list(orc.into_dense(K))
        else:
            dense = [O.bind_low_to_high(t, r) for t in dense]

    for rnd in range(log_t + log_k):
        if bind is not None:
            ingest(rnd - 1, bind)
        evals, aux = dev.prove_round(bind)
        if rnd < log_t:
            e_out, e_in, _ = eq_state.tables()
            want = orc.cycle_round(e_out, e_in, inc_cur)
            assert np.array_equal(evals[:2], want), f"round {rnd}"
            assert np.array_equal(aux[0], eq_state.scalar) and np.array_equal(aux[1], eq_state.point()), rnd
            if compare_cells or rnd in (0, 1, log_t // 2):
                d, o = dev.download(), orc.export()
                assert len(d["rows"]) == len(o["rows"]), rnd
                for k in ("rows", "cols", "val", "ra", "wa", "prev", "next"):
                    assert np.array_equal(d[k], o[k]), (rnd, k)
        else:
            assert np.array_equal(evals, O.regrw_address_round(*dense, inc_cur[0], eq_state.scalar)), f"round {rnd}"
        bind = rand_challenge(seed + 10 + rnd, shifted=(rnd % 3 != 2))
        chal.append(bind)
    dev.finish(bind)
    ingest(log_t + log_k - 1, bind)
    fin = dev.final_values()
    assert np.array_equal(fin[0], dense[2][0]) and np.array_equal(fin[1], dense[1][0]) and np.array_equal(fin[2], dense[0][0])
    assert np.array_equal(fin[3], inc_cur[0]) and np.array_equal(fin[4], eq_state.scalar)
    # operand claims: one-hot evaluations at the bound point (mod.rs:296-372), through the existing one-hot operators
    r_cyc_pt = np.stack(chal[:log_t][::-1])
    eq_adr_host = O.eq_evals(np.stack(chal[log_t:][::-1])) if log_k else O.to_mont([1])
    eq_adr = ctx.upload(eq_adr_host)
    eq_cyc = O.eq_evals(r_cyc_pt)
    for p, name in ((0, "rs1"), (1, "rs2")):
        col = regs.materialize(p, eq_adr)
        assert np.array_equal(ctx.evaluate(col, r_cyc_pt), O.regrw_operand_claim(tr[name], eq_adr_host, eq_cyc)), name
        col.free()
    eq_adr.free()
    dev.free()
    regs.free()
    for c in cols:
        c.free()
    orc.close()


@pytest.mark.parametrize("log_k,log_t,hot,probs", [(3, 4, None, (0.8, 0.6, 0.7)), (7, 6, None, (0.8, 0.6, 0.7)), (2, 5, 2, (1.0, 1.0, 1.0)), (4, 3, None, (0.2, 0.1, 0.3)),
                                                    (7, 9, None, (0.9, 0.8, 0.8)), (5, 7, 3, (0.9, 0.9, 0.9)), (3, 5, None, (0.0, 0.0, 0.0)), (1, 1, None, (1.0, 1.0, 1.0))])
def test_registers_lockstep_with_oracle(ctx, log_k, log_t, hot, probs):
    rng = np.random.default_rng(500 + 7 * log_k + log_t)
    lockstep(ctx, S.consistent_register_trace(log_k, log_t, rng, *probs, hot=hot), 800 + log_t)


def test_registers_at_trace_scale(ctx):
    """T = 2^16 cycles over the 128 registers (~2.1 cells per cycle, thousands of cells per row pair group in the late cycle rounds): round sums
    and final values against the oracle, the cell arrays compared at three checkpoints"""
    rng = np.random.default_rng(77)
    lockstep(ctx, S.consistent_register_trace(7, 16, rng), 4343, compare_cells=False)


@pytest.mark.parametrize("log_t", [10, 16])
def test_registers_on_a_hot_set_of_registers(ctx, log_t):
    """90 % of the operands on 8 of the 128 registers (a compiled loop lives in a handful of registers; jolt_amd.stages.hotset_addresses): write chains tens of thousands
    of cycles long per hot register, up to three cells of one cycle in the same few columns"""
    rng = np.random.default_rng(6100 + log_t)
    lockstep(ctx, S.consistent_register_trace(7, log_t, rng, addresses="hotset"), 6200 + log_t, compare_cells=(log_t <= 10))


def test_registers_argument_checks(ctx):
    rng = np.random.default_rng(5)
    tr = S.consistent_register_trace(3, 4, rng)
    inc = inc_table(tr, O)
    two = ctx.onehot(np.stack([tr["rs1"], tr["rs2"]]), 8)
    cols = [ctx.ints(tr[k]) for k in ("rs1_val", "rs2_val", "rd_pre", "rd_post")]
    with pytest.raises(ffi.JoltError) as e:
        ctx.registers_rw(two, *cols, ctx.upload(inc), rand_fr(4, 1), rand_fr(1, 2)[0])  # three index columns are required
    assert e.value.status == 1
    regs = ctx.onehot(np.stack([tr["rs1"], tr["rs2"], tr["rd"]]), 8)
    with pytest.raises(ffi.JoltError) as e:
        ctx.registers_rw(regs, *cols, ctx.upload(inc[:8]), rand_fr(4, 1), rand_fr(1, 2)[0])  # inc length != cycles
    assert e.value.status == 5
    m = ctx.registers_rw(regs, *cols, ctx.upload(inc), rand_fr(4, 1), rand_fr(1, 2)[0])
    with pytest.raises(ffi.JoltError) as e:
        m.final_values()
    assert e.value.status == 7  # NotFullyBound
    m.free()
