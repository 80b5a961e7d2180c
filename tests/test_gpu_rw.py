"""GPU parity of the sparse read-write matrix (RAM read/write checking) through the C ABI: every round's sums and,
after every bind, the whole entry array equal the oracle's restatement of CycleMajorMatrix / AddressMajorMatrix
(oracle/rw_matrix.c, itself pinned against the dense reference member in tests/test_oracle_rw.py) -- the reference's own
optimized-vs-reference lock step (crates/jolt-kernels/src/optimized/parity.rs:79-118)."""
import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import ffi
from rw_fixture import make_trace
from util import rand_challenge, rand_fr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = ffi.Context(0)
    yield c
    c.close()


def same_entries(dev, orc):
    d, o = dev.download(), orc.export()
    assert len(d["rows"]) == len(o["rows"])
    for k in ("rows", "cols", "val", "ra", "prev", "next"):
        assert np.array_equal(d[k], o[k]), k


def lockstep(ctx, tr, seed, compare_entries=True):
    log_k, log_t = tr["log_k"], tr["log_t"]
    inc = O.to_mont([int(v) % O.R_MOD for v in tr["inc"]])
    val_init = O.fr_from_u64(tr["val_init"])
    tau, gamma = rand_fr(log_t, seed), rand_fr(1, seed + 1)[0]
    dev = ctx.rw_matrix(tr["addresses"], tr["pre"], tr["post"], ctx.upload(inc), ctx.upload(val_init), tau, gamma)
    orc = O.RwMatrix(tr["addresses"], tr["pre"], tr["post"])
    eq_state = O.SplitEqState(tau)
    inc_cur, vi_cur = inc.copy(), val_init.copy()
    assert len(dev) == len(orc)
    bind = None
    for rnd in range(log_t + log_k):
        if bind is not None:
            if rnd - 1 < log_t:
                orc.cycle_bind(bind)
                eq_state.bind(bind)
                inc_cur = O.bind_low_to_high(inc_cur, bind)
                if rnd - 1 == log_t - 1:
                    orc.into_address_major()
            else:
                vi_cur = orc.address_bind(bind, vi_cur)
        evals, aux = dev.prove_round(bind)
        if rnd < log_t:
            e_out, e_in, in_bits = eq_state.tables()
            want = orc.cycle_round(e_out, e_in, in_bits, inc_cur, gamma)
            assert np.array_equal(aux[0], eq_state.scalar) and np.array_equal(aux[1], eq_state.point()), rnd
        else:
            want = orc.address_round(vi_cur, inc_cur, eq_state.scalar.reshape(1, 4), gamma)
        assert np.array_equal(evals, want), f"round {rnd}"
        if compare_entries:
            same_entries(dev, orc)
        bind = rand_challenge(seed + 10 + rnd, shifted=(rnd % 3 != 2))  # both challenge shapes
    dev.finish(bind)
    if log_k:
        vi_cur = orc.address_bind(bind, vi_cur)
    else:
        orc.cycle_bind(bind)
        eq_state.bind(bind)
        inc_cur = O.bind_low_to_high(inc_cur, bind)
        orc.into_address_major()
    ra_f, val_f = orc.final_values(vi_cur)
    fin = dev.final_values()
    assert np.array_equal(fin[0], ra_f) and np.array_equal(fin[1], val_f)
    assert np.array_equal(fin[2], inc_cur[0]) and np.array_equal(fin[3], eq_state.scalar)
    dev.free()
    orc.close()


@pytest.mark.parametrize("log_k,log_t,access,hot", [(3, 4, 0.7, None), (4, 6, 1.0, None), (2, 5, 0.3, None), (5, 5, 0.9, 3), (6, 9, 0.6, None), (10, 8, 1.0, None),
                                                    (3, 3, 0.0, None), (1, 1, 1.0, None), (0, 4, 0.8, None)])
def test_sparse_matrix_lockstep_with_oracle(ctx, log_k, log_t, access, hot):
    lockstep(ctx, make_trace(log_k, log_t, 300 + log_k * 7 + log_t, access=access, hot=hot), 700 + log_t)


def test_sparse_matrix_at_trace_scale(ctx):
    """T = 2^16 cycles over 2^14 addresses (many blocks in every scan, groups of hundreds of columns in the late cycle rounds): round
    sums and final values against the oracle; the entry arrays are compared on the way at three checkpoints inside lockstep()."""
    lockstep(ctx, make_trace(14, 16, 42, access=0.45, write=0.6), 4242, compare_entries=False)


@pytest.mark.parametrize("log_k,log_t", [(10, 12), (14, 16), (16, 17)])
def test_sparse_matrix_on_a_hot_set_address_stream(ctx, log_k, log_t):
    """A btreemap-like trace (BASELINE configs[4]; jolt_amd.stages.hotset_addresses): 90 % of the accesses on <= 2^10 words -- access chains thousands of cycles long per
    hot word, column groups of thousands of cells in the late cycle rounds, most of the K words never touched -- where the uniform traces above spread two or three
    accesses over every word.  The merge-rank scatter of the bind, the match scans and the address rounds against the oracle, round for round."""
    from jolt_amd import stages as S
    tr = S.consistent_ram_trace(log_k, log_t, np.random.default_rng(5000 + log_t), access=0.6, write=0.5, addresses="hotset")
    hit = tr["addresses"][tr["addresses"] != S.NO_ACCESS]
    assert np.unique(hit, return_counts=True)[1].max() > (len(hit) >> 11)  # the stream IS skewed: some word takes hundreds of times its uniform share
    lockstep(ctx, tr, 7100 + log_t, compare_entries=(log_t <= 12))


def test_rw_matrix_argument_checks(ctx):
    tr = make_trace(3, 4, 5)
    inc, vi = ctx.upload(O.to_mont([int(v) % O.R_MOD for v in tr["inc"]])), ctx.upload(O.fr_from_u64(tr["val_init"]))
    tau, gamma = rand_fr(4, 1), rand_fr(1, 2)[0]
    bad = tr["addresses"].copy()
    bad[0] = 8  # outside K = 8 (RamAccessColumns::validate_addresses)
    with pytest.raises(ffi.JoltError) as e:
        ctx.rw_matrix(bad, tr["pre"], tr["post"], inc, vi, tau, gamma)
    assert e.value.status == 1
    with pytest.raises(ffi.JoltError) as e:
        ctx.rw_matrix(tr["addresses"][:8], tr["pre"][:8], tr["post"][:8], inc, vi, tau[:3], gamma)  # inc length != cycles
    assert e.value.status == 5
    m = ctx.rw_matrix(tr["addresses"], tr["pre"], tr["post"], inc, vi, tau, gamma)
    with pytest.raises(ffi.JoltError) as e:
        m.final_values()
    assert e.value.status == 7  # NotFullyBound
    m.free()
