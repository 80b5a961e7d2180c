"""jolt_key_index_* (jolt_amd/csrc/key_index.hip) against oracle/address_ops.c: pushforwards of cycle weights onto address domains from 2 to 2^17 entries
(one and several sort passes), skewed keys (one key holding most rows: many work items per bin), cold cycles, several weight tables at once, the final-memory
column; argument checks."""
import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import ffi
from jolt_amd.workload import rand_fr

pytestmark = pytest.mark.gpu
NONE = np.uint64(0xFFFFFFFFFFFFFFFF)


@pytest.fixture(scope="module")
def ctx():
    c = ffi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("log_t,k,n_w,skew,cold", [(1, 2, 1, 0.0, 0.0), (10, 7, 3, 0.0, 0.2), (12, 300, 5, 0.6, 0.1), (14, 32768, 2, 0.3, 0.0), (15, 32769, 2, 0.0, 0.3),
                                                   (16, 1 << 16, 8, 0.5, 0.1), (17, (1 << 17) - 5, 1, 0.9, 0.0), (13, 1, 2, 0.0, 0.5)])
def test_pushforward_matches_fold_cycles(ctx, log_t, k, n_w, skew, cold):
    rng = np.random.default_rng(log_t * 1000 + n_w)
    T = 1 << log_t
    keys = rng.integers(0, k, size=T).astype(np.uint64)
    keys[rng.random(T) < skew] = np.uint64(k // 3)
    keys[rng.random(T) < cold] = NONE
    if cold:
        keys[0] = np.uint64(k)  # an in-range-looking but cold key (>= k)
    weights = [rand_fr(T, rng) for _ in range(n_w)]
    kd = ctx.ints(keys)
    ix = ctx.key_index(kd, k)
    cycles, kk, items = ix.size()
    assert cycles == T and kk == k and items >= 1
    tabs = [ctx.upload(w) for w in weights]
    out = ix.pushforward(tabs)
    for s in range(n_w):
        assert np.array_equal(out[s].download(), O.fold_cycles(keys, k, weights[s])), s
    again = ix.pushforward(tabs[:1])  # the index serves any number of pushforwards
    assert np.array_equal(again[0].download(), out[0].download())
    post = rng.integers(0, 2**64, size=T, dtype=np.uint64)
    init = rand_fr(k, rng)
    pd, it = ctx.ints(post), ctx.upload(init)
    last = ix.last_value(pd, it)
    assert np.array_equal(last.download(), O.last_value(keys, post, k, init))
    for t in tabs + out + again + [it, last]:
        t.free()
    ix.free()
    kd.free()
    pd.free()


@pytest.mark.parametrize("log_t,k", [(14, 1 << 12), (18, 1 << 16), (20, 1 << 16)])
def test_pushforward_and_last_value_on_a_hot_set_key_stream(ctx, log_t, k):
    """Keys drawn like a btreemap's RAM addresses (90 % on <= 2^10 of the k keys, jolt_amd.stages.hotset_addresses): bins of thousands of rows next to empty bins -- the
    sorted index, its work items and both walks against the oracle's fold_cycles / last_value"""
    from jolt_amd.stages import hotset_addresses
    rng = np.random.default_rng(9000 + log_t)
    T = 1 << log_t
    keys = hotset_addresses(k, T, rng)
    keys[rng.random(T) < 0.4] = NONE  # the RAM column's cold cycles
    weights = [rand_fr(T, rng) for _ in range(2)]
    kd = ctx.ints(keys)
    ix = ctx.key_index(kd, k)
    tabs = [ctx.upload(w) for w in weights]
    out = ix.pushforward(tabs)
    for s in range(2):
        assert np.array_equal(out[s].download(), O.fold_cycles(keys, k, weights[s])), s
    post = rng.integers(0, 2**64, size=T, dtype=np.uint64)
    init = rand_fr(k, rng)
    pd, it = ctx.ints(post), ctx.upload(init)
    last = ix.last_value(pd, it)
    assert np.array_equal(last.download(), O.last_value(keys, post, k, init))
    for t in tabs + out + [it, last]:
        t.free()
    ix.free()
    kd.free()
    pd.free()


def test_argument_checks(ctx):
    keys = ctx.ints(np.arange(8, dtype=np.uint64))
    with pytest.raises(ffi.JoltError):
        ctx.key_index(keys, 0)
    with pytest.raises(ffi.JoltError):
        ctx.key_index(keys, (1 << 24) + 1)
    ix = ctx.key_index(keys, 8)
    short = ctx.upload(rand_fr(4, np.random.default_rng(0)))
    with pytest.raises(ffi.JoltError):
        ix.pushforward([short])
    with pytest.raises(ffi.JoltError):
        ix.pushforward([ctx.upload(rand_fr(8, np.random.default_rng(0)))] * 9)
    signed = ctx.ints(np.arange(8, dtype=np.int64))
    with pytest.raises(ffi.JoltError):
        ctx.key_index(signed, 8)
    ix.free()
