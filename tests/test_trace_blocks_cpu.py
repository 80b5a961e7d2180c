"""The trace description of the stage operators, dealt in blocks (jolt_amd/stages.py: extended_params / extended_block / build_extended(n, n_blocks = G)): what the
ranks of a sharded prover hold is ONE execution -- block g starts from the memory and the register file block g - 1 left.  The generators are vectorised numpy
(sorted chains, searchsorted); here the assembled trace is replayed cycle by cycle on a plain sequential machine, for 1, 2 and 4 blocks, and the per-block descriptions a
rank builds for itself (only_state = the cheap pass that carries the state forward) are compared with the slices of the assembled one."""
import numpy as np
import pytest

from jolt_amd import stages as S

KW = dict(n_tables=4, log_k=5)


def replay_ram(ram):
    mem = ram["val_init"].copy()
    for j in range(ram["addresses"].shape[0]):
        a = ram["addresses"][j]
        if a == S.NO_ACCESS:
            assert ram["pre"][j] == 0 and ram["post"][j] == 0 and ram["inc"][j] == 0, j
            continue
        assert ram["pre"][j] == mem[a], ("a read returns the last word written", j)
        assert ram["inc"][j] == np.int64(ram["post"][j]) - np.int64(ram["pre"][j]), j
        mem[a] = ram["post"][j]
    assert np.array_equal(mem, ram["val_final"])


def replay_registers(reg):
    K = 1 << reg["log_k"]
    regs = np.zeros(K, dtype=np.uint64)
    for j in range(reg["rd"].shape[0]):
        for col, val in (("rs1", "rs1_val"), ("rs2", "rs2_val"), ("rd", "rd_pre")):
            r = reg[col][j]
            assert reg[val][j] == (0 if r == S.REG_NONE else regs[r]), (col, j)
        if reg["rd"][j] != S.REG_NONE:
            regs[reg["rd"][j]] = reg["rd_post"][j]
    return regs


@pytest.mark.parametrize("n_vars,n_blocks", [(6, 1), (6, 2), (7, 4), (5, 4)])
def test_a_trace_dealt_in_blocks_is_one_execution(n_vars, n_blocks):
    d = S.build_extended(n_vars, seed=31 + n_blocks, n_blocks=n_blocks, **KW)
    assert d["n_vars"] == n_vars and d["ram"]["addresses"].shape[0] == 1 << n_vars
    replay_ram(d["ram"])
    replay_registers(d["registers"])
    # the lookup rows and the integer columns of the R1CS inputs cover every cycle exactly once
    assert d["lookup"]["idx"].shape[0] == 2 << n_vars or d["lookup"]["idx"].shape[0] == 1 << n_vars
    assert all(c.shape[0] == 1 << n_vars for c in d["outer_cols"])


@pytest.mark.parametrize("n_blocks", [2, 4])
def test_a_rank_rebuilds_its_own_block_from_the_carried_state(n_blocks):
    """what ShardedExtended does on rank g: walk blocks 0 .. g - 1 with only_state (memory and registers carried forward, nothing else built), then build block g --
    the same columns as the slice [g T_b, (g + 1) T_b) of the trace assembled from all blocks"""
    n_vars, seed = 7, 58
    n_block = n_vars - (n_blocks.bit_length() - 1)
    whole = S.build_extended(n_vars, seed=seed, n_blocks=n_blocks, **KW)
    p = S.extended_params(n_vars, seed, **KW)
    ram_state, reg_state = None, None
    Tb = 1 << n_block
    for g in range(n_blocks):
        blk = S.extended_block(p, n_block, g, seed, ram_state, reg_state, only_state=False)
        for key in ("addresses", "pre", "post", "inc"):
            assert np.array_equal(blk["ram"][key], whole["ram"][key][g * Tb:(g + 1) * Tb]), (g, key)
        for key in ("rs1", "rs2", "rd", "rs1_val", "rs2_val", "rd_pre", "rd_post"):
            assert np.array_equal(blk["registers"][key], whole["registers"][key][g * Tb:(g + 1) * Tb]), (g, key)
        cheap = S.extended_block(p, n_block, g, seed, ram_state, reg_state, only_state=True)
        assert np.array_equal(cheap["ram"]["val_final"], blk["ram"]["val_final"]) and np.array_equal(cheap["registers"]["reg_final"], blk["registers"]["reg_final"])
        ram_state, reg_state = blk["ram"]["val_final"], blk["registers"]["reg_final"]
    assert np.array_equal(ram_state, whole["ram"]["val_final"])


@pytest.mark.parametrize("n_vars,n_blocks", [(9, 1), (10, 4)])
def test_a_hot_set_trace_is_one_consistent_execution_and_is_skewed(n_vars, n_blocks):
    """ram_addresses="hotset" (BASELINE configs[4]'s btreemap shape, jolt_amd.stages.hotset_addresses): 90 % of the RAM accesses on a hot set of words and of the register
    operands on 8 registers -- still ONE consistent execution (every read returns the last write), whole or dealt in blocks, and measurably skewed"""
    kw = dict(KW)
    kw["log_k"] = 8
    d = S.build_extended(n_vars, seed=77, n_blocks=n_blocks, ram_addresses="hotset", **kw)
    replay_ram(d["ram"])
    replay_registers(d["registers"])
    hit = d["ram"]["addresses"][d["ram"]["addresses"] != S.NO_ACCESS]
    counts = np.sort(np.unique(hit, return_counts=True)[1])[::-1]
    hot = max(1, (1 << 8) // 64)
    assert counts[:hot].sum() > 0.8 * len(hit)  # the hot set takes ~ 90 % + its uniform share
    rd = d["registers"]["rd"][d["registers"]["rd"] != S.REG_NONE]
    assert np.sort(np.unique(rd, return_counts=True)[1])[::-1][:8].sum() > 0.8 * len(rd)
    uniform = S.build_extended(n_vars, seed=77, n_blocks=n_blocks, **kw)
    assert not np.array_equal(uniform["ram"]["addresses"], d["ram"]["addresses"])
    with pytest.raises(ValueError):
        S.consistent_ram_trace(4, 4, np.random.default_rng(0), addresses="zipf")
