"""Dense (K x T) grids of the registers read/write-checking relation for the oracle pins (TEST INFRASTRUCTURE): rs1_ra / rs2_ra / rd_wa one-hot
grids, the register-file value grid val(k, j) (the value BEFORE cycle j's write) and RdInc, register-major (index k * T + j) as the
reference member takes them (crates/jolt-kernels/src/reference/registers_read_write.rs:3-10)."""
import numpy as np

from jolt_amd.stages import REG_NONE


def inc_table(tr, O):
    """RdInc(j) = rd_post - rd_pre as field elements (a signed 65-bit difference)"""
    lo = (tr["rd_post"] - tr["rd_pre"]).astype(np.uint64)  # wraps mod 2^64
    borrow = tr["rd_post"] < tr["rd_pre"]
    two64 = O.to_mont([1 << 64])[0]
    corr = np.zeros((lo.shape[0], 4), dtype=np.uint64)
    corr[borrow] = two64
    return O.fr_sub(O.fr_from_u64(lo), corr)


def dense_grids(tr, O):
    K, T = 1 << tr["log_k"], 1 << tr["log_t"]
    rs1, rs2, wa = (np.zeros((K, T), dtype=np.uint64) for _ in range(3))
    val = np.zeros((K, T), dtype=np.uint64)
    regs = np.zeros(K, dtype=np.uint64)
    for j in range(T):
        val[:, j] = regs
        if tr["rs1"][j] != REG_NONE:
            rs1[int(tr["rs1"][j]), j] = 1
            assert tr["rs1_val"][j] == regs[int(tr["rs1"][j])]
        if tr["rs2"][j] != REG_NONE:
            rs2[int(tr["rs2"][j]), j] = 1
            assert tr["rs2_val"][j] == regs[int(tr["rs2"][j])]
        if tr["rd"][j] != REG_NONE:
            wa[int(tr["rd"][j]), j] = 1
            assert tr["rd_pre"][j] == regs[int(tr["rd"][j])]
            regs[int(tr["rd"][j])] = tr["rd_post"][j]
    f = lambda g: O.fr_from_u64(g.reshape(-1))
    return f(rs1), f(rs2), f(wa), f(val)
