"""A synthetic RAM trace for the read/write-checking tests (TEST INFRASTRUCTURE): per cycle an optional access (address, value
before, value after) consistent with an initial memory -- the columns of RamAccessColumns (crates/jolt-kernels/src/optimized/
ram_trace.rs:22-75) -- and the dense (K x T) grids the reference member sums over (reference/ram_read_write.rs:29-69)."""
import numpy as np

NO_ACCESS = np.uint64(0xFFFFFFFFFFFFFFFF)


def make_trace(log_k, log_t, seed, access=0.7, write=0.5, hot=None):
    rng = np.random.default_rng(seed)
    K, T = 1 << log_k, 1 << log_t
    val_init = rng.integers(0, 2**64, size=K, dtype=np.uint64)
    mem = val_init.copy()
    addresses = np.full(T, NO_ACCESS, dtype=np.uint64)
    pre, post = np.zeros(T, dtype=np.uint64), np.zeros(T, dtype=np.uint64)
    inc = np.zeros(T, dtype=object)  # post - pre as a signed integer (RamInc)
    pool = K if hot is None else min(K, hot)
    for j in range(T):
        if rng.random() >= access:
            continue
        a = int(rng.integers(0, pool))
        addresses[j] = a
        pre[j] = mem[a]
        if rng.random() < write:
            mem[a] = rng.integers(0, 2**64, dtype=np.uint64)
        post[j] = mem[a]
        inc[j] = int(post[j]) - int(pre[j])
    return dict(log_k=log_k, log_t=log_t, val_init=val_init, addresses=addresses, pre=pre, post=post, inc=inc)


def dense_grids(tr, O):
    """ra(k, j) and val(k, j) over index k * T + j as Montgomery tables, plus inc(j) and val_init(k)"""
    K, T = 1 << tr["log_k"], 1 << tr["log_t"]
    ra = np.zeros((K, T), dtype=np.uint64)
    val = np.zeros((K, T), dtype=np.uint64)
    mem = tr["val_init"].copy()
    for j in range(T):
        val[:, j] = mem
        a = tr["addresses"][j]
        if a != NO_ACCESS:
            ra[int(a), j] = 1
            mem[int(a)] = tr["post"][j]
    inc = O.to_mont([int(v) % O.R_MOD for v in tr["inc"]])
    return O.fr_from_u64(ra.reshape(-1)), O.fr_from_u64(val.reshape(-1)), inc, O.fr_from_u64(tr["val_init"])
