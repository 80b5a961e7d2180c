"""The T-scale operators of the stages that are not plain cycle-domain relations, driven the way the prover's
stage drivers drive them, over synthetic trace-shaped inputs:

  stage 1   Spartan outer            uni-skip sums off the integer witness columns, Az / Bz of the remainder member, its log T + 1 rounds,
                                     the claimed inputs at the bind point        (crates/jolt-kernels/src/optimized/spartan_outer.rs)
  stage 2   Spartan product          the same three steps over the six product lanes, no stream variable  (optimized/spartan_product.rs)
  stage 2   RAM read / write         the sparse (address x cycle) matrix: log T cycle rounds + log K address rounds
                                     (optimized/ram_read_write.rs:58-330, optimized/rw_matrix.rs)
  stage 4   registers read / write   the same sparse matrix with <= 3 cells per cycle and two coefficient columns: log T cycle rounds on the device,
                                     log K = 7 address rounds over K-sized arrays, the two operand claims as one-hot evaluations
                                     (optimized/registers_read_write/{mod,sparse,rows}.rs)
  stage 5   instruction read + RAF   the whole kernel over the 42 real lookup tables: per address phase the T-scale scans on the device (condensation, RAF
                                     sums, per-table suffix accumulators) and the phase's 8 rounds over 256-entry prefix / suffix polynomials on the host
                                     (jolt_host_read_raf_address_*), then the log T cycle rounds over combined * prod ra on the device, their claim
                                     being the running claim the 128 address rounds leave  (optimized/instruction_read_raf.rs)

  stage 6a  booleanity, address      the pushforward masses G_i[k] = sum_j eq(r_cycle, j) [hot_i(j) = k] of all RA columns (T-scale, device), then the
                                     log K rounds over K-entry tables with the squared-weight bind on the host, where the reference keeps them
                                     (optimized/booleanity.rs:152-427)

  stage 7   Hamming-weight reduction the pushforward masses of all RA columns against the shared eq(r_cycle) (T-scale, device), then log K rounds of sum_i G_i W_i on the host
                                     (optimized/hamming_weight_claim_reduction.rs)
  stage 2   RAM RAF evaluation       ra_folded(k) = sum_{j: address(j) = k} eq(tau_low, j) over the K RAM words (T-scale, device: a key index over the address
                                     column), then the log K rounds of ra_folded * unmap over K-sized tables  (optimized/ram_raf_evaluation.rs)
  stage 2   RAM output check         val_final(k) = the word of the last access to k (T-scale), then the log K rounds of eq(r_address) * io_mask * (val_final - val_io)
                                     (optimized/ram_output_check.rs)
  stage 6a  bytecode read+RAF, addr  the five per-stage pushforwards F_s(k) = sum_{j: pc(j) = k} eq(r_cycle_s, j) onto the bytecode domain in one pass over the
                                     PC index (T-scale), then log K rounds over 13 K-sized tables  (optimized/bytecode_read_raf.rs:152-437)
  stage 6b  bytecode read+RAF, cycle ra_i(j) = eq(chunk_i)[chunk_i(pc_j)] and the combined coefficient column C(j) (T-scale), log T rounds of C * prod ra_i
                                     (optimized/bytecode_read_raf.rs:440-690)

`build_extended` is a pure description (numpy); `DeviceExtended` holds the resident inputs in HBM and proves; tests/workload_oracle.py
instantiates the same description on the CPU oracle.  Every operator absorbs what it sends into a transcript and takes its challenges
from it (the deterministic test transcript, jolt_host_transcript_* / jolt_host_prove_batch), so two runs agree message for message.
"""
import numpy as np

from .workload import rand_fr

NO_ACCESS = np.uint64(0xFFFFFFFFFFFFFFFF)
ADDRESS_BITS, PHASES, CHUNK = 128, 16, 256
# integer Lagrange basis of the 3-node window {-1, 0, 1} at the extended nodes {-2 .. 2} (spartan_product.rs:86-105 extension_coefficients)
PRODUCT_EXTENSION = np.array([[3, -3, 1], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, -3, 3]], dtype=np.int64)


HOTSET_SHARE = 0.9  # share of the accesses that fall on the hot set (addresses="hotset")


def draw_hot_set(K, rng, hot=None):
    """the hot words of a skewed address stream: `hot` addresses scattered over [0, K) (default min(2^10, max(1, K / 64)))"""
    hot = max(1, min(K, hot if hot is not None else min(1 << 10, max(1, K // 64))))
    return rng.permutation(K)[:hot].astype(np.uint64)


def hotset_addresses(K, T, rng, hot=None, hot_set=None):
    """A skewed (btreemap-like) address stream: HOTSET_SHARE of the accesses on a hot set of addresses scattered over [0, K) (the tree's upper levels and the allocator's
    free lists are touched by every operation, specs/byte-addressable-memory.md:119,127-130; crates/jolt-prover/src/profile.rs:80-84), the rest uniform over K.
    hot_set: the hot addresses themselves (a trace dealt in blocks shares ONE hot set: extended_params draws it), else draw_hot_set(K, rng, hot).  Deterministic in `rng`."""
    if hot_set is None:
        hot_set = draw_hot_set(K, rng, hot)
    hot_set = np.asarray(hot_set, dtype=np.uint64)
    a = rng.integers(0, K, size=T, dtype=np.uint64)
    on_hot = rng.random(T) < HOTSET_SHARE
    a[on_hot] = hot_set[rng.integers(0, len(hot_set), size=int(on_hot.sum()))]
    return a


def consistent_ram_trace(log_k, log_t, rng, access=0.6, write=0.5, val_init=None, addresses="uniform", hot_set=None):
    """RamAccessColumns (optimized/ram_trace.rs:22-75) of a memory-consistent synthetic trace: per cycle an optional access (address, word before,
    word after), the word before being what the previous access to that address left (or the initial memory).  val_init: the memory the trace starts from
    (a later block of a longer trace: the final memory of the block before it, `val_final`); drawn when None.  addresses: "uniform" over the K words, or "hotset"
    (hotset_addresses: 90 % of the accesses on ~2^10 words -- chains of thousands of accesses per hot word, most words never touched)."""
    K, T = 1 << log_k, 1 << log_t
    if val_init is None:
        val_init = rng.integers(0, 2**63, size=K, dtype=np.uint64)
    val_init = np.ascontiguousarray(val_init, dtype=np.uint64)
    if addresses not in ("uniform", "hotset"):
        raise ValueError(addresses)
    addresses = rng.integers(0, K, size=T, dtype=np.uint64) if addresses == "uniform" else hotset_addresses(K, T, rng, hot_set=hot_set)
    hit = rng.random(T) < access
    addresses[~hit] = NO_ACCESS
    cyc = np.nonzero(hit)[0]
    a = addresses[cyc].astype(np.int64)
    order = np.lexsort((cyc, a))  # by address, then by cycle: every address's accesses form one contiguous chain
    a_s = a[order]
    n = a_s.shape[0]
    pre, post = np.zeros(T, dtype=np.uint64), np.zeros(T, dtype=np.uint64)
    if n:
        writes = rng.random(n) < write
        fresh = rng.integers(0, 2**63, size=n, dtype=np.uint64)
        pos = np.arange(n)
        chain_start = np.r_[True, a_s[1:] != a_s[:-1]]
        start_pos = np.maximum.accumulate(np.where(chain_start, pos, 0))
        last_write = np.maximum.accumulate(np.where(writes, pos, -1))
        post_s = np.where(last_write >= start_pos, fresh[np.maximum(last_write, 0)], val_init[a_s])  # the last write of the chain so far, else the initial word
        pre_s = np.where(chain_start, val_init[a_s], np.r_[post_s[:1], post_s[:-1]])
        pre[cyc[order]] = pre_s
        post[cyc[order]] = post_s
    val_final = val_init.copy()
    if n:
        chain_end = np.r_[a_s[1:] != a_s[:-1], True]
        val_final[a_s[chain_end]] = post_s[chain_end]
    inc = post.astype(np.int64) - pre.astype(np.int64)  # RamInc: words < 2^63, so the difference fits an i64
    return dict(log_k=log_k, log_t=log_t, val_init=val_init, val_final=val_final, addresses=addresses, pre=pre, post=post, inc=inc)


REG_NONE = np.uint8(0xFF)


def consistent_register_trace(log_k, log_t, rng, p_rs1=0.8, p_rs2=0.6, p_rd=0.7, hot=None, reg_init=None, addresses="uniform", hot_set=None):
    """RegisterCycleRow columns (optimized/registers_read_write/rows.rs:22-31) of a consistent synthetic trace: a read returns what the last
    earlier write to that register left (registers start at 0, or at reg_init for a later block of a longer trace), rd_pre likewise; rd_post is fresh.
    hot: draw registers from the first `hot` only (many cells per register pair).  `reg_final`: the register file after the last cycle."""
    K, T = 1 << log_k, 1 << log_t
    reg_init = np.zeros(K, dtype=np.uint64) if reg_init is None else np.ascontiguousarray(reg_init, dtype=np.uint64)
    pool = K if hot is None else min(K, hot)
    if addresses == "hotset":  # 90 % of the operands on 8 of the K registers (a compiled loop lives in a handful of registers), the rest uniform
        regs_hot = hot_set if hot_set is not None else draw_hot_set(K, rng, min(8, K))
        draw = lambda p: np.where(rng.random(T) < p, hotset_addresses(K, T, rng, hot_set=regs_hot), 0xFF).astype(np.uint8)
    else:
        draw = lambda p: np.where(rng.random(T) < p, rng.integers(0, pool, size=T), 0xFF).astype(np.uint8)
    rs1, rs2, rd = draw(p_rs1), draw(p_rs2), draw(p_rd)
    rd_post = np.where(rd != REG_NONE, rng.integers(0, 2**64, size=T, dtype=np.uint64), 0).astype(np.uint64)
    cyc = np.arange(T, dtype=np.int64)
    w = np.nonzero(rd != REG_NONE)[0]
    wkey = rd[w].astype(np.int64) * T + w  # writes sorted by (register, cycle)
    order = np.argsort(wkey, kind="stable")
    wkey_s, wpost_s, wreg_s = wkey[order], rd_post[w][order], rd[w][order]

    def value_before(reg):  # the register's value at the start of every cycle that names it (0xFF rows: 0)
        key = reg.astype(np.int64) * T + cyc
        pos = np.searchsorted(wkey_s, key, side="left") - 1  # the last write with (register, cycle) < (reg, j)
        ok = (reg != REG_NONE) & (pos >= 0)
        posc = np.maximum(pos, 0)
        same = ok & (wreg_s[posc] == reg) if wkey_s.size else np.zeros(T, dtype=bool)
        start = np.where(reg != REG_NONE, reg_init[np.minimum(reg, K - 1)], 0)  # no earlier write in this block: what the register held when the block began
        return np.where(same, wpost_s[posc] if wkey_s.size else 0, start).astype(np.uint64)

    rs1_val, rs2_val, rd_pre = value_before(rs1), value_before(rs2), value_before(rd)
    reg_final = reg_init.copy()
    if wkey_s.size:
        last = np.r_[wreg_s[1:] != wreg_s[:-1], True]
        reg_final[wreg_s[last]] = wpost_s[last]
    # RdInc as a field table needs post - pre; values are full 64-bit words, so the signed difference is kept as (magnitude, sign)
    return dict(log_k=log_k, log_t=log_t, rs1=rs1, rs1_val=rs1_val, rs2=rs2, rs2_val=rs2_val, rd=rd, rd_pre=rd_pre, rd_post=rd_post, reg_final=reg_final)


N_LOOKUP_TABLES = 42
BITMASK_TABLES = np.array([25, 26, 27, 28], dtype=np.uint8)  # VirtualSRL, VirtualSRA, VirtualROTR, VirtualROTRW


def _spread_bits(v):
    """the 32 low bits of each u64 moved to the even bit positions"""
    v = v & np.uint64(0xFFFFFFFF)
    v = (v | (v << np.uint64(16))) & np.uint64(0x0000FFFF0000FFFF)
    v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF00FF00FF)
    v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
    v = (v | (v << np.uint64(2))) & np.uint64(0x3333333333333333)
    v = (v | (v << np.uint64(1))) & np.uint64(0x5555555555555555)
    return v


def interleave_operands(x, y):
    """interleave_bits (crates/jolt-lookup-tables/src/interleave.rs:14-33): x on the odd bit positions, y on the even ones; -> (n, 2) u64 (lo, hi)"""
    x, y = np.asarray(x, dtype=np.uint64), np.asarray(y, dtype=np.uint64)
    lo = (_spread_bits(x) << np.uint64(1)) | _spread_bits(y)
    hi = (_spread_bits(x >> np.uint64(32)) << np.uint64(1)) | _spread_bits(y >> np.uint64(32))
    return np.stack([lo, hi], axis=1)


def extended_params(n_vars, seed=2026, n_outer=35, n_nodes=9, n_tables=42, ra_count=4, log_k=None, log_kb=None, ram_addresses="uniform"):
    """What a proof over T = 2^n_vars cycles shares between the blocks of its trace (and between the ranks of a sharded prover): the points, the batching
    scalars, the integer / field column weights, the sizes, the K-sized public tables -- everything of the description that is not a witness column."""
    rng = np.random.default_rng([seed + 500, 0xA11])
    p = {"n_vars": n_vars, "n_outer": n_outer, "n_nodes": n_nodes, "ram_addresses": ram_addresses}  # ram_addresses: "uniform" | "hotset" (RAM words AND registers)
    # ---- stage 1: integer uni-skip column weights (~40 % non-zero), field weights of the remainder
    shape = (n_nodes, 2, 1 + n_outer)
    p["outer_iwa"] = rng.integers(-2**20, 2**20, size=shape).astype(np.int64) * (rng.random(shape) < 0.4)
    p["outer_iwb"] = rng.integers(-2**40, 2**40, size=shape).astype(np.int64) * (rng.random(shape) < 0.4)
    p["outer_tau"] = rand_fr(n_vars + 1, rng)
    p["outer_kernel"] = rand_fr(1, rng)[0]
    p["outer_wa"] = rand_fr(2 * (1 + n_outer), rng).reshape(2, 1 + n_outer, 4)
    p["outer_wb"] = rand_fr(2 * (1 + n_outer), rng).reshape(2, 1 + n_outer, 4)
    p["product_tau"] = rand_fr(n_vars, rng)
    p["product_kernel"] = rand_fr(1, rng)[0]
    p["product_w"] = rand_fr(3, rng)
    p["ram_log_k"] = min(16, max(1, n_vars)) if log_k is None else log_k
    p["ram_tau"] = rand_fr(n_vars, rng)
    p["ram_gamma"] = rand_fr(1, rng)[0]
    p["registers_r_cycle"] = rand_fr(n_vars, rng)
    p["registers_gamma"] = rand_fr(1, rng)[0]
    p["lookup_present"] = np.sort(rng.permutation(N_LOOKUP_TABLES)[: min(n_tables, N_LOOKUP_TABLES)]).astype(np.uint8)
    p["lookup_gamma"] = rand_fr(1, rng)[0]
    p["lookup_reduction"] = rand_fr(n_vars, rng)
    p["ra_count"] = ra_count
    n_ra, log_kc = (36, 4) if n_vars >= 4 else (5, 2)
    p["n_ra"], p["log_kc"] = n_ra, log_kc
    p["booleanity"] = dict(log_k=log_kc, reference_cycle=rand_fr(n_vars, rng), reference_address=rand_fr(log_kc, rng), gamma=rand_fr(1, rng)[0])
    p["hamming"] = dict(r_cycle=rand_fr(n_vars, rng), r_address=rand_fr(log_kc, rng), virtualization_points=rand_fr(n_ra * log_kc, rng).reshape(n_ra, log_kc, 4), gamma=rand_fr(1, rng)[0])
    K_ram = 1 << p["ram_log_k"]
    io_lo, io_len = K_ram // 4, max(1, K_ram // 16)
    val_io = np.zeros(K_ram, dtype=np.uint64)
    val_io[io_lo:io_lo + io_len] = rng.integers(0, 2**63, size=io_len, dtype=np.uint64)
    if ram_addresses == "hotset":  # ONE hot set per trace, whatever the number of blocks it is dealt in (its own generator: the other draws stay what they were)
        hs = np.random.default_rng([seed + 500, 0x407])
        p["ram_hot_set"], p["reg_hot_set"] = draw_hot_set(K_ram, hs), draw_hot_set(1 << 7, hs, 8)
    p["ram_raf"] = dict(tau_low=rand_fr(n_vars, rng), lowest_address=np.uint64(0x80000000))
    p["ram_output"] = dict(point=rand_fr(p["ram_log_k"], rng), io_lo=io_lo, io_len=io_len, val_io=val_io)
    log_kb = log_kb if log_kb is not None else min(12, max(2, n_vars))
    Kb = 1 << log_kb
    p["bytecode"] = dict(log_k=log_kb, chunk_bits=4, stage_points=np.stack([rand_fr(n_vars, rng) for _ in range(5)]) if n_vars else np.zeros((5, 0, 4), dtype=np.uint64),
                         stage_values=rand_fr(5 * Kb, rng).reshape(5, Kb, 4), gamma=rand_fr(1, rng)[0], entry_index=int(rng.integers(0, Kb)))
    return p


def extended_block(p, n_block, block, seed=2026, ram_init=None, reg_init=None, only_state=False):
    """The witness columns of block `block` of the trace: T_b = 2^n_block cycles, drawn from the block's own generators, so that any rank of a sharded prover
    builds ITS block without the others' columns.  What ties blocks together is the machine state: the RAM and the register file a block starts from are what the
    block before it left (ram_init / reg_init: its `ram["val_final"]` / `registers["reg_final"]`; block 0 draws its memory, registers start at 0) -- a rank replays
    the RAM / register generators of the blocks before its own (only_state=True: just those two)."""
    T = 1 << n_block
    gen = lambda part: np.random.default_rng([seed + 500, 0xB10C, block, part])
    b = {}
    b["ram"] = consistent_ram_trace(p["ram_log_k"], n_block, gen(1), val_init=ram_init, addresses=p.get("ram_addresses", "uniform"), hot_set=p.get("ram_hot_set"))
    b["registers"] = consistent_register_trace(7, n_block, gen(2), reg_init=reg_init, addresses=p.get("ram_addresses", "uniform"), hot_set=p.get("reg_hot_set"))  # REGISTER_ADDRESS_BITS = 7
    if only_state:
        return b
    rng = gen(3)
    n_outer = p["n_outer"]
    # ---- stage 1: flags and registers of the R1CS inputs
    b["outer_cols"] = [rng.integers(0, 2, size=T, dtype=np.uint64) if v % 3 else rng.integers(0, 2**64, size=T, dtype=np.uint64) for v in range(n_outer)]
    # ---- stage 2: product lanes (SpartanProductRow, spartan_product.rs:62-84); right_instruction_input is an i128 (lo, hi two's complement)
    right_lo = rng.integers(0, 2**64, size=T, dtype=np.uint64)
    right_hi = rng.integers(-2**62, 2**62, size=T, dtype=np.int64).view(np.uint64)
    b["product_rows"] = {
        "left_input": rng.integers(0, 2**64, size=T, dtype=np.uint64), "lookup_output": rng.integers(0, 2**64, size=T, dtype=np.uint64),
        "jump": rng.integers(0, 2, size=T).astype(np.uint8), "right_input": np.stack([right_lo, right_hi], axis=1),
        "branch": rng.integers(0, 2, size=T).astype(np.uint8), "next_is_noop": rng.integers(0, 2, size=T).astype(np.uint8)}
    # ---- stage 5: lookup rows over real tables (LookupTableKind ids)
    rng = gen(4)
    present = p["lookup_present"]
    table = present[rng.integers(0, len(present), size=T)]
    idx = np.frombuffer(rng.bytes(16 * T), dtype=np.uint64).reshape(T, 2).copy()
    shapes = rng.integers(0, 8, size=T)
    idx[shapes == 0] = 0
    idx[shapes == 1, 1] = 0
    idx[shapes == 2] = np.uint64(0xFFFFFFFFFFFFFFFF)
    idx[shapes == 3, 0] &= np.uint64(0xFF)
    # the shift / rotate tables are defined on mask-shaped right operands 1..10..0 (tables/mod.rs:283-290, test_utils.rs:31-38)
    masked = np.isin(table, BITMASK_TABLES)
    if masked.any():
        n = int(masked.sum())
        x = rng.integers(0, 2**64, size=n, dtype=np.uint64)
        zeros = rng.integers(0, 65, size=n)
        y = np.where(zeros >= 64, np.uint64(0), np.uint64(0xFFFFFFFFFFFFFFFF) << np.minimum(zeros, 63).astype(np.uint64))
        idx[masked] = interleave_operands(x, y)
    table = table.copy()
    table[rng.random(T) < 0.1] = 0xFF
    b["lookup"] = dict(idx=idx, table=table, raf=(rng.random(T) < 0.3).astype(np.uint8))
    # ---- stage 6a: the RA selector columns of the booleanity check (instruction, bytecode, RAM chunks: 36 columns at log_k_chunk = 4; RAM cold 40 %)
    rng = gen(5)
    n_ra, log_kc = p["n_ra"], p["log_kc"]
    cols = rng.integers(0, 1 << log_kc, size=(n_ra, T)).astype(np.uint8)
    for q in range(max(1, n_ra // 12)):
        cols[n_ra - 1 - q, rng.random(T) < 0.4] = 0xFF
    b["bool_cols"] = cols
    # ---- stage 6a / 6b: a program-shaped PC column: the trace runs loops of 16 .. 512 instructions (a loop body holds most of the cycles), with a tail of unmapped
    # padding cycles (push_pc 0 in the address phase, cold in the cycle phase)
    rng = gen(6)
    bc = p["bytecode"]
    Kb, pcs, j = 1 << bc["log_k"], np.zeros(T, dtype=np.uint64), 0
    while j < T:
        body = int(rng.integers(min(16, Kb), min(512, Kb) + 1))
        start = int(rng.integers(0, Kb - body + 1))
        run = min(T - j, body * int(rng.integers(1, 200)))
        pcs[j:j + run] = start + (np.arange(run) % body)
        j += run
    mapped = rng.random(T) >= 0.01
    mapped[T - T // 32:] = False
    mapped[0] = True
    chunk_bits = bc["chunk_bits"]
    n_chunks = (bc["log_k"] + chunk_bits - 1) // chunk_bits
    b["bytecode"] = dict(push_pc=np.where(mapped, pcs, 0).astype(np.uint64), mapped=mapped,
                         chunk_cols=np.stack([np.where(mapped, (pcs >> np.uint64((n_chunks - 1 - i) * chunk_bits)) & np.uint64(15), 0xFF).astype(np.uint8) for i in range(n_chunks)]))
    return b


def assemble_description(p, blocks):
    """The description DeviceExtended / OracleExtended take, over the concatenation of `blocks` (one block: a plain single-process trace)."""
    cat = lambda f: np.concatenate([f(b) for b in blocks]) if len(blocks) > 1 else f(blocks[0])
    cat1 = lambda f: np.concatenate([f(b) for b in blocks], axis=1) if len(blocks) > 1 else f(blocks[0])
    d = {k: p[k] for k in ("n_vars", "outer_iwa", "outer_iwb", "outer_tau", "outer_kernel", "outer_wa", "outer_wb", "product_tau", "product_kernel", "product_w", "ram_tau", "ram_gamma",
                           "registers_r_cycle", "registers_gamma", "lookup_gamma", "lookup_reduction", "ra_count", "hamming", "ram_raf", "ram_output")}
    d["outer_cols"] = [cat(lambda b, v=v: b["outer_cols"][v]) for v in range(p["n_outer"])]
    d["product_rows"] = {k: cat(lambda b, k=k: b["product_rows"][k]) for k in blocks[0]["product_rows"]}
    n_block = blocks[0]["ram"]["log_t"]
    log_t = n_block + (len(blocks).bit_length() - 1)
    d["ram"] = dict(log_k=p["ram_log_k"], log_t=log_t, val_init=blocks[0]["ram"]["val_init"], val_final=blocks[-1]["ram"]["val_final"],
                    **{k: cat(lambda b, k=k: b["ram"][k]) for k in ("addresses", "pre", "post", "inc")})
    d["registers"] = dict(log_k=7, log_t=log_t, **{k: cat(lambda b, k=k: b["registers"][k]) for k in ("rs1", "rs1_val", "rs2", "rs2_val", "rd", "rd_pre", "rd_post")})
    d["lookup"] = dict(n_tables=N_LOOKUP_TABLES, present=p["lookup_present"], **{k: cat(lambda b, k=k: b["lookup"][k]) for k in ("idx", "table", "raf")})
    d["booleanity"] = dict(cols=cat1(lambda b: b["bool_cols"]), **p["booleanity"])
    d["bytecode"] = dict(push_pc=cat(lambda b: b["bytecode"]["push_pc"]), mapped=cat(lambda b: b["bytecode"]["mapped"]), chunk_cols=cat1(lambda b: b["bytecode"]["chunk_cols"]),
                         **p["bytecode"])
    return d


def build_blocks(p, n_block, n_blocks, seed=2026):
    blocks, ram, reg = [], None, None
    for g in range(n_blocks):
        b = extended_block(p, n_block, g, seed, ram_init=ram, reg_init=reg)
        ram, reg = b["ram"]["val_final"], b["registers"]["reg_final"]
        blocks.append(b)
    return blocks


def build_extended(n_vars, seed=2026, n_blocks=1, **kw):
    """The description of a trace of T = 2^n_vars cycles as the concatenation of n_blocks (a power of two) blocks: n_blocks = 1 is the single-process trace,
    n_blocks = G the trace a G-rank sharded prover proves (rank g holds block g; jolt_amd/stages_sharded.py) -- the global oracle twin of that run takes this."""
    log_b = n_blocks.bit_length() - 1
    assert (1 << log_b) == n_blocks and log_b <= n_vars
    p = extended_params(n_vars, seed, **kw)
    return assemble_description(p, build_blocks(p, n_vars - log_b, n_blocks, seed))


def committed_address_chunks(r_address, chunk_bits):
    """geometry::dimensions::committed_address_chunks (crates/jolt-claims/src/protocols/jolt/geometry/dimensions.rs:350-366): the big-endian address point, zero-padded
    at the FRONT to a multiple of chunk_bits, cut into chunk points"""
    r = np.asarray(r_address, dtype=np.uint64).reshape(-1, 4)
    pad = (-r.shape[0]) % chunk_bits
    padded = np.concatenate([np.zeros((pad, 4), dtype=np.uint64), r])
    return [padded[i:i + chunk_bits] for i in range(0, padded.shape[0], chunk_bits)]


def bytecode_read_raf(ops, bc, n_vars, label):
    """Stage 6a then 6b of the bytecode read+RAF check over the adapter `ops` (the device or the oracle): AddressKernel (optimized/bytecode_read_raf.rs:238-437) as the
    reference tier's dense member over the 13 address tables -- the optimized kernel's fused group_evals (:371-392) is field-identical to it, which is its own parity
    statement -- and CycleKernel (:440-690) as C(j) * prod_i ra_i(j)."""
    log_k, K = bc["log_k"], 1 << bc["log_k"]
    gp = [ops.one]
    for _ in range(7):
        gp.append(ops.mul(gp[-1], bc["gamma"]))
    # ---- 6a: pushforwards (T-scale) + log K rounds
    eqs = [ops.eq(p) for p in bc["stage_points"]]
    F = ops.pushforward("pc", eqs)
    V = [ops.upload(bc["stage_values"][s]) for s in range(5)]
    hot = np.zeros(K, dtype=np.uint64)
    entry_trace, entry_expected = hot.copy(), hot.copy()
    entry_trace[int(bc["first_pc"]) if "first_pc" in bc else int(bc["push_pc"][0])] = 1  # the PC of the trace's first cycle (a sharded prover's rank > 0 is told it)
    entry_expected[bc["entry_index"]] = 1
    tables = F + V + [ops.u64_table(np.arange(K, dtype=np.uint64)), ops.u64_table(entry_trace), ops.u64_table(entry_expected)]
    terms = [(gp[s], [s, 5 + s]) for s in range(5)] + [(gp[5], [0, 10]), (gp[6], [2, 10]), (gp[7], [11, 12])]  # stage_weights * raf_weights: g^0 g^5, g^2 g^4 (:303-311)
    member = ops.member_expr(tables, terms, 2)
    claim_a = member.input_claim()
    adr = ops.prove(member, claim_a, log_k, 2, label)
    fin = ops.final_values(member)  # bound F_0..4, V_0..4, Int, entry_trace, entry_expected
    ops.destroy(member)
    intermediate = ops.mul(gp[7], ops.mul(fin[11], fin[12]))
    for s in range(5):
        raf = gp[5] if s == 0 else (gp[4] if s == 2 else None)
        val = fin[5 + s] if raf is None else ops.add(fin[5 + s], ops.mul(raf, fin[10]))
        intermediate = ops.add(intermediate, ops.mul(gp[s], ops.mul(fin[s], val)))
    r_address = adr["challenges"][::-1]  # LowToHigh rounds: the last challenge is the most significant address bit
    # ---- 6b: ra columns and the combined coefficient column (T-scale) + log T rounds
    chunks = committed_address_chunks(r_address, bc["chunk_bits"])
    ra = [ops.materialize_chunk(i, ops.host_eq(chunks[i])) for i in range(len(chunks))]
    int_r = fin[10]  # IdentityPolynomial(r_address): the bound Int table
    weights = [ops.mul(gp[s], fin[5 + s]) for s in range(5)]
    weights[0] = ops.add(weights[0], ops.mul(gp[5], int_r))
    weights[2] = ops.add(weights[2], ops.mul(gp[6], int_r))
    entry_scalar = ops.host_eq(r_address)[bc["entry_index"]]
    spike = ops.eq(np.zeros((n_vars, 4), dtype=np.uint64))  # eq(0, j) = [j = 0]
    combined = ops.rlc(eqs + [spike], weights + [ops.mul(gp[7], entry_scalar)])
    for t in eqs + [spike]:
        ops.free(t)
    n_f = 1 + len(ra)
    # the one T-sized member of the relation: a sharded prover proves it over the ranks' blocks of cycles (cycle_product), everyone else as a plain dense member
    cyc, claim_c, fin_c = ops.cycle_product([combined] + ra, n_vars, label + 1)
    ra_claims = fin_c[1:]
    return dict(address=adr, claim_address=claim_a, intermediate=intermediate, val_stages=np.stack(fin[5:10]), r_address=r_address, cycle=cyc, claim_cycle=claim_c,
                ra_claims=np.stack(ra_claims))


def ram_raf_evaluation(ops, ram, raf, label):
    """optimized/ram_raf_evaluation.rs:17-62: ra_folded = fold_cycles(eq(tau_low)) over the RAM address column, unmap(k) = 8 k + lowest_address, the reference tier's dense
    member over the two K-sized tables, log K rounds"""
    K = 1 << ram["log_k"]
    eq = ops.eq(raf["tau_low"])
    folded = ops.pushforward("ram", [eq])[0]
    ops.free(eq)
    unmap = ops.u64_table(np.uint64(8) * np.arange(K, dtype=np.uint64) + raf["lowest_address"])
    member = ops.member_expr([folded, unmap], [(ops.one, [0, 1])], 2)
    claim = member.input_claim()
    out = ops.prove(member, claim, ram["log_k"], 2, label)
    out["claim"] = claim
    out["ra_claim"] = ops.final_values(member)[0]
    ops.destroy(member)
    return out


def ram_output_check(ops, ram, io, label):
    """optimized/ram_output_check.rs:50-215: eq(r_address, k) * io_mask(k) * (val_final(k) - val_io(k)) with the eq factor split (Gruen); val_final is the word every
    address holds after its last access (the witness oracle's ram_val_final column, built here from the resident access columns)"""
    K = 1 << ram["log_k"]
    init = ops.u64_table(ram["val_init"])
    val_final = ops.last_value("ram", init)
    ops.free(init)
    val_io = ops.u64_table(io["val_io"])
    minus_one = ops.sub(np.zeros(4, dtype=np.uint64), ops.one)
    diff = ops.rlc([val_final, val_io], [ops.one, minus_one])
    ops.free(val_io)
    mask = np.zeros(K, dtype=np.uint64)
    mask[io["io_lo"]:io["io_lo"] + io["io_len"]] = 1
    member = ops.member_gruen_product(ops.u64_table(mask), diff, io["point"])
    claim = member.input_claim()
    out = ops.prove(member, claim, ram["log_k"], 3, label)
    out["claim"] = claim
    point = out["challenges"][::-1]
    out["val_final_claim"] = ops.evaluate(val_final, point)
    ops.free(val_final)
    ops.destroy(member)
    return out


def booleanity_address_rounds(kernel, log_k, transcript, from_evals, evaluate):
    """ProveRounds of OptimizedBooleanityAddressKernel driven alone (optimized/booleanity.rs:344-403): four sampled points per round, every message
    absorbed coefficient by coefficient, the input claim is zero.  `kernel`: round() -> 4 evals, bind(r), intermediate()."""
    polys, chal, claim = [], [], np.zeros(4, dtype=np.uint64)
    for rnd in range(log_k):
        poly = from_evals(kernel.round())
        transcript.append(poly)
        r = transcript.challenge()
        claim = evaluate(poly, r)
        kernel.bind(r)
        polys.append(poly)
        chal.append(r)
    return dict(polys=polys, challenges=np.stack(chal), final_claim=claim, intermediate=kernel.intermediate())


def hamming_weight_rounds(kernel, log_k, transcript, from_evals, evaluate, sub):
    """ProveRounds of HammingWeightKernel driven alone (optimized/hamming_weight_claim_reduction.rs:268-298): s(0) and s(2) per round, s(1) recovered from the running
    claim (round_poly_from_skipped_evals), every message absorbed.  `kernel`: round() -> (s(0), s(2), plain sum), bind(r), output_claims()."""
    polys, chal, claim, first = [], [], None, None
    for rnd in range(log_k):
        s0, s2, total = kernel.round()
        if claim is None:
            claim = first = total
        poly = from_evals(np.stack([s0, sub(claim, s0), s2]))
        transcript.append(poly)
        r = transcript.challenge()
        claim = evaluate(poly, r)
        kernel.bind(r)
        polys.append(poly)
        chal.append(r)
    if claim is None:
        claim = first = kernel.round()[2]
    return dict(polys=polys, challenges=np.stack(chal) if chal else np.zeros((0, 4), dtype=np.uint64), claim=first, final_claim=claim, g_claims=kernel.output_claims())


def product_integer_weights():
    """A (left) / B (right) integer column weights [node][1][1 + 6] of the product lanes in the generic column form (lane order:
    left_input, lookup_output, jump, right_input, branch, next_is_noop; right lane 2 is 1 - next_is_noop)"""
    nodes = PRODUCT_EXTENSION.shape[0]
    a, b = np.zeros((nodes, 1, 7), dtype=np.int64), np.zeros((nodes, 1, 7), dtype=np.int64)
    for p in range(nodes):
        c0, c1, c2 = (int(x) for x in PRODUCT_EXTENSION[p])
        a[p, 0, 1:4] = [c0, c1, c2]
        b[p, 0, 0] = c2
        b[p, 0, 4:7] = [c0, c1, -c2]
    return a, b


def product_field_weights(w, neg):
    a, b = np.zeros((1, 7, 4), dtype=np.uint64), np.zeros((1, 7, 4), dtype=np.uint64)
    a[0, 1], a[0, 2], a[0, 3] = w[0], w[1], w[2]
    b[0, 0], b[0, 4], b[0, 5], b[0, 6] = w[2], w[0], w[1], neg(w[2])
    return a, b


def rw_rounds(matrix_round, finish, final_values, log_t, log_k, claim, transcript, gruen_deg_3, from_evals, evaluate, four_point_address=False):
    """ProveRounds of RamReadWriteKernel driven alone (ram_read_write.rs:160-217): cycle rounds complete the cubic with gruen_poly_deg_3 from the
    two sums and the split-eq state the member reports, address rounds interpolate (s(0), claim - s(0), s(2)); every message is absorbed
    coefficient by coefficient, the next bind is the transcript's challenge.  The same loop runs over the device member and the oracle's."""
    polys, chal, bind = [], [], None
    for rnd in range(log_t + log_k):
        evals, aux = matrix_round(rnd, bind)
        if rnd < log_t:
            poly = gruen_deg_3(aux[0], aux[1], evals[0], evals[1], claim)
        elif four_point_address:  # registers: every point sampled, UnivariatePoly::from_evals (registers_read_write/mod.rs:217-252)
            poly = from_evals(np.stack([evals[k] for k in range(4)]))
        else:
            poly = from_evals(np.stack([evals[0], _sub(claim, evals[0]), evals[1]]))
        transcript.append(poly)
        bind = transcript.challenge()
        claim = evaluate(poly, bind)
        polys.append(poly)
        chal.append(bind)
    finish(bind)
    return dict(polys=polys, challenges=np.stack(chal), final_claim=claim, final_values=final_values())


_sub = None  # field subtraction of the side that runs rw_rounds (set by DeviceExtended / OracleExtended: ffi.host_fr_sub / the oracle's)


class DeviceOps:
    """The adapter the address-domain drivers above run over on the device (tests/workload_oracle.py: OracleOps is its twin on the CPU oracle)."""

    def __init__(self, ctx, ffi, indexes, chunk_source):
        self.ctx, self.ffi, self.indexes, self.chunk_source = ctx, ffi, indexes, chunk_source
        self.one = ffi.host_fr_from_u64(1)
        self.mul, self.add, self.sub, self.host_eq = ffi.host_fr_mul, ffi.host_fr_add, ffi.host_fr_sub, ffi.host_eq_evals

    def eq(self, point):
        return self.ctx.eq_evals(point)

    def upload(self, values):
        return self.ctx.upload(values)

    def u64_table(self, values):
        return self.ctx.from_u64(np.ascontiguousarray(values, dtype=np.uint64))

    def pushforward(self, which, tables):
        return self.indexes[which].pushforward(tables)

    def last_value(self, which, init):
        index, post = self.indexes[which], self.indexes[which + "_post"]
        return index.last_value(post, init)

    def materialize_chunk(self, i, eq_chunk):
        scale = self.ctx.upload(eq_chunk)
        col = self.chunk_source.materialize(i, scale)
        scale.free()
        return col

    def rlc(self, tables, scalars):
        return self.ctx.rlc(tables, np.stack(scalars))

    def member_expr(self, tables, terms, degree):
        return self.ctx.member_expr(tables, terms, degree)

    def member_gruen_product(self, a, b, w):
        return self.ctx.member_split_eq_product(a, b, w)

    def cycle_product(self, tables, n_vars, label):
        """sum_j prod_i tables[i](j) over the cycle domain: (transcript, input claim, final values)"""
        n_f = len(tables)
        member = self.member_expr(tables, [(self.one, list(range(n_f)))], n_f)
        claim = member.input_claim()
        out = self.prove(member, claim, n_vars, n_f, label)
        fin = self.final_values(member)
        self.destroy(member)
        return out, claim, fin

    def prove(self, member, claim, n_vars, degree, label):
        out = self.ctx.prove_batch([member], [claim], [self.one], [0], n_vars, degree, label=label)
        return dict(polys=out["polys"], challenges=out["challenges"], final_claim=out["final_claim"])

    def final_values(self, member):
        return list(member.final_values())

    def evaluate(self, table, point):
        return self.ctx.evaluate(table, point)

    def destroy(self, member):
        member.destroy()

    def free(self, table):
        table.free()


class DeviceExtended:
    def __init__(self, ctx, n_vars, seed=2026, description=None, **kw):
        from . import ffi
        import threading
        self._home_ctx, self._tls = ctx, threading.local()  # `self.ctx`: the calling thread's context (bind_context), the home context otherwise
        self.ffi, self.n_vars = ffi, n_vars
        self.d = d = description if description is not None else build_extended(n_vars, seed, **kw)  # description: a prebuilt build_extended(n_vars, ...)
        self.one = ffi.host_fr_from_u64(1)
        zero = np.zeros(4, dtype=np.uint64)
        # ---- resident inputs: integer witness columns, RAM access columns, lookup rows
        self.outer_ints = [ctx.ints(c) for c in d["outer_cols"]]
        r = d["product_rows"]
        self.product_ints = [ctx.ints(r["left_input"]), ctx.ints(r["lookup_output"]), ctx.ints(r["jump"].astype(np.uint64)), ctx.ints(r["right_input"], "i128"),
                             ctx.ints(r["branch"].astype(np.uint64)), ctx.ints(r["next_is_noop"].astype(np.uint64))]
        self.product_ia, self.product_ib = product_integer_weights()
        self.product_fa, self.product_fb = product_field_weights(d["product_w"], lambda x: ffi.host_fr_sub(zero, x))
        ram = d["ram"]
        self.ram_inc, self.ram_val_init = ctx.ints(ram["inc"]), ctx.ints(ram["val_init"])
        self.ram_cols = [ctx.ints(ram[k]) for k in ("addresses", "pre", "post")]  # RamAccessColumns, uploaded once per trace
        reg = d["registers"]
        self.reg_idx = ctx.onehot(np.stack([reg["rs1"], reg["rs2"], reg["rd"]]), 1 << reg["log_k"])  # RegisterCycleRow index columns
        self.reg_cols = [ctx.ints(reg[k]) for k in ("rs1_val", "rs2_val", "rd_pre", "rd_post")]
        lo = (reg["rd_post"] - reg["rd_pre"]).astype(np.uint64)  # RdInc = post - pre as an i128 (lo, hi two's complement)
        hi = np.where(reg["rd_post"] < reg["rd_pre"], np.uint64(0xFFFFFFFFFFFFFFFF), np.uint64(0)).astype(np.uint64)
        self.reg_inc = ctx.ints(np.stack([lo, hi], axis=1), "i128")
        bo = d["booleanity"]
        self.bool_cols = ctx.onehot(bo["cols"], 1 << bo["log_k"])
        lk = d["lookup"]
        self.read_raf = ctx.read_raf(lk["idx"], lk["table"], lk["raf"], lk["n_tables"])
        self.lookup_lists = ffi.lookup_suffix_lists()  # LookupTableKind::suffixes() of the 42 tables
        # the packed output-claim facts of the kernel (claim_columns, instruction_read_raf.rs:1163-1173) as one-hot columns with K = 16, the size the per-lane
        # pushforward kernel is built for: tables 0..15, 16..31, 32..41 in three columns (a row is hot in the one its table falls into), and 0 on RAF rows
        # (one K = 64 column went through the generic kernel: 4.7 ms per proof against ~0.3)
        tab = lk["table"].astype(np.int32)
        claim_cols = [np.where((tab >= 16 * a) & (tab < 16 * a + 16), tab - 16 * a, 0xFF).astype(np.uint8) for a in range(3)]
        self.lookup_claim_columns = ctx.onehot(np.stack(claim_cols + [np.where(lk["raf"] != 0, 0, 0xFF).astype(np.uint8)]), 16)
        bc = d["bytecode"]
        self.pc_ints = ctx.ints(bc["push_pc"])  # the address phase's PC column (unmapped rows on 0) and the cycle phase's chunk columns (unmapped rows cold)
        self.pc_chunks = ctx.onehot(bc["chunk_cols"], 1 << bc["chunk_bits"])
        # ---- input claims (in a real proof the previous stage's output claims): computed once, untimed
        self.claims = {}
        az, bz = ctx.r1cs_materialize_small(self.outer_ints, d["outer_wa"], d["outer_wb"])
        m = ctx.member_split_eq_product(az, bz, d["outer_tau"], scale=d["outer_kernel"])
        self.claims["outer"] = m.input_claim()
        m.destroy()
        left, right = ctx.r1cs_materialize_small(self.product_ints, self.product_fa, self.product_fb, streams=1)
        m = ctx.member_split_eq_product(left, right, d["product_tau"], scale=d["product_kernel"])
        self.claims["product"] = m.input_claim()
        m.destroy()
        # RAM: sum_j eq(tau, j) * [access_j] * (pre_j + gamma * post_j)  (val = the word before the access, val + inc = the word after)
        acc = ram["addresses"] != NO_ACCESS
        eq = ctx.eq_evals(d["ram_tau"])
        pre, post = ctx.from_u64(np.where(acc, ram["pre"], 0).astype(np.uint64)), ctx.from_u64(np.where(acc, ram["post"], 0).astype(np.uint64))
        m = ctx.member_lc([eq, pre, post], [[(None, [(self.one, 0)]), (None, [(self.one, 1), (d["ram_gamma"], 2)])]], 2, borrow=True)
        self.claims["ram"] = m.input_claim()
        m.destroy()
        for t in (eq, pre, post):
            t.free()
        # registers: sum_j eq(r_cycle, j) * ( [rd_j] * (inc_j + rd_pre_j) + gamma * [rs1_j] * rs1_val_j + gamma^2 * [rs2_j] * rs2_val_j ) = sum_j eq * (rd_post + g rs1_val + g^2 rs2_val)
        # (cold rows carry zero values in the trace columns)
        eq = ctx.eq_evals(d["registers_r_cycle"])
        t_post, t_rs1, t_rs2 = (ctx.table_from_ints(self.reg_cols[k]) for k in (3, 0, 1))
        g = d["registers_gamma"]
        m = ctx.member_lc([eq, t_post, t_rs1, t_rs2], [[(None, [(self.one, 0)]), (None, [(self.one, 1), (g, 2), (ffi.host_fr_mul(g, g), 3)])]], 2, borrow=True)
        self.claims["registers"] = m.input_claim()
        m.destroy()
        for t in (eq, t_post, t_rs1, t_rs2):
            t.free()
        self.claims["lookup"] = None  # taken from the first proof's first address message (same every proof)
        ctx.synchronize()

    @property
    def ctx(self):
        return getattr(self._tls, "ctx", None) or self._home_ctx

    def bind_context(self, ctx):
        """operators the CALLING thread creates from now on live on `ctx` (its stream, scratch and pool); the resident inputs stay where they are and are only read"""
        self._tls.ctx = ctx

    # ---- the operators: each one a jolt_stage_op (csrc/stage_ops.hip) -- created (= the slot's PrepareKernel::prepare), driven through the ProveRounds contract by a
    # ---- driver of the library (prove_batch over operators, or alone against a test transcript), asked for its output claims, destroyed.  Nothing else happens here.
    def _alone(self, op, claim, label):
        tr = self.ffi.HostTranscript(label)
        out = op.prove_alone(tr, claim)
        tr.close()
        return out

    def _batch(self, op, claim, rounds, degree, label):
        out = self.ctx.prove_batch_ops([op], [claim], [self.one], [0], rounds, degree, label=label)
        return dict(polys=out["polys"], challenges=out["challenges"], final_claim=out["final_claim"])

    def spartan(self, cols, iwa, iwb, fa, fb, tau, kernel, claim, streams, label):
        ctx, ffi = self.ctx, self.ffi
        sums = ctx.stage_spartan_uniskip_sums(cols, tau, iwa, iwb, streams)
        tr = ffi.HostTranscript(label)
        tr.append(sums)
        r0 = tr.challenge()  # the uni-skip challenge (the Lagrange weights of the remainder are a function of it; fixed weights here)
        tr.close()
        op = ctx.stage_spartan_remainder(cols, fa, fb, tau, kernel, streams)
        out = self._batch(op, claim, len(tau), 3, label + 1)
        values = op.output_claims()
        op.destroy()
        return dict(sums=sums, r0=r0, polys=out["polys"], challenges=out["challenges"], final_claim=out["final_claim"], values=values)

    def ram_read_write(self, label):
        d = self.d
        op = self.ctx.stage_ram_read_write(self.ram_cols[0], self.ram_cols[1], self.ram_cols[2], self.ram_inc, self.ram_val_init, d["ram_tau"], d["ram_gamma"])
        out = self._alone(op, self.claims["ram"], label)
        out["final_values"] = op.output_claims()
        op.destroy()
        return out

    def registers_read_write(self, label):
        d = self.d
        op = self.ctx.stage_registers_read_write(self.reg_idx, *self.reg_cols, self.reg_inc, d["registers_r_cycle"], d["registers_gamma"])
        out = self._alone(op, self.claims["registers"], label)
        claims = op.output_claims()
        op.destroy()
        out["final_values"], out["operand_claims"] = claims[:5], claims[5:7]  # RegistersReadWriteOutputClaims::{rs1_ra, rs2_ra} last
        return out

    def booleanity_address(self, label):
        bo = self.d["booleanity"]
        op = self.ctx.stage_booleanity_address(self.bool_cols, bo["reference_cycle"], bo["reference_address"], bo["gamma"])
        out = self._alone(op, np.zeros(4, dtype=np.uint64), label)
        out["intermediate"] = op.output_claims()[0]
        out["masses"] = op.kept("masses").reshape(bo["cols"].shape[0], 1 << bo["log_k"], 4)
        op.destroy()
        return out

    def booleanity_cycle(self, label, r_address, claim):
        """stage 6b after 6a: `r_address` = the address phase's bound point (its challenges, last first), `claim` = that phase's intermediate output claim (a wrong one
        fails the batch's first round check)"""
        bo = self.d["booleanity"]
        op = self.ctx.stage_booleanity_cycle(self.bool_cols, r_address, bo["reference_address"], bo["reference_cycle"], bo["gamma"])
        out = self._batch(op, claim, self.n_vars, 3, label)
        out["claim"] = claim
        out["ra_claims"] = op.output_claims()
        out["eq_scalar"] = op.kept("eq_scalar")[0]
        op.destroy()
        return out

    def hamming_weight(self, label):
        bo, hw = self.d["booleanity"], self.d["hamming"]
        op = self.ctx.stage_hamming_weight(self.bool_cols, hw["r_cycle"], hw["r_address"], hw["virtualization_points"], hw["gamma"])
        claim = op.input_claim()
        out = self._alone(op, claim, label)
        out["claim"] = claim
        out["g_claims"] = op.output_claims()
        out["masses"] = op.kept("masses").reshape(bo["cols"].shape[0], 1 << bo["log_k"], 4)
        op.destroy()
        return out

    def instruction_read_raf(self, label):
        """OptimizedInstructionReadRafKernel as ONE operator of 128 + log T rounds; the address rounds run against the test transcript `label`, the cycle rounds as a
        one-member batch under `label + 1` (two windows of the operator), as the oracle twin does"""
        ctx, d = self.ctx, self.d
        lk = d["lookup"]
        present = np.zeros(N_LOOKUP_TABLES, dtype=np.uint8)
        present[lk["present"]] = 1
        op = ctx.stage_instruction_read_raf(self.read_raf, self.lookup_claim_columns, d["lookup_reduction"], d["lookup_gamma"], present, d["ra_count"])
        if self.claims["lookup"] is None:  # the relation's input claim (the prover holds it from the earlier stages): summed from the first phase's tables, once
            self.claims["lookup"] = op.input_claim()
        address, cycle = op.window(0, ADDRESS_BITS), op.window(ADDRESS_BITS, self.n_vars)
        adr = self._alone(address, self.claims["lookup"], label)
        n_f = 1 + d["ra_count"]
        out = self._batch(cycle, adr["final_claim"], self.n_vars, n_f + 1, label + 1)  # its round check holds only if the address rounds were right
        claims = op.output_claims()
        n_present = int(present.sum())
        raf_scans = op.kept("scan_raf").reshape(PHASES, 6, CHUNK, 4)
        suf_scans = op.kept("scan_suffix").reshape(PHASES, -1, CHUNK, 4)
        res = dict(lookup_table_flags=claims[:n_present], instruction_raf_flag=claims[n_present], instruction_ra=claims[n_present + 1:],
                   scans=[(raf_scans[ph], suf_scans[ph]) for ph in range(PHASES)], address_polys=np.stack(adr["polys"]), address_challenges=adr["challenges"],
                   v_tables=op.kept("v_tables").reshape(PHASES, CHUNK, 4), table_values=op.kept("table_values")[lk["present"]], raf_values=op.kept("raf_values"),
                   cycle_claim=op.kept("cycle_claim")[0], polys=out["polys"], challenges=out["challenges"], final_claim=out["final_claim"])
        address.destroy()
        cycle.destroy()
        op.destroy()
        return res

    def bytecode_read_raf(self, label):
        """stage 6a / 6b: bytecode read + RAF over a sorted index of the PC column (per-proof work: the reference builds its PC rows once per proof)"""
        ctx, d, n_vars = self.ctx, self.d, self.n_vars
        bc = d["bytecode"]
        pc_index = ctx.key_index(self.pc_ints, 1 << bc["log_k"])
        first_pc = int(bc["first_pc"]) if "first_pc" in bc else int(bc["push_pc"][0])
        a_op = ctx.stage_bytecode_read_raf_address(pc_index, bc["stage_points"], bc["stage_values"], bc["gamma"], first_pc, bc["entry_index"])
        claim_a = a_op.input_claim()
        adr = self._batch(a_op, claim_a, bc["log_k"], 2, label)
        fin = a_op.output_claims()  # the 13 bound tables, then the intermediate claim
        c_op = ctx.stage_bytecode_read_raf_cycle(a_op, self.pc_chunks, bc["chunk_bits"])
        claim_c = c_op.input_claim()
        cyc = self._batch(c_op, claim_c, n_vars, c_op.degree, label + 1)
        bytecode = dict(address=adr, claim_address=claim_a, intermediate=fin[13], val_stages=fin[5:10], r_address=adr["challenges"][::-1], cycle=cyc, claim_cycle=claim_c,
                        ra_claims=c_op.output_claims())
        c_op.destroy()
        a_op.destroy()
        pc_index.free()
        return bytecode

    def ram_address_domain(self, label):
        """stage 2: RAM RAF evaluation and the RAM output check over a sorted index of the address column (RamAccessColumns, once per proof)"""
        ctx, d = self.ctx, self.d
        ram, raf, io = d["ram"], d["ram_raf"], d["ram_output"]
        ram_index = ctx.key_index(self.ram_cols[0], 1 << ram["log_k"])
        op = ctx.stage_ram_raf_evaluation(ram_index, raf["tau_low"], raf["lowest_address"])
        claim = op.input_claim()
        raf_out = self._batch(op, claim, ram["log_k"], 2, label + 10)
        raf_out["claim"] = claim
        raf_out["ra_claim"] = op.output_claims()[0]
        op.destroy()
        op = ctx.stage_ram_output_check(ram_index, self.ram_cols[2], ram["val_init"], io["val_io"], io["io_lo"], io["io_len"], io["point"])
        claim = op.input_claim()
        oc = self._batch(op, claim, ram["log_k"], 3, label + 20)
        oc["claim"] = claim
        oc["val_final_claim"] = op.output_claims()[0]
        op.destroy()
        ram_index.free()
        return {"ram_raf_evaluation": raf_out, "ram_output_check": oc}

    def address_domain(self, label):
        """the joint-domain relations whose rounds run over K-sized tables: bytecode read+RAF (6a, 6b), RAM RAF evaluation, RAM output check"""
        return {"bytecode_read_raf": self.bytecode_read_raf(label), **self.ram_address_domain(label)}

    # The operators by PROTOCOL STAGE (crates/jolt-prover/src/stages: the members of one stage are proved as one batch and share no challenge across members; a stage's
    # inputs are the previous stage's outputs).  `prove` runs the stages in order; DeviceWorkload.step runs each stage's operators here BESIDE that stage's batched sumcheck
    # of the cycle-domain catalogue (own context, own host thread), never across a stage boundary.
    STAGES = (1, 2, 4, 5, 6, 7)

    def stage_chains(self, stage, label=0):
        """the operators of one protocol stage as independent CHAINS: [(slot, fn)] -- fn() -> {name: output}; chains of a stage share no challenge and may run at the same
        time on different contexts (slot 0: the home context, which owns the read-RAF rows' mutable order buffers; slot 1: a second one).  Inside a chain the order matters
        (the booleanity cycle phase starts from the address phase's bound point and claim)."""
        d = self.d
        if stage == 1:
            return [(0, lambda: {"spartan_outer": self.spartan(self.outer_ints, d["outer_iwa"], d["outer_iwb"], d["outer_wa"], d["outer_wb"], d["outer_tau"], d["outer_kernel"],
                                                               self.claims["outer"], 2, label + 100)})]
        if stage == 2:
            return [(0, lambda: {"ram_read_write": self.ram_read_write(label + 300)}),
                    (1, lambda: {"spartan_product": self.spartan(self.product_ints, self.product_ia, self.product_ib, self.product_fa, self.product_fb, d["product_tau"],
                                                                 d["product_kernel"], self.claims["product"], 1, label + 200), **self.ram_address_domain(label + 500)})]
        if stage == 4:
            return [(0, lambda: {"registers_read_write": self.registers_read_write(label + 350)})]
        if stage == 5:
            return [(0, lambda: {"instruction_read_raf": self.instruction_read_raf(label + 400)})]
        if stage == 6:
            def booleanity():
                address = self.booleanity_address(label + 450)
                return {"booleanity_address": address, "booleanity_cycle": self.booleanity_cycle(label + 460, address["challenges"][::-1], address["intermediate"])}
            return [(0, lambda: {"bytecode_read_raf": self.bytecode_read_raf(label + 500)}), (1, booleanity)]
        if stage == 7:
            return [(0, lambda: {"hamming_weight": self.hamming_weight(label + 470)})]
        return []

    def prove_stage(self, stage, label=0):
        out = {}
        for _, chain in self.stage_chains(stage, label):
            out.update(chain())
        return out

    def prove(self, label=0):
        out = {}
        for stage in self.STAGES:
            out.update(self.prove_stage(stage, label))
        return out

    def close(self):
        for c in self.outer_ints + self.product_ints + self.ram_cols + self.reg_cols + [self.ram_inc, self.ram_val_init, self.reg_inc, self.reg_idx, self.bool_cols, self.pc_ints, self.pc_chunks]:
            c.free()
        self.read_raf.free()
        self.lookup_claim_columns.free()
