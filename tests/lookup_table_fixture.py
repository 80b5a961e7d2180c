"""Lookup tables of the instruction read+RAF relation on Python integers, and the reference's test recipes for them (TEST INFRASTRUCTURE).

`table_model` restates LookupTable::materialize_entry of crates/jolt-lookup-tables/src/tables/*.rs a third time, from what each table MEANS
(the RISC-V operation on the two de-interleaved 64-bit operands), so that oracle/lookup_tables.c -- written on machine words -- is pinned by
something that shares none of its loops.  `fixture_rows`, `challenge`, `splitmix` follow the reference kernel's own parity test
(crates/jolt-kernels/src/optimized/instruction_read_raf.rs:1477-1522) constant for constant."""
import numpy as np

M64, M32 = (1 << 64) - 1, (1 << 32) - 1
TABLES = ["RangeCheck", "RangeCheckAligned", "And", "Andn", "Or", "Xor", "Equal", "SignedGreaterThanEqual", "UnsignedGreaterThanEqual", "NotEqual", "SignedLessThan",
          "UnsignedLessThan", "SignMask", "UpperWord", "UnsignedLessThanEqual", "ValidUnsignedRemainder", "ValidDiv0", "HalfwordAlignment", "WordAlignment", "LowerHalfWord",
          "SignExtendWord", "Pow2", "Pow2W", "ShiftRightBitmask", "VirtualRev8W", "VirtualSRL", "VirtualSRA", "VirtualROTR", "VirtualROTRW", "VirtualChangeDivisor",
          "VirtualChangeDivisorW", "MulUNoOverflow", "VirtualXORROT32", "VirtualXORROT24", "VirtualXORROT16", "VirtualXORROT63", "VirtualXORROTW16", "VirtualXORROTW12",
          "VirtualXORROTW8", "VirtualXORROTW7", "WindowMaskW", "PextSigned"]
# tables whose decomposition (and whose integer recurrences) assume a right operand of the form 1..10..0 (tables/mod.rs:283-290, test_utils.rs:31-38)
BITMASK_TABLES = {"VirtualSRL", "VirtualSRA", "VirtualROTR", "VirtualROTRW"}


def operands(index):
    x = sum(((index >> (2 * k + 1)) & 1) << k for k in range(64))
    y = sum(((index >> (2 * k)) & 1) << k for k in range(64))
    return x, y


def interleave(x, y):
    return sum((((x >> k) & 1) << (2 * k + 1)) | (((y >> k) & 1) << (2 * k)) for k in range(64))


def signed(v, bits=64):
    return v - (1 << bits) if (v >> (bits - 1)) & 1 else v


def ror(v, k, width):
    k %= width
    return ((v >> k) | (v << (width - k))) & ((1 << width) - 1)


def table_model(name, index):
    """the table entry as the instruction semantics say it; for the bitmask tables the right operand is 1..10..0 with s zeros = the shift amount"""
    x, y = operands(index)
    low, high = index & M64, index >> 64
    if name == "RangeCheck": return low
    if name == "RangeCheckAligned": return low & ~1
    if name == "And": return x & y
    if name == "Andn": return x & ~y & M64
    if name == "Or": return x | y
    if name == "Xor": return x ^ y
    if name == "Equal": return int(x == y)
    if name == "NotEqual": return int(x != y)
    if name == "SignedGreaterThanEqual": return int(signed(x) >= signed(y))
    if name == "UnsignedGreaterThanEqual": return int(x >= y)
    if name == "SignedLessThan": return int(signed(x) < signed(y))
    if name == "UnsignedLessThan": return int(x < y)
    if name == "UnsignedLessThanEqual": return int(x <= y)
    if name == "SignMask": return M64 if index >> 127 else 0
    if name == "UpperWord": return high
    if name == "ValidUnsignedRemainder": return int(y == 0 or x < y)
    if name == "ValidDiv0": return int(y == M64) if x == 0 else 1
    if name == "HalfwordAlignment": return int(index % 2 == 0)
    if name == "WordAlignment": return int(index % 4 == 0)
    if name == "LowerHalfWord": return index & M32
    if name == "SignExtendWord": return signed(index & M32, 32) & M64
    if name == "Pow2": return 1 << (index % 64)
    if name == "Pow2W": return 1 << (index % 32)
    if name == "ShiftRightBitmask": return (M64 >> (index % 64)) << (index % 64)
    if name == "VirtualRev8W":
        swap = lambda w: int.from_bytes(w.to_bytes(4, "little"), "big")
        return swap(low & M32) | (swap(low >> 32) << 32)
    if name in BITMASK_TABLES:
        shift = (y & -y).bit_length() - 1 if y else 64  # trailing zeros of the mask
        if name == "VirtualSRL": return x >> shift
        if name == "VirtualSRA": return (signed(x) >> shift) & M64
        if name == "VirtualROTR": return ror(x, shift, 64)
        shift_w = ((y & M32) & -(y & M32)).bit_length() - 1 if y & M32 else 32
        return ror(x & M32, shift_w, 32)
    if name == "VirtualChangeDivisor": return 1 if (x == 1 << 63 and y == M64) else y
    if name == "VirtualChangeDivisorW":
        return 1 if (x & M32 == 1 << 31 and y & M32 == M32) else signed(y & M32, 32) & M64
    if name == "MulUNoOverflow": return int(high == 0)
    if name.startswith("VirtualXORROTW"): return ror((x ^ y) & M32, int(name[14:]), 32)
    if name.startswith("VirtualXORROT"): return ror(x ^ y, int(name[13:]), 64)
    if name == "WindowMaskW": return M32 << (32 * ((index >> 2) & 1))
    if name == "PextSigned":
        if y == 0:
            return 0
        positions = [p for p in range(64) if (y >> p) & 1]
        packed = sum(((x >> p) & 1) << k for k, p in enumerate(positions))
        return signed(packed, len(positions)) & M64  # the extracted window, sign-extended by its own top bit
    raise KeyError(name)


# An index on which the REFERENCE's decomposition of VirtualChangeDivisorW is not the table's multilinear extension while the suffix still holds bit 31 of the
# left operand (phases 0 .. 7 of 8-bit phases): the ChangeDivisorW prefix is zero above the low lane (prefixes/change_divisor_w.rs:16-19) and the suffix asks for
# a ZERO low left operand (suffixes/change_divisor_w.rs:12-16), so the (MIN32, -1) adjustment 2 - 2^64 is absent until phase 8 picks it up.  The product follows
# the reference formula for formula; the from-the-definition oracle does not, so parity rows leave this index out -- as the reference's own tests do (uniformly
# random indices, tables/test_utils.rs:64-84, never draw it).  tests/test_read_raf_address_cpu.py pins the behaviour.
CHANGE_DIVISOR_W_CORNER = interleave((0x1234 << 32) | (1 << 31), (0x77 << 32) | M32)


def bitmask_index(rng):
    """gen_bitmask_lookup_index (tables/test_utils.rs:31-38): random left operand, right operand = ones then `zeros` zeros"""
    x = int(rng.integers(0, 2**64, dtype=np.uint64))
    zeros = int(rng.integers(0, 65))
    y = (M64 << zeros) & M64
    return interleave(x, y)


def random_index(name, rng):
    if name in BITMASK_TABLES:
        return bitmask_index(rng)
    return int.from_bytes(rng.bytes(16), "little")


def shaped_index(name, rng):
    """indices with the shapes real operands have (small values, equal operands, sign patterns, all ones): the comparison / division / alignment tables are
    constant on uniformly random indices"""
    if name in BITMASK_TABLES:
        return bitmask_index(rng)
    pattern = int(rng.integers(0, 10))
    x = int(rng.integers(0, 2**64, dtype=np.uint64))
    y = int(rng.integers(0, 2**64, dtype=np.uint64))
    if pattern == 0: y = x
    elif pattern == 1: x = 0
    elif pattern == 2: y = 0
    elif pattern == 3: y = M64
    elif pattern == 4: x, y = 1 << 63, M64
    elif pattern == 5 and name != "VirtualChangeDivisorW": x, y = (x & ~M32) | (1 << 31), y | M32  # (MIN32, -1) in the low lanes; see CHANGE_DIVISOR_W_CORNER
    elif pattern == 6: x, y = x & 0xFF, y & 0xFF
    elif pattern == 7: x, y = 0, y & 0xFFFF  # upper word of the index mostly zero
    elif pattern == 8: y = x ^ (1 << int(rng.integers(0, 64)))
    return interleave(x, y)


# ---- the reference kernel's parity-test recipe (instruction_read_raf.rs:1477-1522) ----
def splitmix(state):
    state[0] = (state[0] + 0x9E3779B97F4A7C15) & M64
    z = state[0]
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def challenge(round_index):
    return 0x9E3779B97F4A7C15 ^ ((round_index * 0xBF58476D1CE4E5B9) & M64) ^ 0x11


def fixture_rows(log_t, seed, all_raf=False):
    """(lookup_index (T, 2) u64 lo / hi, table_index u8 with 0xFF = none, raf_flag u8)"""
    count = len(TABLES)
    tables = [0, 3 % count, 7 % count, 11 % count, count - 1]
    state = [seed]
    T = 1 << log_t
    idx = np.zeros((T, 2), dtype=np.uint64)
    tab = np.zeros(T, dtype=np.uint8)
    raf = np.zeros(T, dtype=np.uint8)
    for j in range(T):
        if j == 0:
            k = 0
        elif j == 1:
            k = (1 << 128) - 1
        elif j == 2:
            k = (M64 << 64) | splitmix(state)
        else:
            hi = splitmix(state)
            k = (hi << 64) | splitmix(state)
        idx[j] = (k & M64, k >> 64)
        tab[j] = 0xFF if j % 7 == 3 else tables[j % len(tables)]
        raf[j] = 1 if (all_raf or j % 3 == 0) else 0
    return idx, tab, raf


def all_table_rows(log_t, seed, none_fraction=0.08, raf_fraction=0.3):
    """every one of the 42 tables present, indices of the shapes each table's decomposition is defined on"""
    rng = np.random.default_rng(seed)
    T = 1 << log_t
    idx = np.zeros((T, 2), dtype=np.uint64)
    tab = np.zeros(T, dtype=np.uint8)
    for j in range(T):
        t = j % len(TABLES) if j < 2 * len(TABLES) else int(rng.integers(0, len(TABLES)))
        k = shaped_index(TABLES[t], rng) if rng.random() < 0.6 else random_index(TABLES[t], rng)
        idx[j] = (k & M64, k >> 64)
        tab[j] = 0xFF if rng.random() < none_fraction else t
    raf = (rng.random(T) < raf_fraction).astype(np.uint8)
    return idx, tab, raf
