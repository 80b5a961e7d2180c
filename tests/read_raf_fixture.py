"""Synthetic instruction read+RAF rows and a Python big-integer model of the suffix polynomials (TEST INFRASTRUCTURE).

The model restates crates/jolt-lookup-tables/src/tables/suffixes/*.rs a third time -- on Python integers, with strings of bits where
that is the most literal reading -- so that the C oracle (oracle/read_raf.c) and the device code (suffix_mle.hip.h), both written on
machine words, are checked against something that shares none of their bit tricks."""
import numpy as np

XLEN = 64
KINDS = ["One", "And", "AndNot", "Xor", "Or", "RightOperand", "RightOperandW", "ChangeDivisor", "ChangeDivisorW", "UpperWord", "LowerWord", "LowerHalfWord", "LessThan",
         "GreaterThan", "Eq", "LeftOperandIsZero", "RightOperandIsZero", "Lsb", "DivByZero", "Pow2", "Pow2W", "Rev8W", "RightShiftPadding", "RightShift", "RightShiftHelper",
         "SignExtension", "LeftShift", "TwoLsb", "SignExtensionUpperHalf", "SignExtensionRightOperand", "RightShiftW", "RightShiftWHelper", "LeftShiftWHelper", "LeftShiftW",
         "OverflowBitsZero", "XorRot16", "XorRot24", "XorRot32", "XorRot63", "XorRotW16", "XorRotW12", "XorRotW8", "XorRotW7", "Pow2OffsetW", "Pext", "PextHelper", "WindowSign",
         "WindowSignPow2"]
ZERO_ONE = {"One", "Eq", "LessThan", "GreaterThan", "LeftOperandIsZero", "RightOperandIsZero", "Lsb", "TwoLsb", "DivByZero", "OverflowBitsZero", "ChangeDivisor", "ChangeDivisorW",
            "WindowSign"}
M64, M32 = (1 << 64) - 1, (1 << 32) - 1


def uninterleave(bits, length):
    """(x, x_len, y, y_len): x from the odd bit positions, y from the even ones"""
    x = sum(((bits >> (2 * k + 1)) & 1) << k for k in range(64))
    y = sum(((bits >> (2 * k)) & 1) << k for k in range(64))
    x_len = length // 2
    y_len = length - x_len
    return x % (1 << x_len) if x_len < 128 else x, x_len, y % (1 << y_len) if y_len < 128 else y, y_len


def trailing_zeros(v, cap):
    n = 0
    while n < cap and not (v >> n) & 1:
        n += 1
    return n


def leading_ones(v, length):
    s = format(v, "b").zfill(length)[-length:] if length else ""
    return len(s) - len(s.lstrip("1"))


def ror(v, k, width):
    k %= width
    return ((v >> k) | (v << (width - k))) & ((1 << width) - 1)


def suffix_model(kind, bits, length):
    name = KINDS[kind]
    bits %= 1 << length
    x, x_len, y, y_len = uninterleave(bits, length)
    u64 = lambda v: v & M64
    if name == "One": return 1
    if name == "And": return x & y
    if name == "AndNot": return x & (~y & M64)
    if name == "Xor": return x ^ y
    if name == "Or": return x | y
    if name == "RightOperand": return y
    if name == "RightOperandW": return y & M32
    if name == "ChangeDivisor": return int((1 << y_len) - 1 == y and x == 0)
    if name == "ChangeDivisorW": return int((1 << min(y_len, 32)) - 1 == (y & M32) and (x & M32) == 0)
    if name == "UpperWord": return u64(bits >> 64)
    if name == "LowerWord": return bits % (1 << 64)
    if name == "LowerHalfWord": return bits % (1 << 32)
    if name == "LessThan": return int(x < y)
    if name == "GreaterThan": return int(x > y)
    if name == "Eq": return int(x == y)
    if name == "LeftOperandIsZero": return int(x == 0)
    if name == "RightOperandIsZero": return int(y == 0)
    if name == "Lsb": return 1 if length == 0 else bits & 1
    if name == "DivByZero": return int(x == 0 and y == (1 << y_len) - 1)
    if name == "Pow2": return 1 if length == 0 else 1 << (bits % 64)
    if name == "Pow2W": return 1 if length == 0 else 1 << (bits % 32)
    if name == "Rev8W":
        v = u64(bits)
        swap = lambda w: int.from_bytes(w.to_bytes(4, "little"), "big")
        return swap(v & M32) + (swap(v >> 32) << 32)
    if name == "RightShiftPadding": return 1 if length == 0 else 1 << (63 - bits % 64)
    if name == "RightShift": return u64(x >> trailing_zeros(y, y_len))
    if name == "RightShiftHelper": return 1 << leading_ones(y, y_len)
    if name == "SignExtension":
        pad = min(trailing_zeros(y, 64), y_len)
        return u64((1 << 64) - (1 << (64 - pad)))
    if name == "LeftShift": return u64((x & (~y & M64)) << leading_ones(y, y_len))
    if name == "TwoLsb": return int(length == 0 or bits % 4 == 0)
    if name == "SignExtensionUpperHalf": return (((1 << 32) - 1) << 32 if (bits >> 31) & 1 else 0) if length >= 32 else 1
    if name == "SignExtensionRightOperand": return ((1 << 64) - (1 << 32) if (bits >> 62) & 1 else 0) if length >= 64 else 1
    if name == "RightShiftW":
        k = min(trailing_zeros(y, y_len), 32)
        return 0 if k >= 32 else (x & M32) >> k
    if name == "RightShiftWHelper":
        yl = min(y_len, 32)
        return 1 << leading_ones(y % (1 << yl), yl)
    if name == "LeftShiftWHelper": return (1 << (leading_ones(y, y_len) % 32)) & M32
    if name == "LeftShiftW":
        yl = min(y_len, 32)
        yw = y % (1 << yl)
        k = leading_ones(yw, yl)
        return 0 if k >= 32 else (((x & M32) & (~yw & M32)) << k) & M32
    if name == "OverflowBitsZero": return int(bits >> 64 == 0)
    if name.startswith("XorRotW"): return ror((x ^ y) & M32, int(name[7:]), 32)
    if name.startswith("XorRot"): return ror(u64(x ^ y), int(name[6:]), 64)
    if name == "Pow2OffsetW": return 1 if length < 3 else 1 << (32 * ((bits >> 2) & 1))
    if name == "Pext":
        out, k = 0, 0
        for pos in range(64):
            if (y >> pos) & 1:
                out |= ((x >> pos) & 1) << k
                k += 1
        return out
    if name == "PextHelper": return u64(1 << bin(y).count("1"))
    sign = 0 if y == 0 else (x >> (y.bit_length() - 1)) & 1
    if name == "WindowSign": return sign
    if name == "WindowSignPow2": return u64(sign << bin(y).count("1"))
    raise KeyError(name)


def interesting_bits(rng, length):
    """lookup-index suffixes that exercise the corner branches: zero, all ones, one operand zero / all ones, single bits, random"""
    full = (1 << length) - 1 if length else 0
    odd = sum(1 << k for k in range(1, 128, 2)) & full    # x all ones, y zero
    even = sum(1 << k for k in range(0, 128, 2)) & full   # y all ones, x zero
    out = [0, full, odd, even, 1 & full, 2 & full, 3 & full, 4 & full, (1 << 62) & full, (1 << 63) & full, (1 << 31) & full, full >> 1, full ^ 1]
    for _ in range(40):
        v = int.from_bytes(rng.bytes(16), "little") & full
        out += [v, v & even, v & odd, v | even, v | odd]
        hi_ones = int(rng.integers(0, 64))
        out.append((v | (even ^ (even >> (2 * hi_ones)))) & full)  # y with a run of leading ones
    return out


def make_rows(T, n_tables, seed, raf_fraction=0.3, none_fraction=0.1):
    rng = np.random.default_rng(seed)
    idx = np.frombuffer(rng.bytes(16 * T), dtype=np.uint64).reshape(T, 2).copy()
    # operands of real lookups are often small / sign-extended / equal: mix such shapes in so that the 0/1-valued suffixes fire
    for j in range(0, T, 3):
        pattern = j % 5
        if pattern == 0: idx[j] = 0
        elif pattern == 1: idx[j, 1] = 0
        elif pattern == 2: idx[j] = np.uint64(0xFFFFFFFFFFFFFFFF)
        elif pattern == 3: idx[j, 0] &= np.uint64(0xFF)
    table = rng.integers(0, n_tables, size=T).astype(np.uint8)
    table[rng.random(T) < none_fraction] = 0xFF
    raf = (rng.random(T) < raf_fraction).astype(np.uint8)
    return idx, table, raf


def suffix_lists(n_tables, seed):
    """every suffix kind appears in at least one table; tables hold 1..6 suffixes, the way LookupTableKind::suffixes() lists do"""
    rng = np.random.default_rng(seed)
    kinds = list(rng.permutation(len(KINDS)))
    lists = [[] for _ in range(n_tables)]
    for i, k in enumerate(kinds):
        lists[i % n_tables].append(int(k))
    for l in lists:
        if 0 not in l and rng.random() < 0.5:
            l.insert(0, 0)  # most tables carry Suffixes::One
    return lists
