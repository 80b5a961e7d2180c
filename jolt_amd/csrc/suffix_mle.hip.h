// jolt_amd/csrc/suffix_mle.hip.h -- the suffix polynomials of the prefix-suffix decomposition, evaluated on the device.
//
// Replaces Suffixes::suffix_mle (crates/jolt-lookup-tables/src/tables/suffixes/mod.rs:200-252 and the 42 files next to it) for the
// T-scale scans of instruction read-RAF checking (read_raf.hip).  A suffix is a function of the UNBOUND low bits of a lookup index:
// `bits` (already masked to `len` bits, LookupBits::new, lookup_bits.rs:14-24) and `len`; x / y are the operands de-interleaved from
// the odd / even bit positions (uninterleave_bits, interleave.rs:38-58; x has len / 2 bits, y the rest).  Values are u64, XLEN = 64.
// The kind ids ARE the discriminants of `enum Suffixes` (#[repr(u8)], mod.rs:120-170) so that the Rust shim passes
// `table.suffixes()` through unchanged.  Shift amounts at or above the operand width follow the reference: `unbounded_shl / shr`
// give zero; plain `1 << k` never reaches 64 for the suffix lengths a proof uses (multiples of 8 up to 120, operands <= 60 bits).
#pragma once
#include <cstdint>

namespace jolt {

constexpr int kNumSuffixKinds = 48;
enum SuffixKind : uint8_t {
    kSufOne = 0, kSufAnd, kSufAndNot, kSufXor, kSufOr, kSufRightOperand, kSufRightOperandW, kSufChangeDivisor, kSufChangeDivisorW, kSufUpperWord,
    kSufLowerWord, kSufLowerHalfWord, kSufLessThan, kSufGreaterThan, kSufEq, kSufLeftOperandIsZero, kSufRightOperandIsZero, kSufLsb, kSufDivByZero,
    kSufPow2, kSufPow2W, kSufRev8W, kSufRightShiftPadding, kSufRightShift, kSufRightShiftHelper, kSufSignExtension, kSufLeftShift, kSufTwoLsb,
    kSufSignExtensionUpperHalf, kSufSignExtensionRightOperand, kSufRightShiftW, kSufRightShiftWHelper, kSufLeftShiftWHelper, kSufLeftShiftW,
    kSufOverflowBitsZero, kSufXorRot16, kSufXorRot24, kSufXorRot32, kSufXorRot63, kSufXorRotW16, kSufXorRotW12, kSufXorRotW8, kSufXorRotW7,
    kSufPow2OffsetW, kSufPext, kSufPextHelper, kSufWindowSign, kSufWindowSignPow2
};

// the even bit positions of a 64-bit word packed into 32 bits
__host__ __device__ inline uint32_t compact_even_bits(uint64_t v) {
    v &= 0x5555555555555555ull;
    v = (v | (v >> 1)) & 0x3333333333333333ull;
    v = (v | (v >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    v = (v | (v >> 4)) & 0x00FF00FF00FF00FFull;
    v = (v | (v >> 8)) & 0x0000FFFF0000FFFFull;
    v = (v | (v >> 16)) & 0x00000000FFFFFFFFull;
    return (uint32_t)v;
}
struct Operands {
    uint64_t x, y;
    uint32_t x_len, y_len;
};
__host__ __device__ inline Operands uninterleave(uint64_t lo, uint64_t hi, uint32_t len) {
    Operands o;
    o.y = (uint64_t)compact_even_bits(lo) | ((uint64_t)compact_even_bits(hi) << 32);
    o.x = (uint64_t)compact_even_bits(lo >> 1) | ((uint64_t)compact_even_bits(hi >> 1) << 32);
    o.x_len = len / 2;
    o.y_len = len - o.x_len;
    return o;
}
// Bit counts.  On the device these are single instructions (s_ff1 / v_ffbl, bcnt, ffbh); the bit loops are the definitions the host build keeps, and
// tests/test_oracle_read_raf.py / test_gpu_read_raf.py hold the two against the oracle.  (As loops on the device they made a phase scan's time follow the
// suffix length: 1.4 ms at 120 unbound bits against 0.23 ms at 0.)
__host__ __device__ inline uint32_t ctz64(uint64_t v) {  // 64 for zero
    if (v == 0) return 64;
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__ffsll((unsigned long long)v) - 1u;
#else
    uint32_t n = 0;
    while (!(v & 1)) { v >>= 1; ++n; }
    return n;
#endif
}
__host__ __device__ inline uint32_t popcount64(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popcll((unsigned long long)v);
#else
    uint32_t n = 0;
    while (v) { v &= v - 1; ++n; }
    return n;
#endif
}
__host__ __device__ inline uint32_t top_bit64(uint64_t v) {  // index of the highest set bit, v != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return 63u - (uint32_t)__clzll((long long)v);
#else
    uint32_t top = 63;
    while (!((v >> top) & 1)) --top;
    return top;
#endif
}
// LookupBits::leading_ones of a value of `len` bits (lookup_bits.rs:66-70), len <= 64: the distance from bit len - 1 down to the highest ZERO bit
__host__ __device__ inline uint32_t leading_ones_in(uint64_t v, uint32_t len) {
    if (len == 0) return 0;
    const uint64_t zeros = ~v & (len >= 64 ? ~0ull : ((1ull << len) - 1));
    return zeros == 0 ? len : len - 1 - top_bit64(zeros);
}
__host__ __device__ inline uint64_t shl_unbounded(uint64_t v, uint32_t k) { return k >= 64 ? 0 : v << k; }
__host__ __device__ inline uint64_t shr_unbounded(uint64_t v, uint32_t k) { return k >= 64 ? 0 : v >> k; }
__host__ __device__ inline uint32_t shl32_unbounded(uint32_t v, uint32_t k) { return k >= 32 ? 0 : v << k; }
__host__ __device__ inline uint32_t shr32_unbounded(uint32_t v, uint32_t k) { return k >= 32 ? 0 : v >> k; }
__host__ __device__ inline uint64_t rotr64(uint64_t v, uint32_t k) { return (v >> k) | (v << ((64 - k) & 63)); }
__host__ __device__ inline uint32_t rotr32(uint32_t v, uint32_t k) { return (v >> k) | (v << ((32 - k) & 31)); }
__host__ __device__ inline uint32_t bswap32(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u) | (v << 24); }
// pext (suffixes/pext.rs:12-35): the bits of x at the set positions of y, packed lowest first
__host__ __device__ inline uint64_t pext64(uint64_t x, uint64_t y) {
    uint64_t out = 0;
    uint32_t k = 0;
    while (y) {
        out |= ((x >> ctz64(y)) & 1) << k;
        ++k;
        y &= y - 1;
    }
    return out;
}
// window_sign_bit (suffixes/window_sign.rs:9-16): bit ilog2(y) of x
__host__ __device__ inline uint64_t window_sign(uint64_t x, uint64_t y) {
    if (y == 0) return 0;
    return (x >> top_bit64(y)) & 1;
}

__host__ __device__ inline uint64_t suffix_mle(uint32_t kind, uint64_t lo, uint64_t hi, uint32_t len) {
    const Operands o = uninterleave(lo, hi, len);
    const uint64_t x = o.x, y = o.y;
    const uint32_t y_len_w = o.y_len < 32 ? o.y_len : 32;  // the W (32-bit) tables clip the right operand
    const uint64_t y_w = y_len_w >= 64 ? y : (y & ((1ull << y_len_w) - 1));  // LookupBits::new(y, min(y.len, 32))
    switch (kind) {
        case kSufOne: return 1;
        case kSufAnd: return x & y;
        case kSufAndNot: return x & ~y;
        case kSufXor: return x ^ y;
        case kSufOr: return x | y;
        case kSufRightOperand: return y;
        case kSufRightOperandW: return (uint32_t)y;
        case kSufChangeDivisor: return (shl_unbounded(1, o.y_len) - 1 == y && x == 0) ? 1 : 0;
        case kSufChangeDivisorW: return ((1ull << y_len_w) - 1 == (uint64_t)(uint32_t)y && (uint32_t)x == 0) ? 1 : 0;
        case kSufUpperWord: return hi;
        case kSufLowerWord: return lo;
        case kSufLowerHalfWord: return (uint32_t)lo;
        case kSufLessThan: return x < y ? 1 : 0;
        case kSufGreaterThan: return x > y ? 1 : 0;
        case kSufEq: return x == y ? 1 : 0;
        case kSufLeftOperandIsZero: return x == 0 ? 1 : 0;
        case kSufRightOperandIsZero: return y == 0 ? 1 : 0;
        case kSufLsb: return len == 0 ? 1 : (lo & 1);
        case kSufDivByZero: return (x == 0 && y == shl_unbounded(1, o.y_len) - 1) ? 1 : 0;  // (divisor, quotient) = (x, y)
        case kSufPow2: return len == 0 ? 1 : 1ull << (lo & 63);
        case kSufPow2W: return len == 0 ? 1 : 1ull << (lo & 31);
        case kSufRev8W: return (uint64_t)bswap32((uint32_t)lo) + ((uint64_t)bswap32((uint32_t)(lo >> 32)) << 32);
        case kSufRightShiftPadding: return len == 0 ? 1 : 1ull << (63 - (lo & 63));
        case kSufRightShift: { const uint32_t tz = ctz64(y) < o.y_len ? ctz64(y) : o.y_len; return shr_unbounded(x, tz); }
        case kSufRightShiftHelper: return shl_unbounded(1, leading_ones_in(y, o.y_len));
        case kSufSignExtension: {
            const uint32_t tz = ctz64(y), pad = tz < o.y_len ? tz : o.y_len;  // min(trailing_zeros(y as u64), y.len)
            return pad == 0 ? 0 : ~0ull << (64 - pad);  // 2^64 - 2^(64 - pad)
        }
        case kSufLeftShift: return shl_unbounded(x & ~y, leading_ones_in(y, o.y_len));
        case kSufTwoLsb: return (len == 0 || (lo & 3) == 0) ? 1 : 0;
        case kSufSignExtensionUpperHalf: return len >= 32 ? (((lo >> 31) & 1) ? 0xFFFFFFFF00000000ull : 0) : 1;
        case kSufSignExtensionRightOperand: return len >= 64 ? (((lo >> 62) & 1) ? 0xFFFFFFFF00000000ull : 0) : 1;
        case kSufRightShiftW: { uint32_t tz = ctz64(y) < o.y_len ? ctz64(y) : o.y_len; if (tz > 32) tz = 32; return shr32_unbounded((uint32_t)x, tz); }
        case kSufRightShiftWHelper: return shl_unbounded(1, leading_ones_in(y_w, y_len_w));
        case kSufLeftShiftWHelper: return (uint64_t)(1u << (leading_ones_in(y, o.y_len) & 31));  // 1u32 << k: release-mode shift (amount taken mod 32)
        case kSufLeftShiftW: return shl32_unbounded((uint32_t)x & ~(uint32_t)y_w, leading_ones_in(y_w, y_len_w));
        case kSufOverflowBitsZero: return hi == 0 ? 1 : 0;
        case kSufXorRot16: return rotr64(x ^ y, 16);
        case kSufXorRot24: return rotr64(x ^ y, 24);
        case kSufXorRot32: return rotr64(x ^ y, 32);
        case kSufXorRot63: return rotr64(x ^ y, 63);
        case kSufXorRotW16: return rotr32((uint32_t)x ^ (uint32_t)y, 16);
        case kSufXorRotW12: return rotr32((uint32_t)x ^ (uint32_t)y, 12);
        case kSufXorRotW8: return rotr32((uint32_t)x ^ (uint32_t)y, 8);
        case kSufXorRotW7: return rotr32((uint32_t)x ^ (uint32_t)y, 7);
        case kSufPow2OffsetW: return len < 3 ? 1 : 1ull << (32 * ((lo >> 2) & 1));
        case kSufPext: return pext64(x, y);
        case kSufPextHelper: return shl_unbounded(1, popcount64(y));
        case kSufWindowSign: return window_sign(x, y);
        case kSufWindowSignPow2: return shl_unbounded(window_sign(x, y), popcount64(y));
        default: return 0;
    }
}

}  // namespace jolt
