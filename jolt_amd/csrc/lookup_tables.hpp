// jolt_amd/csrc/lookup_tables.hpp -- the prefix side of the prefix-suffix decomposition of the instruction lookup tables (host code).
//
// Instruction read+RAF checking binds its 128 address variables in 16 phases of 8 (read_raf_address.hip).  What is T-sized in a phase -- the suffix
// accumulators -- is summed on the device (read_raf.hip, suffix_mle.hip.h); what is left is 256 entries per polynomial and lives here:
//   * the 49 prefix polynomials restricted to one phase's chunk (`SparseDensePrefix::evaluate(checkpoints, b, suffix_len)`,
//     crates/jolt-lookup-tables/src/tables/prefixes/*.rs through Prefixes::evaluate, prefixes/mod.rs:236-250) and their initial checkpoints;
//   * for each of the 42 tables of `enum LookupTableKind` (tables/mod.rs:121-166) the prefixes / suffixes it reads and its `combine`
//     (PrefixSuffixDecomposition, tables/mod.rs:258-283 and the table files).
// Ids are the enums' discriminants, so a Rust caller passes `table.index()`, `Prefixes as u8`, `Suffixes as u8` through unchanged.
// Every `combine` of the reference is a sum of  coefficient * [prefix] * [suffix]  with coefficients in {+-1, 2^64, 2^64 - 1, 2^32 - 1}: the tables are
// DATA here (TableDesc), and one evaluator walks the terms.  A chunk `b` is `b_len` <= 16 bits (the reference's phases are 8; its own decomposition test
// also runs 16 and 2); x / y are its odd / even bit positions (LookupBits::uninterleave, lookup_bits.rs:40-45), x of b_len / 2 bits.
#pragma once
#include <cstdint>

#include "field.hip.h"
#include "suffix_mle.hip.h"

namespace jolt_lookup {

using jolt::Fr;

constexpr int kXlen = 64;
constexpr int kLogK = 2 * kXlen;
constexpr int kNumPrefixes = 49;
constexpr int kNumTables = 42;

enum Prefix : uint8_t {
    kPreLowerWord = 0, kPreLowerHalfWord, kPreUpperWord, kPreEq, kPreAnd, kPreAndn, kPreOr, kPreXor, kPreLessThan, kPreLeftOperandIsZero, kPreRightOperandIsZero,
    kPreLeftOperandMsb, kPreRightOperandMsb, kPreDivByZero, kPrePositiveRemainderEqualsDivisor, kPrePositiveRemainderLessThanDivisor, kPreNegativeDivisorZeroRemainder,
    kPreNegativeDivisorEqualsRemainder, kPreNegativeDivisorGreaterThanRemainder, kPreLsb, kPrePow2, kPrePow2W, kPreRev8W, kPreRightShift, kPreSignExtension, kPreLeftShift,
    kPreLeftShiftHelper, kPreTwoLsb, kPreSignExtensionUpperHalf, kPreChangeDivisor, kPreChangeDivisorW, kPreRightOperand, kPreRightOperandW, kPreSignExtensionRightOperand,
    kPreRightShiftW, kPreLeftShiftWHelper, kPreLeftShiftW, kPreOverflowBitsZero, kPreXorRot16, kPreXorRot24, kPreXorRot32, kPreXorRot63, kPreXorRotW7, kPreXorRotW8,
    kPreXorRotW12, kPreXorRotW16, kPrePow2OffsetW, kPreWindowSign, kPreWindowSignPow2
};
static_assert(kPreWindowSignPow2 == kNumPrefixes - 1, "Prefixes discriminants (prefixes/mod.rs:104-154)");

// ---- small field helpers (host) ----
inline Fr fr_pow2(unsigned k) {  // 2^k, k < 254
    Fr v = jolt::fr_from_u64(1ull << (k & 31));
    const Fr two32 = jolt::fr_from_u64(1ull << 32);
    for (unsigned i = 0; i < k / 32; ++i) v = jolt::mul(v, two32);
    return v;
}
inline Fr fr_from_u128(unsigned __int128 v) { return jolt::add(jolt::mul(jolt::fr_from_u64((uint64_t)(v >> 64)), fr_pow2(64)), jolt::fr_from_u64((uint64_t)v)); }
inline Fr fr_scale_u64(const Fr& a, uint64_t k) { return jolt::mul(a, jolt::fr_from_u64(k)); }

// 2 - 2^64: what the divisor of (MIN, -1) is replaced by, relative to the divisor itself (prefixes/change_divisor.rs:10-12)
inline Fr change_divisor_adjustment() { return jolt::sub(jolt::fr_from_u64(2), fr_pow2(kXlen)); }

inline Fr prefix_default_checkpoint(unsigned prefix) {
    switch (prefix) {
        case kPreEq: case kPreLeftOperandIsZero: case kPreRightOperandIsZero: case kPreDivByZero: case kPrePositiveRemainderEqualsDivisor:
        case kPreNegativeDivisorZeroRemainder: case kPreNegativeDivisorEqualsRemainder: case kPreLsb: case kPrePow2: case kPrePow2W: case kPreLeftShiftHelper:
        case kPreTwoLsb: case kPreSignExtensionUpperHalf: case kPreLeftShiftWHelper: case kPreOverflowBitsZero: case kPrePow2OffsetW:
            return Fr::one();  // products over the bound bits start at one
        case kPreChangeDivisor: return change_divisor_adjustment();
        default: return Fr::zero();  // sums over the bound bits start at zero
    }
}

struct Chunk {  // one phase's bits, de-interleaved
    uint32_t b, len, x, y, x_len, y_len;
    uint32_t suffix_len, j_start;  // bits below the chunk; address variables already bound above it
};
inline Chunk make_chunk(uint32_t b, uint32_t b_len, uint32_t suffix_len) {
    Chunk c;
    c.b = b_len >= 32 ? b : (b & ((1u << b_len) - 1));
    c.len = b_len;
    c.x_len = b_len / 2;
    c.y_len = b_len - c.x_len;
    c.y = jolt::compact_even_bits(c.b);
    c.x = jolt::compact_even_bits(c.b >> 1);
    c.suffix_len = suffix_len;
    c.j_start = kLogK - suffix_len - b_len;
    return c;
}
inline uint32_t top_bit_of(uint32_t v, uint32_t len) { return len ? (v >> (len - 1)) & 1 : 0; }
inline bool all_ones(uint32_t v, uint32_t len) { return v == (len >= 32 ? ~0u : ((1u << len) - 1)); }

// The doubling recurrence of the shift tables over the chunk's (x_i, y_i) pairs, most significant pair first: a set mask bit shifts the accumulator up and
// takes the operand bit (prefixes/right_shift.rs:15-30, right_shift_w.rs).
inline Fr right_shift_walk(Fr acc, const Chunk& c) {
    for (uint32_t i = 0; i < c.y_len; ++i) {
        const uint32_t pos = c.y_len - 1 - i;
        if ((c.y >> pos) & 1) acc = jolt::add(jolt::add(acc, acc), jolt::fr_from_u64((c.x >> pos) & 1));
    }
    return acc;
}
// Left-shift part of a rotation: an operand bit under a CLEAR mask bit lands at 2^(top - i), scaled by 2^(set mask bits seen so far) (prefixes/left_shift.rs:14-37)
inline Fr left_shift_walk(Fr acc, Fr scale, const Chunk& c, uint32_t top) {
    for (uint32_t i = 0; i < c.y_len; ++i) {
        const uint32_t pos = c.y_len - 1 - i;
        const uint32_t x_i = (c.x >> pos) & 1, y_i = (c.y >> pos) & 1;
        if (!y_i && x_i) acc = jolt::add(acc, fr_scale_u64(scale, 1ull << ((top - i) & 63)));
        if (y_i) scale = jolt::add(scale, scale);
    }
    return acc;
}
inline uint64_t rotl64(uint64_t v, uint32_t k) { k &= 63; return k ? (v << k) | (v >> (64 - k)) : v; }
inline uint32_t rotl32(uint32_t v, uint32_t k) { k &= 31; return k ? (v << k) | (v >> (32 - k)) : v; }

// Prefixes::evaluate: the prefix polynomial at (bound challenges folded into `cp`, chunk bits b, the suffix still free)
inline Fr prefix_evaluate(unsigned prefix, const Fr* cp, uint32_t b, uint32_t b_len, uint32_t suffix_len) {
    const Chunk c = make_chunk(b, b_len, suffix_len);
    const Fr self = cp[prefix];
    const bool first = c.j_start == 0 && b_len != 0;  // the chunk holds the operands' sign bits
    switch (prefix) {
        // ---- words read off the index: checkpoint + this chunk's bits at their place value ----
        case kPreLowerWord: return c.j_start < (uint32_t)kXlen ? Fr::zero() : jolt::add(self, fr_from_u128((unsigned __int128)c.b << suffix_len));
        case kPreLowerHalfWord: return c.j_start < (uint32_t)(kXlen + kXlen / 2) ? Fr::zero() : jolt::add(self, fr_from_u128((unsigned __int128)c.b << suffix_len));
        case kPreUpperWord:
            if (c.j_start >= (uint32_t)kXlen) return self;
            return jolt::add(self, jolt::fr_from_u64(suffix_len > (uint32_t)kXlen ? (uint64_t)c.b << (suffix_len - kXlen) : (uint64_t)c.b >> (kXlen - suffix_len)));
        case kPreAnd: return jolt::add(self, jolt::fr_from_u64((uint64_t)(c.x & c.y) << (suffix_len / 2)));
        case kPreAndn: return jolt::add(self, jolt::fr_from_u64((uint64_t)(c.x & ~c.y) << (suffix_len / 2)));
        case kPreOr: return jolt::add(self, jolt::fr_from_u64((uint64_t)(c.x | c.y) << (suffix_len / 2)));
        case kPreXor: return jolt::add(self, jolt::fr_from_u64((uint64_t)(c.x ^ c.y) << (suffix_len / 2)));
        case kPreRightOperand: return jolt::add(self, fr_from_u128((unsigned __int128)c.y << (suffix_len / 2)));
        case kPreRightOperandW: return suffix_len < (uint32_t)kXlen ? jolt::add(self, fr_from_u128((unsigned __int128)c.y << (suffix_len / 2))) : self;
        case kPreRev8W:  // the chunk's bits at their place in the low word, bytes swapped within each half (prefixes/rev8w.rs)
            if (suffix_len >= 64) return Fr::zero();
            {
                const uint64_t placed = (uint64_t)c.b << suffix_len;
                return jolt::add(self, jolt::fr_from_u64((uint64_t)jolt::bswap32((uint32_t)placed) + ((uint64_t)jolt::bswap32((uint32_t)(placed >> 32)) << 32)));
            }
        case kPreXorRot16: case kPreXorRot24: case kPreXorRot32: case kPreXorRot63: {  // prefixes/xor_rot.rs: the chunk's XOR bits, rotated to where the table puts them
            static const uint32_t rot[4] = {16, 24, 32, 63};
            const uint32_t r = rot[prefix - kPreXorRot16], half = suffix_len / 2;
            return jolt::add(self, jolt::fr_from_u64(rotl64((uint64_t)(c.x ^ c.y), half >= r ? half - r : kXlen + half - r)));
        }
        case kPreXorRotW7: case kPreXorRotW8: case kPreXorRotW12: case kPreXorRotW16: {  // prefixes/xor_rotw.rs: the same inside the low 32-bit lane
            if (c.j_start < (uint32_t)kXlen) return Fr::zero();
            static const uint32_t rot[4] = {7, 8, 12, 16};
            const uint32_t r = rot[prefix - kPreXorRotW7], half = suffix_len / 2;
            return jolt::add(self, jolt::fr_from_u64(rotl32(c.x ^ c.y, half >= r ? half - r : 32 + half - r)));
        }
        // ---- indicator products: the checkpoint survives while the chunk keeps the condition alive ----
        case kPreEq: return c.x == c.y ? self : Fr::zero();
        case kPreLeftOperandIsZero: return c.x == 0 ? self : Fr::zero();
        case kPreRightOperandIsZero: return c.y == 0 ? self : Fr::zero();
        case kPreDivByZero: return (c.x == 0 && all_ones(c.y, c.y_len)) ? self : Fr::zero();  // (divisor, quotient) = (x, y)
        case kPreOverflowBitsZero: {  // the chunk's share of index bits 64 .. 127 must be zero (prefixes/overflow_bits_zero.rs)
            if (c.j_start >= (uint32_t)(kLogK - kXlen)) return self;
            const uint32_t overflow_bits = suffix_len >= (uint32_t)kXlen ? c.b : c.b >> (kXlen - suffix_len);
            return overflow_bits ? Fr::zero() : self;
        }
        case kPreChangeDivisor: case kPreChangeDivisorW: {  // dividend = 10..0 and divisor = 1..1 (in the low lane for W); change_divisor.rs, change_divisor_w.rs
            const bool w = prefix == kPreChangeDivisorW;
            if (w && c.j_start < (uint32_t)kXlen) return Fr::zero();
            const bool opens = c.j_start == (w ? (uint32_t)kXlen : 0u);  // this chunk holds the dividend's sign bit, which must be SET
            const uint32_t want_x = opens ? (c.x_len ? 1u << (c.x_len - 1) : 0u) : 0u;
            if (c.x != want_x || !all_ones(c.y, c.y_len)) return Fr::zero();
            return (w && opens) ? change_divisor_adjustment() : self;
        }
        case kPrePositiveRemainderEqualsDivisor: case kPreNegativeDivisorEqualsRemainder: {
            if (c.x != c.y) return Fr::zero();
            const uint32_t sign = prefix == kPreNegativeDivisorEqualsRemainder ? 1u : 0u;
            if (first && (top_bit_of(c.x, c.x_len) != sign || top_bit_of(c.y, c.y_len) != sign)) return Fr::zero();
            return self;
        }
        case kPreNegativeDivisorZeroRemainder:
            if (c.x != 0) return Fr::zero();
            if (first && top_bit_of(c.y, c.y_len) != 1) return Fr::zero();
            return self;
        // ---- comparisons: strict part + (equal-so-far checkpoint if this chunk decides) ----
        case kPreLessThan: return c.x < c.y ? jolt::add(self, cp[kPreEq]) : self;
        case kPrePositiveRemainderLessThanDivisor:
            if (first && (top_bit_of(c.x, c.x_len) != 0 || top_bit_of(c.y, c.y_len) != 0)) return Fr::zero();
            return c.x < c.y ? jolt::add(self, cp[kPrePositiveRemainderEqualsDivisor]) : self;
        case kPreNegativeDivisorGreaterThanRemainder:
            if (first && (top_bit_of(c.x, c.x_len) != 1 || top_bit_of(c.y, c.y_len) != 1)) return Fr::zero();
            return c.x > c.y ? jolt::add(self, cp[kPreNegativeDivisorEqualsRemainder]) : self;
        // ---- single bits ----
        case kPreLeftOperandMsb: return c.j_start > 0 ? self : jolt::fr_from_u64(top_bit_of(c.x, c.x_len));
        case kPreRightOperandMsb: return c.j_start > 0 ? self : jolt::fr_from_u64(top_bit_of(c.y, c.y_len));
        case kPreLsb: return suffix_len == 0 ? jolt::fr_from_u64(c.b & 1) : Fr::one();
        case kPreTwoLsb: return suffix_len == 0 ? ((c.b & 3) == 0 ? Fr::one() : Fr::zero()) : self;
        case kPrePow2: return suffix_len != 0 ? Fr::one() : fr_scale_u64(self, 1ull << (c.b & (kXlen - 1)));
        case kPrePow2W: return suffix_len != 0 ? Fr::one() : fr_scale_u64(self, 1ull << (c.b & 31));
        case kPrePow2OffsetW: {  // 2^(32 * index bit 2), wherever bit 2 falls (prefixes/pow2_offset_w.rs)
            if (suffix_len >= 3) return Fr::one();
            if (suffix_len + b_len > 2) return fr_scale_u64(self, 1ull << (32 * ((c.b >> (2 - suffix_len)) & 1)));
            return self;
        }
        case kPreSignExtensionUpperHalf: {  // (2^32 - 1) 2^32 * index bit 31 (prefixes/sign_extension_upper_half.rs)
            if (suffix_len >= (uint32_t)(kXlen / 2)) return Fr::one();
            const uint32_t sign_bit_round = kXlen + kXlen / 2;
            if (c.j_start <= sign_bit_round && sign_bit_round < c.j_start + b_len)
                return fr_scale_u64(fr_from_u128((unsigned __int128)0xFFFFFFFFull << 32), top_bit_of(c.x, c.x_len));
            return self;
        }
        case kPreSignExtensionRightOperand: {  // (2^64 - 2^32) * bit 31 of the right operand (prefixes/sign_extension_right_operand.rs)
            if (suffix_len >= (uint32_t)kXlen) return Fr::one();
            if (c.j_start >= (uint32_t)kXlen + 2) return self;
            return fr_scale_u64(fr_from_u128(((unsigned __int128)1 << kXlen) - ((unsigned __int128)1 << (kXlen / 2))), top_bit_of(c.y, c.y_len));
        }
        // ---- shifts and rotations by a mask operand ----
        case kPreRightShift: return right_shift_walk(self, c);
        case kPreRightShiftW: return c.j_start < (uint32_t)kXlen ? Fr::zero() : right_shift_walk(self, c);
        case kPreLeftShiftHelper: return fr_scale_u64(self, 1ull << jolt::popcount64(c.y));
        case kPreLeftShiftWHelper: return c.j_start < (uint32_t)kXlen ? Fr::one() : fr_scale_u64(self, 1ull << jolt::popcount64(c.y));
        case kPreLeftShift: return left_shift_walk(self, cp[kPreLeftShiftHelper], c, kXlen - 1 - c.j_start / 2);
        case kPreLeftShiftW: return c.j_start < (uint32_t)kXlen ? Fr::zero() : left_shift_walk(self, cp[kPreLeftShiftWHelper], c, kXlen - 1 - c.j_start / 2);
        case kPreSignExtension: {  // sign * sum of 2^i over the CLEAR mask bits i >= 1, i counted from the top (prefixes/sign_extension.rs)
            uint64_t places = 0;
            const uint32_t base = c.j_start / 2;
            for (uint32_t i = first ? 1 : 0; i < c.y_len; ++i)
                if (!((c.y >> (c.y_len - 1 - i)) & 1)) places += 1ull << (base + i);
            if (first) return top_bit_of(c.x, c.x_len) ? jolt::fr_from_u64(places) : Fr::zero();
            return jolt::add(self, fr_scale_u64(cp[kPreLeftOperandMsb], places));
        }
        // ---- sign of a mask window (pext): sigma = x at the mask's top set bit; alive while no mask bit was seen above ----
        case kPreWindowSign: return jolt::add(self, fr_scale_u64(cp[kPreRightOperandIsZero], jolt::window_sign(c.x, c.y)));
        case kPreWindowSignPow2:
            return fr_scale_u64(jolt::add(self, fr_scale_u64(cp[kPreRightOperandIsZero], jolt::window_sign(c.x, c.y))), 1ull << jolt::popcount64(c.y));
        default: return Fr::zero();
    }
}

// ---- the tables as data ----
enum Coef : int8_t { kPlus = 0, kMinus, kOnes64 /* 2^64 - 1 */, kPow64 /* 2^64 */, kMask32 /* 2^32 - 1 */ };
struct Term { int8_t coef, prefix /* -1: none */, suffix /* position in the table's suffix list; -1: none */; };
struct TableDesc {
    const char* name;
    uint8_t n_prefixes, prefixes[4];
    uint8_t n_suffixes, suffixes[5];  // jolt::SuffixKind ids, in LookupTableKind::suffixes() order
    uint8_t n_terms;
    Term terms[6];
};
#define S(k) (uint8_t) jolt::kSuf##k
#define P(k) (uint8_t) kPre##k
#define PP(k) (int8_t) kPre##k
inline const TableDesc* table_descs() {
    using namespace jolt;
    static const TableDesc t[kNumTables] = {
        {"RangeCheck", 1, {P(LowerWord)}, 2, {S(One), S(LowerWord)}, 2, {{kPlus, PP(LowerWord), 0}, {kPlus, -1, 1}}},
        {"RangeCheckAligned", 2, {P(LowerWord), P(Lsb)}, 3, {S(One), S(LowerWord), S(Lsb)}, 3, {{kPlus, PP(LowerWord), 0}, {kPlus, -1, 1}, {kMinus, PP(Lsb), 2}}},
        {"And", 1, {P(And)}, 2, {S(One), S(And)}, 2, {{kPlus, PP(And), 0}, {kPlus, -1, 1}}},
        {"Andn", 1, {P(Andn)}, 2, {S(One), S(AndNot)}, 2, {{kPlus, PP(Andn), 0}, {kPlus, -1, 1}}},
        {"Or", 1, {P(Or)}, 2, {S(One), S(Or)}, 2, {{kPlus, PP(Or), 0}, {kPlus, -1, 1}}},
        {"Xor", 1, {P(Xor)}, 2, {S(One), S(Xor)}, 2, {{kPlus, PP(Xor), 0}, {kPlus, -1, 1}}},
        {"Equal", 1, {P(Eq)}, 1, {S(Eq)}, 1, {{kPlus, PP(Eq), 0}}},
        {"SignedGreaterThanEqual", 4, {P(RightOperandMsb), P(LeftOperandMsb), P(LessThan), P(Eq)}, 2, {S(One), S(LessThan)}, 5,
         {{kPlus, -1, 0}, {kPlus, PP(RightOperandMsb), 0}, {kMinus, PP(LeftOperandMsb), 0}, {kMinus, PP(LessThan), 0}, {kMinus, PP(Eq), 1}}},
        {"UnsignedGreaterThanEqual", 2, {P(LessThan), P(Eq)}, 2, {S(One), S(LessThan)}, 3, {{kPlus, -1, 0}, {kMinus, PP(LessThan), 0}, {kMinus, PP(Eq), 1}}},
        {"NotEqual", 1, {P(Eq)}, 2, {S(One), S(Eq)}, 2, {{kPlus, -1, 0}, {kMinus, PP(Eq), 1}}},
        {"SignedLessThan", 4, {P(LeftOperandMsb), P(RightOperandMsb), P(LessThan), P(Eq)}, 2, {S(One), S(LessThan)}, 4,
         {{kPlus, PP(LeftOperandMsb), 0}, {kMinus, PP(RightOperandMsb), 0}, {kPlus, PP(LessThan), 0}, {kPlus, PP(Eq), 1}}},
        {"UnsignedLessThan", 2, {P(LessThan), P(Eq)}, 2, {S(One), S(LessThan)}, 2, {{kPlus, PP(LessThan), 0}, {kPlus, PP(Eq), 1}}},
        {"SignMask", 1, {P(LeftOperandMsb)}, 1, {S(One)}, 1, {{kOnes64, PP(LeftOperandMsb), 0}}},
        {"UpperWord", 1, {P(UpperWord)}, 2, {S(One), S(UpperWord)}, 2, {{kPlus, PP(UpperWord), 0}, {kPlus, -1, 1}}},
        {"UnsignedLessThanEqual", 2, {P(LessThan), P(Eq)}, 3, {S(One), S(LessThan), S(Eq)}, 3, {{kPlus, PP(LessThan), 0}, {kPlus, PP(Eq), 1}, {kPlus, PP(Eq), 2}}},
        {"ValidUnsignedRemainder", 3, {P(RightOperandIsZero), P(LessThan), P(Eq)}, 3, {S(One), S(LessThan), S(RightOperandIsZero)}, 3,
         {{kPlus, PP(RightOperandIsZero), 2}, {kPlus, PP(LessThan), 0}, {kPlus, PP(Eq), 1}}},
        {"ValidDiv0", 2, {P(LeftOperandIsZero), P(DivByZero)}, 3, {S(One), S(LeftOperandIsZero), S(DivByZero)}, 3,
         {{kPlus, -1, 0}, {kMinus, PP(LeftOperandIsZero), 1}, {kPlus, PP(DivByZero), 2}}},
        {"HalfwordAlignment", 1, {P(Lsb)}, 2, {S(One), S(Lsb)}, 2, {{kPlus, -1, 0}, {kMinus, PP(Lsb), 1}}},
        {"WordAlignment", 1, {P(TwoLsb)}, 1, {S(TwoLsb)}, 1, {{kPlus, PP(TwoLsb), 0}}},
        {"LowerHalfWord", 1, {P(LowerHalfWord)}, 2, {S(One), S(LowerHalfWord)}, 2, {{kPlus, PP(LowerHalfWord), 0}, {kPlus, -1, 1}}},
        {"SignExtendWord", 2, {P(LowerHalfWord), P(SignExtensionUpperHalf)}, 3, {S(One), S(LowerHalfWord), S(SignExtensionUpperHalf)}, 3,
         {{kPlus, PP(LowerHalfWord), 0}, {kPlus, -1, 1}, {kPlus, PP(SignExtensionUpperHalf), 2}}},
        {"Pow2", 1, {P(Pow2)}, 1, {S(Pow2)}, 1, {{kPlus, PP(Pow2), 0}}},
        {"Pow2W", 1, {P(Pow2W)}, 1, {S(Pow2W)}, 1, {{kPlus, PP(Pow2W), 0}}},
        {"ShiftRightBitmask", 1, {P(Pow2)}, 2, {S(One), S(Pow2)}, 2, {{kPow64, -1, 0}, {kMinus, PP(Pow2), 1}}},
        {"VirtualRev8W", 1, {P(Rev8W)}, 2, {S(One), S(Rev8W)}, 2, {{kPlus, PP(Rev8W), 0}, {kPlus, -1, 1}}},
        {"VirtualSRL", 1, {P(RightShift)}, 2, {S(RightShift), S(RightShiftHelper)}, 2, {{kPlus, PP(RightShift), 1}, {kPlus, -1, 0}}},
        {"VirtualSRA", 3, {P(RightShift), P(LeftOperandMsb), P(SignExtension)}, 4, {S(One), S(RightShift), S(RightShiftHelper), S(SignExtension)}, 4,
         {{kPlus, PP(RightShift), 2}, {kPlus, -1, 1}, {kPlus, PP(LeftOperandMsb), 3}, {kPlus, PP(SignExtension), 0}}},
        {"VirtualROTR", 3, {P(RightShift), P(LeftShiftHelper), P(LeftShift)}, 4, {S(RightShiftHelper), S(RightShift), S(LeftShift), S(One)}, 4,
         {{kPlus, PP(RightShift), 0}, {kPlus, -1, 1}, {kPlus, PP(LeftShiftHelper), 2}, {kPlus, PP(LeftShift), 3}}},
        {"VirtualROTRW", 3, {P(RightShiftW), P(LeftShiftWHelper), P(LeftShiftW)}, 4, {S(RightShiftWHelper), S(RightShiftW), S(LeftShiftW), S(One)}, 4,
         {{kPlus, PP(RightShiftW), 0}, {kPlus, -1, 1}, {kPlus, PP(LeftShiftWHelper), 2}, {kPlus, PP(LeftShiftW), 3}}},
        {"VirtualChangeDivisor", 2, {P(RightOperand), P(ChangeDivisor)}, 3, {S(One), S(RightOperand), S(ChangeDivisor)}, 3,
         {{kPlus, PP(RightOperand), 0}, {kPlus, -1, 1}, {kPlus, PP(ChangeDivisor), 2}}},
        {"VirtualChangeDivisorW", 3, {P(RightOperandW), P(ChangeDivisorW), P(SignExtensionRightOperand)}, 4,
         {S(One), S(RightOperandW), S(ChangeDivisorW), S(SignExtensionRightOperand)}, 4,
         {{kPlus, PP(RightOperandW), 0}, {kPlus, -1, 1}, {kPlus, PP(ChangeDivisorW), 2}, {kPlus, PP(SignExtensionRightOperand), 3}}},
        {"MulUNoOverflow", 1, {P(OverflowBitsZero)}, 1, {S(OverflowBitsZero)}, 1, {{kPlus, PP(OverflowBitsZero), 0}}},
        {"VirtualXORROT32", 1, {P(XorRot32)}, 2, {S(One), S(XorRot32)}, 2, {{kPlus, PP(XorRot32), 0}, {kPlus, -1, 1}}},
        {"VirtualXORROT24", 1, {P(XorRot24)}, 2, {S(One), S(XorRot24)}, 2, {{kPlus, PP(XorRot24), 0}, {kPlus, -1, 1}}},
        {"VirtualXORROT16", 1, {P(XorRot16)}, 2, {S(One), S(XorRot16)}, 2, {{kPlus, PP(XorRot16), 0}, {kPlus, -1, 1}}},
        {"VirtualXORROT63", 1, {P(XorRot63)}, 2, {S(One), S(XorRot63)}, 2, {{kPlus, PP(XorRot63), 0}, {kPlus, -1, 1}}},
        {"VirtualXORROTW16", 1, {P(XorRotW16)}, 2, {S(One), S(XorRotW16)}, 2, {{kPlus, PP(XorRotW16), 0}, {kPlus, -1, 1}}},
        {"VirtualXORROTW12", 1, {P(XorRotW12)}, 2, {S(One), S(XorRotW12)}, 2, {{kPlus, PP(XorRotW12), 0}, {kPlus, -1, 1}}},
        {"VirtualXORROTW8", 1, {P(XorRotW8)}, 2, {S(One), S(XorRotW8)}, 2, {{kPlus, PP(XorRotW8), 0}, {kPlus, -1, 1}}},
        {"VirtualXORROTW7", 1, {P(XorRotW7)}, 2, {S(One), S(XorRotW7)}, 2, {{kPlus, PP(XorRotW7), 0}, {kPlus, -1, 1}}},
        {"WindowMaskW", 1, {P(Pow2OffsetW)}, 1, {S(Pow2OffsetW)}, 1, {{kMask32, PP(Pow2OffsetW), 0}}},
        {"PextSigned", 4, {P(RightShift), P(RightOperandIsZero), P(WindowSign), P(WindowSignPow2)}, 5, {S(One), S(Pext), S(PextHelper), S(WindowSign), S(WindowSignPow2)}, 6,
         {{kPlus, PP(RightShift), 2}, {kPlus, -1, 1}, {kPow64, PP(WindowSign), 0}, {kPow64, PP(RightOperandIsZero), 3}, {kMinus, PP(WindowSignPow2), 2},
          {kMinus, PP(RightOperandIsZero), 4}}},
    };
    return t;
}
#undef S
#undef P
#undef PP

// PrefixSuffixDecomposition::combine: prefixes indexed by discriminant (all 49 slots), suffixes in the table's own order
inline Fr table_combine(const TableDesc& t, const Fr* prefixes, const Fr* suffixes) {
    Fr acc = Fr::zero();
    for (uint32_t k = 0; k < t.n_terms; ++k) {
        const Term& term = t.terms[k];
        Fr v = term.prefix >= 0 ? prefixes[term.prefix] : Fr::one();
        if (term.suffix >= 0) v = term.prefix >= 0 ? jolt::mul(v, suffixes[term.suffix]) : suffixes[term.suffix];
        switch (term.coef) {
            case kPlus: acc = jolt::add(acc, v); break;
            case kMinus: acc = jolt::sub(acc, v); break;
            case kOnes64: acc = jolt::add(acc, fr_scale_u64(v, ~0ull)); break;
            case kPow64: acc = jolt::add(acc, jolt::mul(v, fr_pow2(64))); break;
            default: acc = jolt::add(acc, fr_scale_u64(v, 0xFFFFFFFFull)); break;
        }
    }
    return acc;
}

}  // namespace jolt_lookup
