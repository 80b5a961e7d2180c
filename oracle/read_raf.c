/* oracle/read_raf.c -- CPU restatement (TEST INFRASTRUCTURE) of the T-scale scans of instruction read+RAF checking:
 *   crates/jolt-kernels/src/optimized/instruction_read_raf.rs  init_phase (:747-900), init_suffix_tables (:901-971),
 *   pending_combined_base / pending_ra_base (:1203-1232)
 * and of the suffix polynomials they evaluate: crates/jolt-lookup-tables/src/tables/suffixes/ (one file per suffix) through Suffixes::suffix_mle
 * (mod.rs:200-252), LookupBits (lookup_bits.rs) and uninterleave_bits (interleave.rs:38-58).  XLEN = 64.
 * Written with 128-bit integers and bit loops, independently of the device's limb / mask code.  Parity unpinned by vectors (the
 * reference holds none); tests/test_oracle_read_raf.py checks the suffixes against a Python big-integer model of the same source text
 * and the scan against brute-force sums. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "fr.h"

#define EXPORT __attribute__((visibility("default")))
typedef unsigned __int128 u128;

#define XLEN 64
#define CHUNK_LEN 8
#define CHUNK_SIZE 256

typedef struct { u128 bits; unsigned len; } lookup_bits;               /* lookup_bits.rs:6-24 */
static lookup_bits lb_new(u128 bits, unsigned len) {
    lookup_bits b;
    b.bits = len < 128 ? bits % ((u128)1 << len) : bits;
    b.len = len;
    return b;
}
static void lb_uninterleave(lookup_bits b, lookup_bits *x, lookup_bits *y) { /* x = odd positions, y = even positions; lookup_bits.rs:26-31 */
    u128 xb = 0, yb = 0;
    for (unsigned k = 0; k < 64; ++k) {
        yb |= ((b.bits >> (2 * k)) & 1) << k;
        xb |= ((b.bits >> (2 * k + 1)) & 1) << k;
    }
    *x = lb_new(xb, b.len / 2);
    *y = lb_new(yb, b.len - b.len / 2);
}
static unsigned lb_trailing_zeros(lookup_bits b) { /* min(u128 trailing zeros, len), lookup_bits.rs:59-64 */
    unsigned n = 0;
    while (n < 128 && !((b.bits >> n) & 1)) ++n;
    return n < b.len ? n : b.len;
}
static unsigned lb_leading_ones(lookup_bits b) { /* bits.wrapping_shl(128 - len).leading_ones(), lookup_bits.rs:66-70 */
    unsigned n = 0;
    while (n < b.len && ((b.bits >> (b.len - 1 - n)) & 1)) ++n;
    return n;
}
static unsigned tz_u64(uint64_t v) { unsigned n = 0; if (!v) return 64; while (!((v >> n) & 1)) ++n; return n; }
static unsigned ones_u64(uint64_t v) { unsigned n = 0; for (unsigned k = 0; k < 64; ++k) n += (unsigned)((v >> k) & 1); return n; }
static uint64_t shl_u64(uint64_t v, unsigned k) { return k >= 64 ? 0 : v << k; }       /* u64::unbounded_shl */
static uint64_t shr_u64(uint64_t v, unsigned k) { return k >= 64 ? 0 : v >> k; }
static uint32_t shl_u32(uint32_t v, unsigned k) { return k >= 32 ? 0 : v << k; }
static uint32_t shr_u32(uint32_t v, unsigned k) { return k >= 32 ? 0 : v >> k; }
static uint64_t ror_u64(uint64_t v, unsigned k) { k %= 64; return k ? (v >> k) | (v << (64 - k)) : v; }
static uint32_t ror_u32(uint32_t v, unsigned k) { k %= 32; return k ? (v >> k) | (v << (32 - k)) : v; }
static uint32_t swap_bytes_u32(uint32_t v) { return ((v & 0xFFu) << 24) | ((v & 0xFF00u) << 8) | ((v >> 8) & 0xFF00u) | (v >> 24); }
static uint64_t pext_u64(uint64_t x, uint64_t y) { /* suffixes/pext.rs:12-35 (general branch; the contiguous-mask branch is the same function) */
    uint64_t out = 0;
    unsigned k = 0;
    for (unsigned pos = 0; pos < 64; ++pos)
        if ((y >> pos) & 1) out |= ((x >> pos) & 1) << k++;
    return out;
}
static uint64_t window_sign_bit(uint64_t x, uint64_t y) { /* suffixes/window_sign.rs:9-16 */
    if (y == 0) return 0;
    unsigned lg = 63;
    while (!((y >> lg) & 1)) --lg;
    return (x >> lg) & 1;
}

/* enum Suffixes discriminants, mod.rs:120-170 */
enum { S_One, S_And, S_AndNot, S_Xor, S_Or, S_RightOperand, S_RightOperandW, S_ChangeDivisor, S_ChangeDivisorW, S_UpperWord, S_LowerWord, S_LowerHalfWord,
       S_LessThan, S_GreaterThan, S_Eq, S_LeftOperandIsZero, S_RightOperandIsZero, S_Lsb, S_DivByZero, S_Pow2, S_Pow2W, S_Rev8W, S_RightShiftPadding, S_RightShift,
       S_RightShiftHelper, S_SignExtension, S_LeftShift, S_TwoLsb, S_SignExtensionUpperHalf, S_SignExtensionRightOperand, S_RightShiftW, S_RightShiftWHelper,
       S_LeftShiftWHelper, S_LeftShiftW, S_OverflowBitsZero, S_XorRot16, S_XorRot24, S_XorRot32, S_XorRot63, S_XorRotW16, S_XorRotW12, S_XorRotW8, S_XorRotW7,
       S_Pow2OffsetW, S_Pext, S_PextHelper, S_WindowSign, S_WindowSignPow2, S_COUNT };

static uint64_t suffix_mle(unsigned kind, lookup_bits b) {
    lookup_bits xb, yb;
    lb_uninterleave(b, &xb, &yb);
    const uint64_t x = (uint64_t)xb.bits, y = (uint64_t)yb.bits;
    switch (kind) {
        case S_One: return 1;                                                               /* one.rs */
        case S_And: return x & y;                                                           /* and.rs */
        case S_AndNot: return x & ~y;                                                       /* andnot.rs */
        case S_Xor: return x ^ y;                                                           /* xor.rs */
        case S_Or: return x | y;                                                            /* or.rs */
        case S_RightOperand: return y;                                                      /* right_operand.rs */
        case S_RightOperandW: return (uint32_t)y;                                           /* right_operand_w.rs */
        case S_ChangeDivisor: return (shl_u64(1, yb.len) - 1 == y) && x == 0;               /* change_divisor.rs */
        case S_ChangeDivisorW: {                                                            /* change_divisor_w.rs */
            const unsigned y_len = yb.len < XLEN / 2 ? yb.len : XLEN / 2;
            return (((uint64_t)1 << y_len) - 1 == (uint64_t)(uint32_t)y) && (uint32_t)x == 0;
        }
        case S_UpperWord: return (uint64_t)(b.bits >> XLEN);                                /* upper_word.rs */
        case S_LowerWord: return (uint64_t)b.bits;                                          /* lower_word.rs: bits % 2^64 */
        case S_LowerHalfWord: return (uint64_t)(b.bits % ((u128)1 << (XLEN / 2)));          /* lower_half_word.rs */
        case S_LessThan: return x < y;                                                      /* lt.rs */
        case S_GreaterThan: return x > y;                                                   /* gt.rs */
        case S_Eq: return xb.bits == yb.bits;                                               /* eq.rs */
        case S_LeftOperandIsZero: return x == 0;                                            /* left_is_zero.rs */
        case S_RightOperandIsZero: return y == 0;                                           /* right_is_zero.rs */
        case S_Lsb: return b.len == 0 ? 1 : (uint64_t)(b.bits & 1);                         /* lsb.rs */
        case S_DivByZero: return x == 0 && y == shl_u64(1, yb.len) - 1;                     /* div_by_zero.rs: (divisor, quotient) */
        case S_Pow2: return b.len == 0 ? 1 : (uint64_t)1 << (unsigned)(b.bits % 64);        /* pow2.rs: split(log2 XLEN) */
        case S_Pow2W: return b.len == 0 ? 1 : (uint64_t)1 << (unsigned)(b.bits % 32);       /* pow2_w.rs: split(5) */
        case S_Rev8W: {                                                                     /* rev8w.rs -> virtual_rev8w.rs:14-18 */
            const uint64_t v = (uint64_t)b.bits;
            return (uint64_t)swap_bytes_u32((uint32_t)v) + ((uint64_t)swap_bytes_u32((uint32_t)(v >> 32)) << 32);
        }
        case S_RightShiftPadding: return b.len == 0 ? 1 : (uint64_t)1 << (XLEN - 1 - (unsigned)(b.bits % 64)); /* right_shift_padding.rs */
        case S_RightShift: return shr_u64(x, lb_trailing_zeros(yb));                        /* right_shift.rs */
        case S_RightShiftHelper: return shl_u64(1, lb_leading_ones(yb));                    /* right_shift_helper.rs */
        case S_SignExtension: {                                                             /* sign_extension.rs */
            const unsigned tz = tz_u64(y), padding_len = tz < yb.len ? tz : yb.len;
            return (uint64_t)(((u128)1 << XLEN) - ((u128)1 << (XLEN - padding_len)));
        }
        case S_LeftShift: return shl_u64(x & ~y, lb_leading_ones(yb));                      /* left_shift.rs */
        case S_TwoLsb: {                                                                    /* two_lsb.rs: trailing_zeros(u128) >= 2 */
            if (b.len == 0) return 1;
            return (b.bits & 3) == 0;
        }
        case S_SignExtensionUpperHalf:                                                      /* sign_extension_upper_half.rs */
            if (b.len >= XLEN / 2) return ((b.bits >> (XLEN / 2 - 1)) & 1) ? (((uint64_t)1 << (XLEN / 2)) - 1) << (XLEN / 2) : 0;
            return 1;
        case S_SignExtensionRightOperand:                                                   /* sign_extension_right_operand.rs */
            if (b.len >= XLEN) return ((b.bits >> (XLEN - 2)) & 1) ? (uint64_t)(((u128)1 << XLEN) - ((u128)1 << (XLEN / 2))) : 0;
            return 1;
        case S_RightShiftW: {                                                               /* right_shift_w.rs */
            unsigned tz = lb_trailing_zeros(yb);
            if (tz > XLEN / 2) tz = XLEN / 2;
            return shr_u32((uint32_t)x, tz);
        }
        case S_RightShiftWHelper: {                                                         /* right_shift_w_helper.rs */
            const lookup_bits yw = lb_new(yb.bits, yb.len < XLEN / 2 ? yb.len : XLEN / 2);
            return shl_u64(1, lb_leading_ones(yw));
        }
        case S_LeftShiftWHelper: return (uint64_t)((uint32_t)1 << (lb_leading_ones(yb) % 32)); /* left_shift_w_helper.rs: 1u32 << k, release-mode shift */
        case S_LeftShiftW: {                                                                /* left_shift_w.rs */
            const lookup_bits yw = lb_new(yb.bits, yb.len < XLEN / 2 ? yb.len : XLEN / 2);
            return shl_u32((uint32_t)x & ~(uint32_t)yw.bits, lb_leading_ones(yw));
        }
        case S_OverflowBitsZero: return (b.bits >> XLEN) == 0;                              /* overflow_bits_zero.rs */
        case S_XorRot16: return ror_u64(x ^ y, 16);                                         /* xor_rot.rs */
        case S_XorRot24: return ror_u64(x ^ y, 24);
        case S_XorRot32: return ror_u64(x ^ y, 32);
        case S_XorRot63: return ror_u64(x ^ y, 63);
        case S_XorRotW16: return ror_u32((uint32_t)x ^ (uint32_t)y, 16);                    /* xor_rotw.rs */
        case S_XorRotW12: return ror_u32((uint32_t)x ^ (uint32_t)y, 12);
        case S_XorRotW8: return ror_u32((uint32_t)x ^ (uint32_t)y, 8);
        case S_XorRotW7: return ror_u32((uint32_t)x ^ (uint32_t)y, 7);
        case S_Pow2OffsetW: return b.len < 3 ? 1 : (uint64_t)1 << (32 * (unsigned)((b.bits >> 2) & 1)); /* pow2_offset_w.rs */
        case S_Pext: return pext_u64(x, y);                                                 /* pext.rs */
        case S_PextHelper: return shl_u64(1, ones_u64(y));                                  /* pext_helper.rs */
        case S_WindowSign: return window_sign_bit(x, y);                                    /* window_sign.rs */
        case S_WindowSignPow2: return shl_u64(window_sign_bit(x, y), ones_u64(y));          /* window_sign_pow2.rs */
        default: return 0;
    }
}
EXPORT uint64_t orc_suffix_mle(uint32_t kind, uint64_t lo, uint64_t hi, uint32_t len) { return suffix_mle(kind, lb_new(((u128)hi << 64) | lo, len)); }
EXPORT int32_t orc_suffix_is_01_valued(uint32_t kind) { /* mod.rs:181-198 */
    switch (kind) {
        case S_One: case S_Eq: case S_LessThan: case S_GreaterThan: case S_LeftOperandIsZero: case S_RightOperandIsZero: case S_Lsb: case S_TwoLsb:
        case S_DivByZero: case S_OverflowBitsZero: case S_ChangeDivisor: case S_ChangeDivisorW: case S_WindowSign: return 1;
        default: return 0;
    }
}
EXPORT void orc_uninterleave(uint64_t lo, uint64_t hi, uint64_t *x, uint64_t *y) {
    lookup_bits xb, yb;
    lb_uninterleave(lb_new(((u128)hi << 64) | lo, 128), &xb, &yb);
    *x = (uint64_t)xb.bits;
    *y = (uint64_t)yb.bits;
}

static u128 row_index(const uint64_t *lookup_index, size_t j) { return ((u128)lookup_index[2 * j + 1] << 64) | lookup_index[2 * j]; }

/* One phase's scans.  raf_out: [left, right, identity, shift_half, shift_full, upper_all_ones][256] (raw sums: the caller applies
 * mul_pow_2 to the shift sums, :814-823); suffix_out[(suffix_offsets[t] + s) * 256 + chunk]. */
EXPORT void orc_read_raf_phase_scan(const uint64_t *lookup_index, const uint8_t *table_index, const uint8_t *raf_flag, size_t cycles, uint32_t n_tables, const fr_t *u,
                                    uint32_t suffix_len, uint32_t address_bits, int canonical, const uint32_t *suffix_offsets, const uint8_t *suffix_kinds,
                                    fr_t *raf_out, fr_t *suffix_out) {
    const u128 suffix_mask = suffix_len == 128 ? ~(u128)0 : (((u128)1 << suffix_len) - 1);
    const uint32_t upper_suffix_bits = suffix_len > address_bits / 2 ? suffix_len - address_bits / 2 : 0;
    for (size_t k = 0; k < 6 * CHUNK_SIZE; ++k) raf_out[k] = fr_zero();
    for (size_t k = 0; k < (size_t)suffix_offsets[n_tables] * CHUNK_SIZE; ++k) suffix_out[k] = fr_zero();
    for (size_t j = 0; j < cycles; ++j) {
        const u128 index = row_index(lookup_index, j);
        const unsigned chunk = (unsigned)((suffix_len >= 128 ? 0 : index >> suffix_len) & (CHUNK_SIZE - 1));
        const u128 suffix_bits = index & suffix_mask;
        const fr_t uj = u[j];
        if (canonical && raf_flag[j] &&
            (upper_suffix_bits == 0 || (suffix_bits >> (suffix_len - upper_suffix_bits)) == (((u128)1 << upper_suffix_bits) - 1)))
            raf_out[5 * CHUNK_SIZE + chunk] = FADD(raf_out[5 * CHUNK_SIZE + chunk], uj);
        if (!raf_flag[j]) {
            raf_out[3 * CHUNK_SIZE + chunk] = FADD(raf_out[3 * CHUNK_SIZE + chunk], uj);
            lookup_bits l, r;
            lb_uninterleave(lb_new(suffix_bits, suffix_len), &l, &r);
            if ((uint64_t)l.bits) raf_out[0 * CHUNK_SIZE + chunk] = FADD(raf_out[0 * CHUNK_SIZE + chunk], fr_mul_u64(uj, (uint64_t)l.bits));
            if ((uint64_t)r.bits) raf_out[1 * CHUNK_SIZE + chunk] = FADD(raf_out[1 * CHUNK_SIZE + chunk], fr_mul_u64(uj, (uint64_t)r.bits));
        } else {
            raf_out[4 * CHUNK_SIZE + chunk] = FADD(raf_out[4 * CHUNK_SIZE + chunk], uj);
            if (suffix_bits) raf_out[2 * CHUNK_SIZE + chunk] = FADD(raf_out[2 * CHUNK_SIZE + chunk], fr_mul_u128(uj, (uint64_t)suffix_bits, (uint64_t)(suffix_bits >> 64)));
        }
        if (table_index[j] != 0xFF) { /* init_suffix_tables: only the row's own table */
            const uint32_t t = table_index[j];
            const lookup_bits sb = lb_new(suffix_bits, suffix_len);
            for (uint32_t s = suffix_offsets[t]; s < suffix_offsets[t + 1]; ++s) {
                const uint64_t value = suffix_mle(suffix_kinds[s], sb);
                if (value) suffix_out[(size_t)s * CHUNK_SIZE + chunk] = FADD(suffix_out[(size_t)s * CHUNK_SIZE + chunk], fr_mul_u64(uj, value));
            }
        }
    }
}

/* condensation (:750-758) */
EXPORT void orc_read_raf_condense(const uint64_t *lookup_index, size_t cycles, const fr_t *v /* 256 */, uint32_t shift, fr_t *u) {
    for (size_t j = 0; j < cycles; ++j) u[j] = FMUL(u[j], v[(unsigned)((row_index(lookup_index, j) >> shift) & (CHUNK_SIZE - 1))]);
}

/* pending_combined_base / pending_ra_base (:1203-1232) for every cycle */
EXPORT void orc_read_raf_cycle_tables(const uint64_t *lookup_index, const uint8_t *table_index, const uint8_t *raf_flag, size_t cycles, const fr_t *table_values,
                                      const fr_t *raf_interleaved, const fr_t *raf_identity, const fr_t *v_tables /* phases * 256 */, uint32_t phases, uint32_t address_bits,
                                      uint32_t ra_count, fr_t *combined, fr_t *ra /* ra_count * cycles */) {
    const uint32_t phases_per_ra = phases / ra_count;
    for (size_t j = 0; j < cycles; ++j) {
        const fr_t table_value = table_index[j] == 0xFF ? fr_zero() : table_values[table_index[j]];
        combined[j] = FADD(table_value, raf_flag[j] ? *raf_identity : *raf_interleaved);
        const u128 index = row_index(lookup_index, j);
        for (uint32_t i = 0; i < ra_count; ++i) {
            uint32_t phase = i * phases_per_ra;
            uint32_t shift = address_bits - (phase + 1) * CHUNK_LEN;
            fr_t product = v_tables[(size_t)phase * CHUNK_SIZE + ((unsigned)(index >> shift) & (CHUNK_SIZE - 1))];
            for (uint32_t p = 1; p < phases_per_ra; ++p) {
                phase += 1;
                shift -= CHUNK_LEN;
                product = FMUL(product, v_tables[(size_t)phase * CHUNK_SIZE + ((unsigned)(index >> shift) & (CHUNK_SIZE - 1))]);
            }
            ra[(size_t)i * cycles + j] = product;
        }
    }
}
