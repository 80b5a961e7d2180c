/* oracle/lookup_tables.c -- CPU restatement (TEST INFRASTRUCTURE, never linked into or called by the product) of the lookup tables the
 * instruction read+RAF relation reads, as FULL multilinear extensions, and of the address rounds computed from them directly:
 *   crates/jolt-lookup-tables/src/tables/mod.rs:121-166      enum LookupTableKind (the table id IS the discriminant, XLEN = 64)
 *   crates/jolt-lookup-tables/src/tables/<table>.rs          LookupTable::materialize_entry / evaluate_mle of each of the 42 tables
 *   crates/jolt-kernels/src/optimized/instruction_read_raf.rs:1545-1575   input_claim "from first principles"
 * The product (jolt_amd/csrc/read_raf_address.hip) computes the 128 address-round polynomials the way the reference kernel does: 8-variable
 * phases over 256-entry prefix polynomials (tables/prefixes/), per-table `combine` and the suffix accumulators of the T-scale scans.
 * This file deliberately shares NONE of that: a round polynomial here is the definition
 *     s_i(c) = sum_j eq(r_reduction, j) * eq(r_{<i}, k_j[<i]) * eq(c, k_j[i]) * F_j(r_{<i}, c, k_j[>i]),
 *     F_j    = Val_{table(j)} + (raf_flag_j ? gamma^2 Identity (+ gamma^3 UpperAllOnes, `akita`) : gamma Left + gamma^2 Right)
 * with Val_t evaluated by the table's own evaluate_mle at the mixed point -- the identity the reference's prefix_suffix_test
 * (tables/test_utils.rs:100-200) asserts between `combine(prefixes, suffixes)` and `evaluate_mle`.
 * Pins: evaluate_mle on Boolean points == materialize_entry (the reference's mle_random_test / mle_full_hypercube_test,
 * tests/test_oracle_lookup_tables.py); s_0(0) + s_0(1) == the first-principles input claim.  Parity unpinned by vectors (the reference
 * holds none for this layer; no Rust toolchain here). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "fr.h"
#ifdef _OPENMP
#include <omp.h>
#endif

#define EXPORT __attribute__((visibility("default")))
typedef unsigned __int128 u128;
#define XLEN 64
#define LOG_K 128

enum { T_RangeCheck, T_RangeCheckAligned, T_And, T_Andn, T_Or, T_Xor, T_Equal, T_SignedGreaterThanEqual, T_UnsignedGreaterThanEqual, T_NotEqual,
       T_SignedLessThan, T_UnsignedLessThan, T_SignMask, T_UpperWord, T_UnsignedLessThanEqual, T_ValidUnsignedRemainder, T_ValidDiv0, T_HalfwordAlignment,
       T_WordAlignment, T_LowerHalfWord, T_SignExtendWord, T_Pow2, T_Pow2W, T_ShiftRightBitmask, T_VirtualRev8W, T_VirtualSRL, T_VirtualSRA, T_VirtualROTR,
       T_VirtualROTRW, T_VirtualChangeDivisor, T_VirtualChangeDivisorW, T_MulUNoOverflow, T_VirtualXORROT32, T_VirtualXORROT24, T_VirtualXORROT16,
       T_VirtualXORROT63, T_VirtualXORROTW16, T_VirtualXORROTW12, T_VirtualXORROTW8, T_VirtualXORROTW7, T_WindowMaskW, T_PextSigned, T_COUNT };

/* ---- integer side: materialize_entry ---- */
static void split_operands(u128 index, uint64_t *x, uint64_t *y) { /* uninterleave_bits, interleave.rs:38-58: x from the odd positions, y from the even ones */
    uint64_t xb = 0, yb = 0;
    for (unsigned k = 0; k < 64; ++k) {
        yb |= (uint64_t)((index >> (2 * k)) & 1) << k;
        xb |= (uint64_t)((index >> (2 * k + 1)) & 1) << k;
    }
    *x = xb;
    *y = yb;
}
static uint32_t swap32(uint32_t v) { return ((v & 0xFFu) << 24) | ((v & 0xFF00u) << 8) | ((v >> 8) & 0xFF00u) | (v >> 24); }
static unsigned popcount_u64(uint64_t v) { unsigned n = 0; for (unsigned k = 0; k < 64; ++k) n += (unsigned)((v >> k) & 1); return n; }
static uint64_t rot_right(uint64_t v, unsigned r, unsigned width) { /* ((v >> r) | (v << (width - r))) & mask, virtual_xor_rot.rs:20-26 */
    const u128 wide = v, mask = ((u128)1 << width) - 1;
    return (uint64_t)(((wide >> r) | (wide << (width - r))) & mask);
}
static uint64_t xor_rot(u128 index, unsigned rotation) { uint64_t x, y; split_operands(index, &x, &y); return rot_right(x ^ y, rotation % 64, 64); }
static uint64_t xor_rotw(u128 index, unsigned rotation) { /* virtual_xor_rotw.rs:16-24 */
    uint64_t x, y;
    split_operands(index, &x, &y);
    return rot_right((x ^ y) & 0xFFFFFFFFull, rotation % 32, 32);
}

static uint64_t materialize_entry(unsigned kind, u128 index) {
    uint64_t x, y;
    split_operands(index, &x, &y);
    const uint64_t low = (uint64_t)index, high = (uint64_t)(index >> 64);
    switch (kind) {
        case T_RangeCheck: return low;                                                /* range_check.rs:16-18 */
        case T_RangeCheckAligned: return low & ~(uint64_t)1;                          /* range_check_aligned.rs */
        case T_And: return x & y;
        case T_Andn: return x & ~y;
        case T_Or: return x | y;
        case T_Xor: return x ^ y;
        case T_Equal: return x == y;
        case T_SignedGreaterThanEqual: return (int64_t)x >= (int64_t)y;               /* signed_greater_than_equal.rs:15-22 (XLEN = 64: the shift is 0) */
        case T_UnsignedGreaterThanEqual: return x >= y;
        case T_NotEqual: return x != y;
        case T_SignedLessThan: return (int64_t)x < (int64_t)y;
        case T_UnsignedLessThan: return x < y;
        case T_SignMask: return (index >> 127) & 1 ? ~(uint64_t)0 : 0;                /* sign_mask.rs:16-25 */
        case T_UpperWord: return high;
        case T_UnsignedLessThanEqual: return x <= y;
        case T_ValidUnsignedRemainder: return y == 0 || x < y;                        /* (remainder, divisor) = (x, y) */
        case T_ValidDiv0: return x == 0 ? (y == ~(uint64_t)0) : 1;                    /* (divisor, quotient) = (x, y) */
        case T_HalfwordAlignment: return (index & 1) == 0;
        case T_WordAlignment: return (index & 3) == 0;
        case T_LowerHalfWord: return low & 0xFFFFFFFFull;
        case T_SignExtendWord: {                                                      /* sign_extend_word.rs:17-27 */
            const uint64_t lower_half = low & 0xFFFFFFFFull;
            return (lower_half >> 31) & 1 ? lower_half | 0xFFFFFFFF00000000ull : lower_half;
        }
        case T_Pow2: return (uint64_t)1 << (unsigned)(index % 64);
        case T_Pow2W: return (uint64_t)1 << (unsigned)(index % 32);
        case T_ShiftRightBitmask: {                                                   /* shift_right_bitmask.rs:15-19 */
            const unsigned shift = (unsigned)(index % 64);
            return (uint64_t)((((u128)1 << (64 - shift)) - 1) << shift);
        }
        case T_VirtualRev8W: return (uint64_t)swap32((uint32_t)low) + ((uint64_t)swap32((uint32_t)(low >> 32)) << 32); /* virtual_rev8w.rs:12-17 */
        case T_VirtualSRL: case T_VirtualSRA: {                                       /* virtual_srl.rs:16-29, virtual_sra.rs:16-34 */
            uint64_t entry = 0, sign_extension = 0;
            for (unsigned i = 0; i < 64; ++i) {
                const uint64_t x_i = (x >> (63 - i)) & 1, y_i = (y >> (63 - i)) & 1;
                entry *= 1 + y_i;
                entry += x_i * y_i;
                if (i != 0) sign_extension += ((uint64_t)1 << i) * (1 - y_i);
            }
            return kind == T_VirtualSRL ? entry : entry + (x >> 63) * sign_extension;
        }
        case T_VirtualROTR: case T_VirtualROTRW: {                                    /* virtual_rotr.rs:16-31, virtual_rotrw.rs:16-30 (the W table walks the low 32 pairs only) */
            u128 prod_one_plus_y = 1;
            uint64_t first_sum = 0, second_sum = 0;
            for (int i = kind == T_VirtualROTR ? 63 : 31; i >= 0; --i) {
                const uint64_t xb = (x >> i) & 1, yb = (y >> i) & 1;
                first_sum *= 1 + yb;
                first_sum += xb * yb;
                second_sum += xb * (uint64_t)((1 - (u128)yb) * prod_one_plus_y) * ((uint64_t)1 << i);
                prod_one_plus_y *= 1 + (u128)yb;
            }
            return first_sum + second_sum;
        }
        case T_VirtualChangeDivisor: return (x == (uint64_t)1 << 63 && y == ~(uint64_t)0) ? 1 : y;  /* (dividend, divisor) = (x, y) */
        case T_VirtualChangeDivisorW: {                                               /* virtual_change_divisor_w.rs:16-38 */
            const uint64_t dividend_lo = x & 0xFFFFFFFFull, divisor_lo = y & 0xFFFFFFFFull;
            if (dividend_lo == (uint64_t)1 << 31 && divisor_lo == 0xFFFFFFFFull) return 1;
            return (divisor_lo >> 31) & 1 ? divisor_lo | 0xFFFFFFFF00000000ull : divisor_lo;
        }
        case T_MulUNoOverflow: return high == 0;
        case T_VirtualXORROT32: return xor_rot(index, 32);
        case T_VirtualXORROT24: return xor_rot(index, 24);
        case T_VirtualXORROT16: return xor_rot(index, 16);
        case T_VirtualXORROT63: return xor_rot(index, 63);
        case T_VirtualXORROTW16: return xor_rotw(index, 16);
        case T_VirtualXORROTW12: return xor_rotw(index, 12);
        case T_VirtualXORROTW8: return xor_rotw(index, 8);
        case T_VirtualXORROTW7: return xor_rotw(index, 7);
        case T_WindowMaskW: return 0xFFFFFFFFull << (32 * (unsigned)((index >> 2) & 1));  /* window_mask_w.rs:21-26 */
        case T_PextSigned: {                                                          /* pext_signed.rs:29-43 */
            const unsigned pc = popcount_u64(y);
            if (pc == 0) return 0;
            uint64_t pext = 0;
            unsigned k = 0, top = 0;
            for (unsigned pos = 0; pos < 64; ++pos)
                if ((y >> pos) & 1) { pext |= ((x >> pos) & 1) << k++; top = pos; }
            const uint64_t sign = (x >> top) & 1;
            return pext + (sign ? (uint64_t)(((u128)1 << 64) - ((u128)1 << pc)) : 0);
        }
        default: return 0;
    }
}
EXPORT uint64_t orc_table_materialize_entry(uint32_t kind, uint64_t lo, uint64_t hi) { return materialize_entry(kind, ((u128)hi << 64) | lo); }
EXPORT uint32_t orc_table_count(void) { return T_COUNT; }

/* ---- field side: evaluate_mle at r[0 .. 128), r[0] the variable of index bit 127 ---- */
static fr_t ONE(void) { return fr_one(); }
static fr_t pow2_fr(unsigned k) { return k < 64 ? fr_from_u64((uint64_t)1 << k) : fr_from_u128(0, (uint64_t)1 << (k - 64)); }
static fr_t one_minus(fr_t a) { return FSUB(ONE(), a); }
static fr_t eq_pair(fr_t x, fr_t y) { return FADD(FMUL(x, y), FMUL(one_minus(x), one_minus(y))); }       /* x y + (1 - x)(1 - y) */
static fr_t xor_pair(fr_t x, fr_t y) { return FADD(FMUL(one_minus(x), y), FMUL(x, one_minus(y))); }       /* (1 - x) y + x (1 - y) */

static fr_t mle_less_than(const fr_t *r, fr_t *eq_out) { /* unsigned_less_than.rs:28-38 */
    fr_t result = fr_zero(), eq_term = ONE();
    for (unsigned i = 0; i < XLEN; ++i) {
        const fr_t x_i = r[2 * i], y_i = r[2 * i + 1];
        result = FADD(result, FMUL(FMUL(one_minus(x_i), y_i), eq_term));
        eq_term = FMUL(eq_term, eq_pair(x_i, y_i));
    }
    if (eq_out) *eq_out = eq_term;
    return result;
}
static fr_t mle_signed_less_than(const fr_t *r) { return FADD(FSUB(r[0], r[1]), mle_less_than(r, NULL)); } /* signed_less_than.rs:29-41 */
static fr_t mle_equal(const fr_t *r) {
    fr_t result = ONE();
    for (unsigned i = 0; i < LOG_K; i += 2) result = FMUL(result, eq_pair(r[i], r[i + 1]));
    return result;
}
static fr_t mle_bitwise(const fr_t *r, int op) { /* and.rs / andn.rs / or.rs / xor.rs: sum_i 2^(63 - i) * op(x_i, y_i) */
    fr_t result = fr_zero();
    for (unsigned i = 0; i < XLEN; ++i) {
        const fr_t x_i = r[2 * i], y_i = r[2 * i + 1];
        fr_t bit;
        switch (op) {
            case 0: bit = FMUL(x_i, y_i); break;
            case 1: bit = FMUL(x_i, one_minus(y_i)); break;
            case 2: bit = FSUB(FADD(x_i, y_i), FMUL(x_i, y_i)); break;
            default: bit = xor_pair(x_i, y_i); break;
        }
        result = FADD(result, FMUL(pow2_fr(XLEN - 1 - i), bit));
    }
    return result;
}
static fr_t mle_pow2(const fr_t *r, unsigned log_width) { /* pow2.rs:20-28: prod_i (1 + (2^(2^i) - 1) r[127 - i]) */
    fr_t result = ONE();
    for (unsigned i = 0; i < log_width; ++i)
        result = FMUL(result, FADD(ONE(), FMUL(fr_from_u64(((uint64_t)1 << ((uint64_t)1 << i)) - 1), r[LOG_K - i - 1])));
    return result;
}
static fr_t mle_xor_rot(const fr_t *r, unsigned rotation) { /* virtual_xor_rot.rs:27-40 */
    fr_t result = fr_zero();
    for (unsigned i = 0; i < XLEN; ++i) {
        const unsigned bit_position = XLEN - 1 - (i + rotation) % XLEN;
        result = FADD(result, FMUL(pow2_fr(bit_position), xor_pair(r[2 * i], r[2 * i + 1])));
    }
    return result;
}
static fr_t mle_xor_rotw(const fr_t *r, unsigned rotation) { /* virtual_xor_rotw.rs:25-41 */
    fr_t result = fr_zero();
    for (unsigned idx = XLEN / 2; idx < XLEN; ++idx) {
        const unsigned position = idx - XLEN / 2, rotated = XLEN / 2 - 1 - (position + rotation) % (XLEN / 2);
        result = FADD(result, FMUL(pow2_fr(rotated), xor_pair(r[2 * idx], r[2 * idx + 1])));
    }
    return result;
}
static fr_t mle_rotr(const fr_t *r, unsigned first_pair) { /* virtual_rotr.rs:32-51, virtual_rotrw.rs:31-50 (skip(XLEN / 2)) */
    fr_t prod_one_plus_y = ONE(), first_sum = fr_zero(), second_sum = fr_zero();
    for (unsigned i = first_pair; i < XLEN; ++i) {
        const fr_t r_x = r[2 * i], r_y = r[2 * i + 1];
        first_sum = FADD(FMUL(first_sum, FADD(ONE(), r_y)), FMUL(r_x, r_y));
        second_sum = FADD(second_sum, FMUL(FMUL(FMUL(r_x, one_minus(r_y)), prod_one_plus_y), pow2_fr(XLEN - 1 - i)));
        prod_one_plus_y = FMUL(prod_one_plus_y, FADD(ONE(), r_y));
    }
    return FADD(first_sum, second_sum);
}

static fr_t evaluate_mle(unsigned kind, const fr_t *r) {
    switch (kind) {
        case T_RangeCheck: case T_RangeCheckAligned: {                                /* range_check.rs:19-31; the aligned table skips the last bit */
            fr_t result = fr_zero();
            const unsigned n = kind == T_RangeCheck ? XLEN : XLEN - 1;
            for (unsigned i = 0; i < n; ++i) result = FADD(result, FMUL(pow2_fr(XLEN - 1 - i), r[XLEN + i]));
            return result;
        }
        case T_And: return mle_bitwise(r, 0);
        case T_Andn: return mle_bitwise(r, 1);
        case T_Or: return mle_bitwise(r, 2);
        case T_Xor: return mle_bitwise(r, 3);
        case T_Equal: return mle_equal(r);
        case T_SignedGreaterThanEqual: return one_minus(mle_signed_less_than(r));
        case T_UnsignedGreaterThanEqual: return one_minus(mle_less_than(r, NULL));
        case T_NotEqual: return one_minus(mle_equal(r));
        case T_SignedLessThan: return mle_signed_less_than(r);
        case T_UnsignedLessThan: return mle_less_than(r, NULL);
        case T_SignMask: return FMUL(r[0], fr_from_u64(~(uint64_t)0));                /* sign_mask.rs:26-35 */
        case T_UpperWord: {
            fr_t result = fr_zero();
            for (unsigned i = 0; i < XLEN; ++i) result = FADD(result, FMUL(pow2_fr(XLEN - 1 - i), r[i]));
            return result;
        }
        case T_UnsignedLessThanEqual: { fr_t eq; const fr_t lt = mle_less_than(r, &eq); return FADD(lt, eq); }
        case T_ValidUnsignedRemainder: {                                              /* valid_unsigned_remainder.rs:20-37 */
            fr_t divisor_is_zero = ONE();
            for (unsigned i = 0; i < XLEN; ++i) divisor_is_zero = FMUL(divisor_is_zero, one_minus(r[2 * i + 1]));
            return FADD(mle_less_than(r, NULL), divisor_is_zero);
        }
        case T_ValidDiv0: {                                                           /* valid_div0.rs:25-41 */
            fr_t divisor_is_zero = ONE(), is_valid_div_by_zero = ONE();
            for (unsigned i = 0; i < XLEN; ++i) {
                const fr_t x_i = r[2 * i], y_i = r[2 * i + 1];
                divisor_is_zero = FMUL(divisor_is_zero, one_minus(x_i));
                is_valid_div_by_zero = FMUL(is_valid_div_by_zero, FMUL(one_minus(x_i), y_i));
            }
            return FADD(FSUB(ONE(), divisor_is_zero), is_valid_div_by_zero);
        }
        case T_HalfwordAlignment: return one_minus(r[LOG_K - 1]);
        case T_WordAlignment: return FMUL(one_minus(r[LOG_K - 1]), one_minus(r[LOG_K - 2]));
        case T_LowerHalfWord: case T_SignExtendWord: {                                /* lower_half_word.rs:22-33, sign_extend_word.rs:28-47 */
            fr_t lower_half = fr_zero();
            for (unsigned i = 0; i < 32; ++i) lower_half = FADD(lower_half, FMUL(pow2_fr(31 - i), r[XLEN + 32 + i]));
            if (kind == T_LowerHalfWord) return lower_half;
            fr_t upper_half = fr_zero();
            for (unsigned i = 0; i < 32; ++i) upper_half = FADD(upper_half, FMUL(pow2_fr(31 - i), r[XLEN + 32]));
            return FADD(lower_half, FMUL(upper_half, pow2_fr(32)));
        }
        case T_Pow2: return mle_pow2(r, 6);
        case T_Pow2W: return mle_pow2(r, 5);
        case T_ShiftRightBitmask: {                                                   /* shift_right_bitmask.rs:20-42 */
            const fr_t *tail = r + LOG_K - 6;
            fr_t sum = fr_zero();
            for (unsigned s = 0; s < XLEN; ++s) {
                const u128 bitmask = (((u128)1 << (XLEN - s)) - 1) << s;
                fr_t eq_val = ONE();
                for (unsigned i = 0; i < 6; ++i) eq_val = FMUL(eq_val, (s >> i) & 1 ? tail[6 - i - 1] : one_minus(tail[6 - i - 1]));
                sum = FADD(sum, FMUL(fr_from_u128((uint64_t)bitmask, (uint64_t)(bitmask >> 64)), eq_val));
            }
            return sum;
        }
        case T_VirtualRev8W: {                                                        /* virtual_rev8w.rs:26-47: bytes a .. h from the low end, output d c b a h g f e */
            fr_t bytes[8];
            for (unsigned b = 0; b < 8; ++b) {
                bytes[b] = fr_zero();
                for (unsigned i = 0; i < 8; ++i) bytes[b] = FADD(bytes[b], fr_mul_u64(r[LOG_K - 1 - (8 * b + i)], (uint64_t)1 << i));
            }
            static const unsigned order[8] = {3, 2, 1, 0, 7, 6, 5, 4};
            fr_t result = fr_zero();
            for (unsigned i = 0; i < 8; ++i) result = FADD(result, fr_mul_u64(bytes[order[i]], (uint64_t)1 << (8 * i)));
            return result;
        }
        case T_VirtualSRL: case T_VirtualSRA: {                                       /* virtual_srl.rs:31-45, virtual_sra.rs:35-54 */
            fr_t result = fr_zero(), sign_extension = fr_zero();
            for (unsigned i = 0; i < XLEN; ++i) {
                const fr_t x_i = r[2 * i], y_i = r[2 * i + 1];
                result = FADD(FMUL(result, FADD(ONE(), y_i)), FMUL(x_i, y_i));
                if (i != 0) sign_extension = FADD(sign_extension, FMUL(pow2_fr(i), one_minus(y_i)));
            }
            return kind == T_VirtualSRL ? result : FADD(result, FMUL(r[0], sign_extension));
        }
        case T_VirtualROTR: return mle_rotr(r, 0);
        case T_VirtualROTRW: return mle_rotr(r, XLEN / 2);
        case T_VirtualChangeDivisor: {                                                /* virtual_change_divisor.rs:29-52 */
            fr_t divisor_value = fr_zero(), x_product = r[0], y_product = ONE();
            for (unsigned i = 0; i < XLEN; ++i) divisor_value = FADD(divisor_value, FMUL(pow2_fr(XLEN - 1 - i), r[2 * i + 1]));
            for (unsigned i = 1; i < XLEN; ++i) x_product = FMUL(x_product, one_minus(r[2 * i]));
            for (unsigned i = 0; i < XLEN; ++i) y_product = FMUL(y_product, r[2 * i + 1]);
            const fr_t adjustment = FSUB(fr_from_u64(2), pow2_fr(XLEN));
            return FADD(divisor_value, FMUL(FMUL(x_product, y_product), adjustment));
        }
        case T_VirtualChangeDivisorW: {                                               /* virtual_change_divisor_w.rs:39-66 */
            const fr_t sign_bit = r[XLEN + 1];
            fr_t divisor_value = fr_zero(), x_product = r[XLEN], y_product = ONE();
            for (unsigned i = XLEN / 2; i < XLEN; ++i) divisor_value = FADD(divisor_value, FMUL(pow2_fr(XLEN - 1 - i), r[2 * i + 1]));
            for (unsigned i = XLEN / 2 + 1; i < XLEN; ++i) x_product = FMUL(x_product, one_minus(r[2 * i]));
            for (unsigned i = XLEN / 2; i < XLEN; ++i) y_product = FMUL(y_product, r[2 * i + 1]);
            const fr_t sign_extension = FMUL(FSUB(pow2_fr(XLEN), pow2_fr(XLEN / 2)), sign_bit);
            const fr_t adjustment = FSUB(fr_from_u64(2), pow2_fr(XLEN));
            return FADD(FADD(divisor_value, FMUL(FMUL(adjustment, x_product), y_product)), sign_extension);
        }
        case T_MulUNoOverflow: {
            fr_t result = ONE();
            for (unsigned i = 0; i < XLEN; ++i) result = FMUL(result, one_minus(r[i]));
            return result;
        }
        case T_VirtualXORROT32: return mle_xor_rot(r, 32);
        case T_VirtualXORROT24: return mle_xor_rot(r, 24);
        case T_VirtualXORROT16: return mle_xor_rot(r, 16);
        case T_VirtualXORROT63: return mle_xor_rot(r, 63);
        case T_VirtualXORROTW16: return mle_xor_rotw(r, 16);
        case T_VirtualXORROTW12: return mle_xor_rotw(r, 12);
        case T_VirtualXORROTW8: return mle_xor_rotw(r, 8);
        case T_VirtualXORROTW7: return mle_xor_rotw(r, 7);
        case T_WindowMaskW: {                                                         /* window_mask_w.rs:28-38: mask (1 + (2^32 - 1) bit2) */
            const fr_t mask = fr_from_u64(0xFFFFFFFFull);
            return FADD(mask, FMUL(mask, FMUL(fr_from_u64(0xFFFFFFFFull), r[LOG_K - 3])));
        }
        case T_PextSigned: {                                                          /* pext_signed.rs:53-77 */
            fr_t pext = fr_zero(), sigma = fr_zero(), sig2pc = fr_zero(), none = ONE();
            for (unsigned i = 0; i < XLEN; ++i) {
                const fr_t x_i = r[2 * i], y_i = r[2 * i + 1], xy = FMUL(x_i, y_i), one_plus_y = FADD(ONE(), y_i);
                pext = FADD(FMUL(pext, one_plus_y), xy);
                sig2pc = FADD(FMUL(sig2pc, one_plus_y), FMUL(none, FADD(xy, xy)));
                sigma = FADD(sigma, FMUL(none, xy));
                none = FMUL(none, one_minus(y_i));
            }
            return FSUB(FADD(pext, FMUL(sigma, pow2_fr(XLEN))), sig2pc);
        }
        default: return fr_zero();
    }
}
EXPORT void orc_table_evaluate_mle(uint32_t kind, const fr_t *r /* 128 */, fr_t *out) { *out = evaluate_mle(kind, r); }

/* ---- operand polynomials of the RAF half (instruction_read_raf.rs:824-871: the prefix / suffix split of these sums) ---- */
static fr_t mle_left_operand(const fr_t *p) { fr_t s = fr_zero(); for (unsigned i = 0; i < XLEN; ++i) s = FADD(s, FMUL(pow2_fr(XLEN - 1 - i), p[2 * i])); return s; }
static fr_t mle_right_operand(const fr_t *p) { fr_t s = fr_zero(); for (unsigned i = 0; i < XLEN; ++i) s = FADD(s, FMUL(pow2_fr(XLEN - 1 - i), p[2 * i + 1])); return s; }
static fr_t mle_identity(const fr_t *p) { fr_t s = fr_zero(); for (unsigned i = 0; i < LOG_K; ++i) s = FADD(s, FMUL(pow2_fr(LOG_K - 1 - i), p[i])); return s; }
static fr_t mle_upper_all_ones(const fr_t *p) { fr_t s = ONE(); for (unsigned i = 0; i < XLEN; ++i) s = FMUL(s, p[i]); return s; }

static fr_t row_summand(const fr_t *point, unsigned table, int raf_flag, fr_t gamma, int canonical) {
    const fr_t gamma_sqr = FMUL(gamma, gamma);
    fr_t value = table == 0xFF ? fr_zero() : evaluate_mle(table, point);
    if (!raf_flag) {
        value = FADD(value, FADD(FMUL(gamma, mle_left_operand(point)), FMUL(gamma_sqr, mle_right_operand(point))));
    } else {
        value = FADD(value, FMUL(gamma_sqr, mle_identity(point)));
        if (canonical) value = FADD(value, FMUL(FMUL(gamma_sqr, gamma), mle_upper_all_ones(point)));
    }
    return value;
}

/* Threads for a sweep over `cycles` rows: a team per 8 rows at most, never more than 32 below 2^16 rows.  A 256-thread team for a 16-row test case costs more in
 * fork / join (and fights the product's own host workers for cores) than the rows cost: 23 s per small GPU test on the 256-thread box before this cap.  Trace-scale
 * sweeps (the sampled from-the-definition rounds at T = 2^20 / 2^22: ~0.3 ms of table evaluations per row) take every thread omp_set_num_threads allows. */
static int row_team(size_t cycles) {
#ifdef _OPENMP
    size_t n = cycles / 8;
    const size_t small_cap = cycles >= ((size_t)1 << 16) ? (size_t)1 << 20 : 32;
    const size_t cap = (size_t)omp_get_max_threads() < small_cap ? (size_t)omp_get_max_threads() : small_cap;
    if (n > cap) n = cap;
    return n < 1 ? 1 : (int)n;
#else
    (void)cycles;
    return 1;
#endif
}

/* input_claim (instruction_read_raf.rs:1545-1575): on the INTEGER side -- materialize_entry and the operands of the index */
EXPORT void orc_read_raf_input_claim(const uint64_t *lookup_index, const uint8_t *table_index, const uint8_t *raf_flag, size_t cycles, const fr_t *u, const fr_t *gamma,
                                     int canonical, fr_t *out) {
    const fr_t gamma_sqr = FMUL(*gamma, *gamma);
    fr_t sum = fr_zero();
    for (size_t j = 0; j < cycles; ++j) {
        const u128 k = ((u128)lookup_index[2 * j + 1] << 64) | lookup_index[2 * j];
        fr_t value = table_index[j] == 0xFF ? fr_zero() : fr_from_u64(materialize_entry(table_index[j], k));
        if (!raf_flag[j]) {
            uint64_t left, right;
            split_operands(k, &left, &right);
            value = FADD(value, FADD(FMUL(*gamma, fr_from_u64(left)), FMUL(gamma_sqr, fr_from_u64(right))));
        } else {
            value = FADD(value, FMUL(gamma_sqr, FADD(fr_from_u64((uint64_t)k), fr_mul_pow_2(fr_from_u64((uint64_t)(k >> 64)), 64))));
            if (canonical && (uint64_t)(k >> 64) == ~(uint64_t)0) value = FADD(value, FMUL(gamma_sqr, *gamma));
        }
        sum = FADD(sum, FMUL(u[j], value));
    }
    *out = sum;
}

/* The address rounds from the definition.  challenges[i] binds address variable i (i = 0 is index bit 127); evals_out[3 * i + c] = s_i(c), c = 0, 1, 2;
 * after the last round: table_values_out[t] = Val_t(r_address) for every table, and the three operand polynomials at r_address
 * (what init_cycle_rounds, :1140-1160, reads off the checkpoints). */
EXPORT void orc_read_raf_address_rounds(const uint64_t *lookup_index, const uint8_t *table_index, const uint8_t *raf_flag, size_t cycles, const fr_t *u, const fr_t *gamma,
                                        int canonical, const fr_t *challenges /* 128 */, fr_t *evals_out /* 128 * 3 */, fr_t *table_values_out /* T_COUNT */,
                                        fr_t *operands_out /* left, right, identity, upper_all_ones */) {
    fr_t *weight = (fr_t *)malloc(cycles * sizeof(fr_t));
    for (size_t j = 0; j < cycles; ++j) weight[j] = u[j];
    for (unsigned i = 0; i < LOG_K; ++i) {
        fr_t sums[3] = {fr_zero(), fr_zero(), fr_zero()};
#pragma omp parallel num_threads(row_team(cycles))
        {
            fr_t local[3] = {fr_zero(), fr_zero(), fr_zero()};
            fr_t point[LOG_K];
            for (unsigned t = 0; t < i; ++t) point[t] = challenges[t];
#pragma omp for schedule(static)
            for (size_t j = 0; j < cycles; ++j) {
                const u128 k = ((u128)lookup_index[2 * j + 1] << 64) | lookup_index[2 * j];
                for (unsigned t = i + 1; t < LOG_K; ++t) point[t] = (k >> (LOG_K - 1 - t)) & 1 ? fr_one() : fr_zero();
                const int bit = (int)((k >> (LOG_K - 1 - i)) & 1);
                for (unsigned c = 0; c < 3; ++c) {
                    const fr_t cf = fr_from_u64(c);
                    const fr_t eq_c = bit ? cf : FSUB(fr_one(), cf);
                    if (fr_is_zero(&eq_c)) continue;
                    point[i] = cf;
                    local[c] = FADD(local[c], FMUL(FMUL(weight[j], eq_c), row_summand(point, table_index[j], raf_flag[j], *gamma, canonical)));
                }
            }
#pragma omp critical
            for (unsigned c = 0; c < 3; ++c) sums[c] = FADD(sums[c], local[c]);
        }
        for (unsigned c = 0; c < 3; ++c) evals_out[3 * i + c] = sums[c];
        for (size_t j = 0; j < cycles; ++j) {
            const u128 k = ((u128)lookup_index[2 * j + 1] << 64) | lookup_index[2 * j];
            const fr_t r = challenges[i];
            weight[j] = FMUL(weight[j], (k >> (LOG_K - 1 - i)) & 1 ? r : FSUB(fr_one(), r));
        }
    }
    free(weight);
    for (unsigned t = 0; t < T_COUNT; ++t) table_values_out[t] = evaluate_mle(t, challenges);
    operands_out[0] = mle_left_operand(challenges);
    operands_out[1] = mle_right_operand(challenges);
    operands_out[2] = mle_identity(challenges);
    operands_out[3] = mle_upper_all_ones(challenges);
}

/* The same rounds one at a time, for a prover that draws challenge i from a transcript that has absorbed message i: `weight` (cycles entries, in / out) holds
 * eq(r_reduction, j) * eq(r_{<i}, k_j[<i]); round: evals_out = s_i(0), s_i(1), s_i(2); bind: the factor of challenge i. */
EXPORT void orc_read_raf_address_round(const uint64_t *lookup_index, const uint8_t *table_index, const uint8_t *raf_flag, size_t cycles, const fr_t *weight, const fr_t *gamma,
                                       int canonical, const fr_t *challenges /* i of them */, uint32_t i, fr_t *evals_out /* 3 */) {
    fr_t sums[3] = {fr_zero(), fr_zero(), fr_zero()};
#pragma omp parallel num_threads(row_team(cycles))
    {
        fr_t local[3] = {fr_zero(), fr_zero(), fr_zero()};
        fr_t point[LOG_K];
        for (unsigned t = 0; t < i; ++t) point[t] = challenges[t];
#pragma omp for schedule(static)
        for (size_t j = 0; j < cycles; ++j) {
            const u128 k = ((u128)lookup_index[2 * j + 1] << 64) | lookup_index[2 * j];
            for (unsigned t = i + 1; t < LOG_K; ++t) point[t] = (k >> (LOG_K - 1 - t)) & 1 ? fr_one() : fr_zero();
            const int bit = (int)((k >> (LOG_K - 1 - i)) & 1);
            for (unsigned c = 0; c < 3; ++c) {
                const fr_t cf = fr_from_u64(c);
                const fr_t eq_c = bit ? cf : FSUB(fr_one(), cf);
                if (fr_is_zero(&eq_c)) continue;
                point[i] = cf;
                local[c] = FADD(local[c], FMUL(FMUL(weight[j], eq_c), row_summand(point, table_index[j], raf_flag[j], *gamma, canonical)));
            }
        }
#pragma omp critical
        for (unsigned c = 0; c < 3; ++c) sums[c] = FADD(sums[c], local[c]);
    }
    for (unsigned c = 0; c < 3; ++c) evals_out[c] = sums[c];
}
EXPORT void orc_read_raf_address_bind(const uint64_t *lookup_index, size_t cycles, fr_t *weight, uint32_t i, const fr_t *challenge) {
    for (size_t j = 0; j < cycles; ++j) {
        const u128 k = ((u128)lookup_index[2 * j + 1] << 64) | lookup_index[2 * j];
        weight[j] = FMUL(weight[j], (k >> (LOG_K - 1 - i)) & 1 ? *challenge : FSUB(fr_one(), *challenge));
    }
}
/* Val_t(r_address) of every table and the operand polynomials at r_address: left, right, identity, upper_all_ones */
EXPORT void orc_read_raf_address_values(const fr_t *challenges /* 128 */, fr_t *table_values_out /* T_COUNT */, fr_t *operands_out /* 4 */) {
    for (unsigned t = 0; t < T_COUNT; ++t) table_values_out[t] = evaluate_mle(t, challenges);
    operands_out[0] = mle_left_operand(challenges);
    operands_out[1] = mle_right_operand(challenges);
    operands_out[2] = mle_identity(challenges);
    operands_out[3] = mle_upper_all_ones(challenges);
}
