/*
 * oracle/fr.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Restatement of the jolt-field BN254 Fr conversion / Barrett / deferred-reduction helpers,
 * following /root/reference/crates/jolt-field/src/bn254/mont.rs and mod.rs (line refs inline).
 * Exported `orc_*` symbols are the ctypes surface used by tests/ and bench.py's cpu_baseline.
 */
#include "fr.h"

static const u256 FR_P2 = {FR_P2_LIMBS};
static const u256 FR_P3 = {FR_P3_LIMBS};

/* mont.rs:134-152 barrett_cond_subtract: value < 4p -> < p (top limb known zero) */
static fr_t barrett_cond_subtract(const uint64_t r[5]) {
    u256 v = {{r[0], r[1], r[2], r[3]}};
    if (u256_geq(&v, &FR_P2)) {
        if (u256_geq(&v, &FR_P3)) u256_sub(&v, &v, &FR_P3);
        else u256_sub(&v, &v, &FR_P2);
    } else if (u256_geq(&v, &FR.p)) {
        u256_sub(&v, &v, &FR.p);
    }
    return v;
}

/* mont.rs:154-181 barrett_reduce_5_to_4: 5-limb integer -> integer mod p */
fr_t fr_barrett_reduce_5(const uint64_t c[5]) {
    uint64_t tilde_c = (c[4] << FR_SPARE_BITS) + (c[3] >> (64 - FR_SPARE_BITS));
    uint64_t m = (uint64_t)(((u128)tilde_c * FR_BARRETT_MU) >> 64);
    /* m * 2p, 5 limbs */
    uint64_t m2p[5] = {FR_P2.l[0], FR_P2.l[1], FR_P2.l[2], FR_P2.l[3], 0};
    uint64_t carry = 0;
    for (int i = 0; i < 5; ++i) {
        u128 prod = (u128)m2p[i] * m + carry;
        m2p[i] = (uint64_t)prod;
        carry = (uint64_t)(prod >> 64);
    }
    uint64_t r[5];
    uint64_t borrow = 0;
    for (int i = 0; i < 5; ++i) {
        u128 t = (u128)c[i] - m2p[i] - borrow;
        r[i] = (uint64_t)t;
        borrow = (uint64_t)(t >> 64) & 1;
    }
    return barrett_cond_subtract(r);
}

/* mont.rs:240-250 + 286-295 */
fr_t fr_mul_u64(fr_t a, uint64_t b) {
    if (b == 0 || fr_is_zero(&a)) return fr_zero();
    if (b == 1) return a;
    uint64_t res[5];
    uint64_t carry = 0;
    for (int i = 0; i < 4; ++i) {
        u128 t = (u128)a.l[i] * b + carry;
        res[i] = (uint64_t)t;
        carry = (uint64_t)(t >> 64);
    }
    res[4] = carry;
    return fr_barrett_reduce_5(res);
}

/* mont.rs:252-283 + 297-305 */
fr_t fr_mul_u128(fr_t a, uint64_t b_lo, uint64_t b_hi) {
    if (b_hi == 0) return fr_mul_u64(a, b_lo);
    uint64_t res[6] = {0, 0, 0, 0, 0, 0};
    uint64_t carry = 0;
    for (int i = 0; i < 4; ++i) {
        u128 t = (u128)a.l[i] * b_lo + res[i] + carry;
        res[i] = (uint64_t)t;
        carry = (uint64_t)(t >> 64);
    }
    res[4] = carry;
    uint64_t carry2 = 0;
    for (int i = 0; i < 4; ++i) {
        u128 t = (u128)a.l[i] * b_hi + res[i + 1] + carry2;
        res[i + 1] = (uint64_t)t;
        carry2 = (uint64_t)(t >> 64);
    }
    res[5] = carry2;
    /* from_unchecked_nplus2: two Barrett rounds */
    uint64_t c1[5] = {res[1], res[2], res[3], res[4], res[5]};
    fr_t r1 = fr_barrett_reduce_5(c1);
    uint64_t c2[5] = {res[0], r1.l[0], r1.l[1], r1.l[2], r1.l[3]};
    return fr_barrett_reduce_5(c2);
}

/* mont.rs:307-325: small values come from a table of Montgomery forms, the rest via mul(R, n);
 * both equal the Montgomery form of n. */
fr_t fr_from_u64(uint64_t v) { return fr_mul_u64(fr_one(), v); }
fr_t fr_from_u128(uint64_t lo, uint64_t hi) { return fr_mul_u128(fr_one(), lo, hi); }

/* mod.rs:271-278 */
fr_t fr_from_i64(int64_t v) {
    if (v < 0) return FNEG(fr_from_u64((uint64_t)0 - (uint64_t)v));
    return fr_from_u64((uint64_t)v);
}
/* mod.rs:285-292 (magnitude passed explicitly: C has no i128 ABI for ctypes) */
fr_t fr_from_i128(uint64_t mag_lo, uint64_t mag_hi, int negative) {
    fr_t m = fr_from_u128(mag_lo, mag_hi);
    return negative ? FNEG(m) : m;
}

fr_t fr_mul_pow_2(fr_t a, unsigned k) {
    for (unsigned i = 0; i < k; ++i) a = FADD(a, a);
    return a;
}

/* mod.rs:129-132 -> ark from_le_bytes_mod_order: integer value of the bytes, reduced mod p */
fr_t fr_from_bytes_le_reduced(const uint8_t *b, size_t n) {
    fr_t acc = fr_zero();
    fr_t base = fr_from_u64(256);
    for (size_t i = n; i > 0; --i) {
        acc = FADD(FMUL(acc, base), fr_from_u64(b[i - 1]));
    }
    return acc;
}

/* mod.rs:171-184 with the Fr arm at mod.rs:254: the masked 125-bit value is placed in the two
 * HIGH limbs and handed to the fork's raw from_bigint_unchecked, i.e. those limbs ARE the
 * Montgomery representation (pinned by golden_bytes.rs:186-220). */
fr_t fr_from_challenge_bytes(const uint8_t *b, size_t n) {
    uint8_t buf[16] = {0};
    size_t len = n < 16 ? n : 16;
    memcpy(buf, b, len);
    uint64_t low = 0, high = 0;
    for (int i = 7; i >= 0; --i) low = (low << 8) | buf[i];
    for (int i = 15; i >= 8; --i) high = (high << 8) | buf[i];
    high &= (UINT64_MAX >> 3);
    fr_t out = {{0, 0, low, high}};
    return out;
}

/* mod.rs:188-193 */
fr_t fr_from_scalar_challenge_bytes(const uint8_t *b, size_t n) {
    uint8_t buf[64];
    if (n > sizeof buf) n = sizeof buf;
    for (size_t i = 0; i < n; ++i) buf[i] = b[n - 1 - i];
    return fr_from_bytes_le_reduced(buf, n);
}

/* mod.rs:116-123: canonical 32-byte little-endian */
void fr_to_bytes_le(uint8_t out[32], fr_t a) {
    u256 c;
    mont_to_canonical(&c, &a, &FR);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) out[8 * i + j] = (uint8_t)(c.l[i] >> (8 * j));
}

/* mont.rs:204-238: L-limb integer (L >= 8) -> value * R^-1 mod p.  Tail above 2N limbs is folded
 * with Barrett rounds, then the standard 4-step REDC runs. */
fr_t fr_from_montgomery_reduce(const uint64_t *limbs, size_t L) {
    uint64_t buf[16] = {0};
    if (L > 16) L = 16;
    memcpy(buf, limbs, L * sizeof(uint64_t));
    if (L > 8) {
        uint64_t acc[4] = {0, 0, 0, 0};
        size_t i = L;
        while (i > 4) {
            i -= 1;
            uint64_t c5[5] = {buf[i], acc[0], acc[1], acc[2], acc[3]};
            fr_t red = fr_barrett_reduce_5(c5);
            memcpy(acc, red.l, sizeof acc);
        }
        memcpy(&buf[4], acc, sizeof acc);
        for (size_t k = 8; k < L; ++k) buf[k] = 0;
    }
    fr_t out;
    mont_redc(&out, buf, &FR);
    return out;
}

/* ------------------------------------------------------------------------------------------
 * ctypes surface
 * ------------------------------------------------------------------------------------------ */
#define EXPORT __attribute__((visibility("default")))

EXPORT void orc_fr_add_vec(const fr_t *a, const fr_t *b, fr_t *o, size_t n) { for (size_t i = 0; i < n; ++i) fr_add(&o[i], &a[i], &b[i]); }
EXPORT void orc_fr_sub_vec(const fr_t *a, const fr_t *b, fr_t *o, size_t n) { for (size_t i = 0; i < n; ++i) fr_sub(&o[i], &a[i], &b[i]); }
EXPORT void orc_fr_mul_vec(const fr_t *a, const fr_t *b, fr_t *o, size_t n) { for (size_t i = 0; i < n; ++i) fr_mul(&o[i], &a[i], &b[i]); }
EXPORT void orc_fr_neg_vec(const fr_t *a, fr_t *o, size_t n) { for (size_t i = 0; i < n; ++i) fr_neg(&o[i], &a[i]); }
EXPORT void orc_fr_inv_vec(const fr_t *a, fr_t *o, size_t n) { for (size_t i = 0; i < n; ++i) fr_inv(&o[i], &a[i]); }
EXPORT void orc_fq_add_vec(const fq_t *a, const fq_t *b, fq_t *o, size_t n) { for (size_t i = 0; i < n; ++i) fq_add(&o[i], &a[i], &b[i]); }
EXPORT void orc_fq_sub_vec(const fq_t *a, const fq_t *b, fq_t *o, size_t n) { for (size_t i = 0; i < n; ++i) fq_sub(&o[i], &a[i], &b[i]); }
EXPORT void orc_fq_mul_vec(const fq_t *a, const fq_t *b, fq_t *o, size_t n) { for (size_t i = 0; i < n; ++i) fq_mul(&o[i], &a[i], &b[i]); }
EXPORT void orc_fq_inv_vec(const fq_t *a, fq_t *o, size_t n) { for (size_t i = 0; i < n; ++i) fq_inv(&o[i], &a[i]); }
EXPORT void orc_fr_from_u64_vec(const uint64_t *v, fr_t *o, size_t n) {
#pragma omp parallel for schedule(static) if (n >= 65536)
    for (size_t i = 0; i < n; ++i) o[i] = fr_from_u64(v[i]);
}
EXPORT void orc_fr_from_i64_vec(const int64_t *v, fr_t *o, size_t n) {
#pragma omp parallel for schedule(static) if (n >= 65536)
    for (size_t i = 0; i < n; ++i) o[i] = fr_from_i64(v[i]);
}
EXPORT void orc_fr_from_u128(uint64_t lo, uint64_t hi, fr_t *o) { *o = fr_from_u128(lo, hi); }
EXPORT void orc_fr_from_i128(uint64_t lo, uint64_t hi, int neg, fr_t *o) { *o = fr_from_i128(lo, hi, neg); }
EXPORT void orc_fr_mul_u64(const fr_t *a, uint64_t b, fr_t *o) { *o = fr_mul_u64(*a, b); }
EXPORT void orc_fr_mul_u128(const fr_t *a, uint64_t lo, uint64_t hi, fr_t *o) { *o = fr_mul_u128(*a, lo, hi); }
EXPORT void orc_fr_mul_pow_2(const fr_t *a, unsigned k, fr_t *o) { *o = fr_mul_pow_2(*a, k); }
EXPORT void orc_fr_from_bytes_le_reduced(const uint8_t *b, size_t n, fr_t *o) { *o = fr_from_bytes_le_reduced(b, n); }
EXPORT void orc_fr_from_challenge_bytes(const uint8_t *b, size_t n, fr_t *o) { *o = fr_from_challenge_bytes(b, n); }
EXPORT void orc_fr_from_scalar_challenge_bytes(const uint8_t *b, size_t n, fr_t *o) { *o = fr_from_scalar_challenge_bytes(b, n); }
EXPORT void orc_fr_to_bytes_le(const fr_t *a, uint8_t *out) { fr_to_bytes_le(out, *a); }
EXPORT void orc_fr_to_canonical_vec(const fr_t *a, u256 *o, size_t n) { for (size_t i = 0; i < n; ++i) mont_to_canonical(&o[i], &a[i], &FR); }
EXPORT void orc_fr_from_canonical_vec(const u256 *a, fr_t *o, size_t n) { for (size_t i = 0; i < n; ++i) mont_from_canonical(&o[i], &a[i], &FR); }
EXPORT void orc_fq_to_canonical_vec(const fq_t *a, u256 *o, size_t n) { for (size_t i = 0; i < n; ++i) mont_to_canonical(&o[i], &a[i], &FQ); }
EXPORT void orc_fq_from_canonical_vec(const u256 *a, fq_t *o, size_t n) { for (size_t i = 0; i < n; ++i) mont_from_canonical(&o[i], &a[i], &FQ); }
EXPORT void orc_fr_from_montgomery_reduce(const uint64_t *limbs, size_t L, fr_t *o) { *o = fr_from_montgomery_reduce(limbs, L); }

/* WideAccumulator (mont.rs:334-336, 565-602): Sum a_i*b_i (+ plain adds) with ONE deferred reduction.
 * slots are positional u128 sums of 64-bit product halves; reduce = carry pass + from_montgomery_reduce<9>. */
EXPORT void orc_wide_accumulate(const fr_t *a, const fr_t *b, size_t n_fmadd, const fr_t *adds, size_t n_add, fr_t *o) {
    u128 slots[8] = {0};
    for (size_t k = 0; k < n_fmadd; ++k) {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                u128 prod = (u128)a[k].l[i] * b[k].l[j];
                slots[i + j] += (uint64_t)prod;
                slots[i + j + 1] += (uint64_t)(prod >> 64);
            }
    }
    for (size_t k = 0; k < n_add; ++k) /* mont.rs:573-578: element enters at limbs 4..8 (= value * R) */
        for (int i = 0; i < 4; ++i) slots[4 + i] += adds[k].l[i];
    uint64_t out[9];
    u128 carry = 0;
    for (int i = 0; i < 8; ++i) {
        u128 sum = slots[i] + carry;
        int overflow = sum < carry;
        out[i] = (uint64_t)sum;
        carry = (sum >> 64) + ((u128)overflow << 64);
    }
    out[8] = (uint64_t)carry;
    *o = fr_from_montgomery_reduce(out, 9);
}

/* FrSmallScalarAccumulator (mont.rs:343-427): sum_k value_k * scalar_k for signed 64-bit scalars; positive and negative terms are
 * held separately as unreduced FIVE-limb integers (add_assign_trunc: wraps at 2^320 like the reference) and reduced once
 * (reduce: Barrett of |pos - neg|, negated when neg > pos).  fmadd_i64 (:417-426), fmadd_magnitude (:369-379), reduce (:397-409). */
static void limbs5_add(uint64_t acc[5], const uint64_t v[5]) {
    u128 c = 0;
    for (int i = 0; i < 5; ++i) {
        c += (u128)acc[i] + v[i];
        acc[i] = (uint64_t)c;
        c >>= 64;
    }
}
static int limbs5_geq(const uint64_t a[5], const uint64_t b[5]) {
    for (int i = 4; i >= 0; --i)
        if (a[i] != b[i]) return a[i] > b[i];
    return 1;
}
static void limbs5_sub(uint64_t o[5], const uint64_t a[5], const uint64_t b[5]) {
    u128 br = 0;
    for (int i = 0; i < 5; ++i) {
        u128 d = (u128)a[i] - b[i] - br;
        o[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}
EXPORT void orc_small_scalar_accumulate(const fr_t *values, const int64_t *scalars, size_t n, fr_t *o) {
    uint64_t pos[5] = {0}, neg[5] = {0};
    for (size_t k = 0; k < n; ++k) {
        const int64_t sc = scalars[k];
        const uint64_t mag = sc < 0 ? (uint64_t)0 - (uint64_t)sc : (uint64_t)sc;
        uint64_t *slots = sc >= 0 ? pos : neg;
        if (mag == 0) continue;
        uint64_t prod[5] = {0};
        if (mag == 1) {
            for (int i = 0; i < 4; ++i) prod[i] = values[k].l[i];
        } else { /* bigint4_mul_u64 */
            u128 c = 0;
            for (int i = 0; i < 4; ++i) {
                c += (u128)values[k].l[i] * mag;
                prod[i] = (uint64_t)c;
                c >>= 64;
            }
            prod[4] = (uint64_t)c;
        }
        limbs5_add(slots, prod);
    }
    uint64_t d[5];
    if (limbs5_geq(pos, neg)) {
        limbs5_sub(d, pos, neg);
        *o = fr_barrett_reduce_5(d);
    } else {
        limbs5_sub(d, neg, pos);
        *o = FNEG(fr_barrett_reduce_5(d));
    }
}
