/*
 * oracle/mont256.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of 256-bit Montgomery prime-field arithmetic as the reference
 * gets it from its (un-vendored) arkworks dependency:
 *   ark-ff 0.5.0 @ a16z/arkworks-algebra dev/twist-shout 76bb3a4518928f1ff7f15875f940d614bb9845e6
 *   (Cargo.lock:801,819,885).  Representation = 4 x u64 little-endian limbs holding a*R mod p,
 *   R = 2^256, always canonical (< p)  -- /root/reference/crates/jolt-field/src/bn254/mod.rs:33-43.
 * The reference call sites this stands in for:
 *   add/sub/mul/neg      crates/jolt-field/src/bn254/mod.rs:76-83  (delegating to ark Fp)
 *   inverse              crates/jolt-field/src/bn254/mod.rs:101-105
 *   from_montgomery_reduce (the same 4-step REDC)  crates/jolt-field/src/bn254/mont.rs:186-238
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
 * Parity pins: tests/test_oracle_field.py checks it against the reference's golden vectors
 * (crates/jolt-field/tests/golden_bytes.rs:68-254) and a Python big-integer model
 * (the reference's own differential strategy, crates/jolt-field/tests/bn254_differential.rs:76-272).
 */
#pragma once
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } u256;

typedef struct {
    u256 p;       /* modulus */
    u256 r;       /* R mod p   = Montgomery form of 1 */
    u256 r2;      /* R^2 mod p */
    uint64_t inv; /* -p^-1 mod 2^64 */
} mont_field;

static inline int u256_geq(const u256 *a, const u256 *b) {
    for (int i = 3; i >= 0; --i) {
        if (a->l[i] != b->l[i]) return a->l[i] > b->l[i];
    }
    return 1;
}
static inline int u256_eq(const u256 *a, const u256 *b) {
    return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
static inline int u256_is_zero(const u256 *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }

/* out = a + b, returns carry */
static inline uint64_t u256_add(u256 *out, const u256 *a, const u256 *b) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) {
        c += (u128)a->l[i] + b->l[i];
        out->l[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}
/* out = a - b, returns borrow */
static inline uint64_t u256_sub(u256 *out, const u256 *a, const u256 *b) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 t = (u128)a->l[i] - b->l[i] - borrow;
        out->l[i] = (uint64_t)t;
        borrow = (uint64_t)(t >> 64) & 1;
    }
    return borrow;
}

static inline void mont_add(u256 *out, const u256 *a, const u256 *b, const mont_field *F) {
    u256 s;
    uint64_t carry = u256_add(&s, a, b);
    if (carry || u256_geq(&s, &F->p)) u256_sub(&s, &s, &F->p);
    *out = s;
}
static inline void mont_sub(u256 *out, const u256 *a, const u256 *b, const mont_field *F) {
    u256 d;
    if (u256_sub(&d, a, b)) u256_add(&d, &d, &F->p);
    *out = d;
}
static inline void mont_neg(u256 *out, const u256 *a, const mont_field *F) {
    if (u256_is_zero(a)) { *out = *a; return; }
    u256_sub(out, &F->p, a);
}
static inline void mont_double(u256 *out, const u256 *a, const mont_field *F) { mont_add(out, a, a, F); }

/* REDC of an 8-limb integer t < p*2^256: returns t * 2^-256 mod p (canonical).
 * Same 4-step word-serial reduction as mont.rs:186-198. */
static inline void mont_redc(u256 *out, const uint64_t t_in[8], const mont_field *F) {
    uint64_t t[9];
    memcpy(t, t_in, 8 * sizeof(uint64_t));
    t[8] = 0;
    for (int i = 0; i < 4; ++i) {
        uint64_t m = t[i] * F->inv;
        u128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (u128)m * F->p.l[j] + t[i + j];
            t[i + j] = (uint64_t)c;
            c >>= 64;
        }
        for (int j = i + 4; j < 9 && c; ++j) {
            c += t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
    }
    u256 r = {{t[4], t[5], t[6], t[7]}};
    if (t[8] || u256_geq(&r, &F->p)) u256_sub(&r, &r, &F->p);
    *out = r;
}

static inline void u256_mul_wide(uint64_t t[8], const u256 *a, const u256 *b) {
    memset(t, 0, 8 * sizeof(uint64_t));
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (u128)a->l[i] * b->l[j] + t[i + j];
            t[i + j] = (uint64_t)c;
            c >>= 64;
        }
        t[i + 4] = (uint64_t)c;
    }
}

static inline void mont_mul(u256 *out, const u256 *a, const u256 *b, const mont_field *F) {
    uint64_t t[8];
    u256_mul_wide(t, a, b);
    mont_redc(out, t, F);
}
static inline void mont_sqr(u256 *out, const u256 *a, const mont_field *F) { mont_mul(out, a, a, F); }

/* canonical integer (4 limbs, must be < p) -> Montgomery form */
static inline void mont_from_canonical(u256 *out, const u256 *a, const mont_field *F) { mont_mul(out, a, &F->r2, F); }
/* Montgomery form -> canonical integer */
static inline void mont_to_canonical(u256 *out, const u256 *a, const mont_field *F) {
    uint64_t t[8] = {a->l[0], a->l[1], a->l[2], a->l[3], 0, 0, 0, 0};
    mont_redc(out, t, F);
}

/* a^e for a 256-bit exponent given as canonical limbs (square-and-multiply, MSB first) */
static inline void mont_pow(u256 *out, const u256 *a, const u256 *e, const mont_field *F) {
    u256 acc = F->r;
    for (int i = 255; i >= 0; --i) {
        mont_sqr(&acc, &acc, F);
        if ((e->l[i / 64] >> (i % 64)) & 1) mont_mul(&acc, &acc, a, F);
    }
    *out = acc;
}
/* a^-1 via Fermat (p prime); returns 0 for a == 0 (reference: inverse() -> None) */
static inline int mont_inv(u256 *out, const u256 *a, const mont_field *F) {
    if (u256_is_zero(a)) { memset(out, 0, sizeof(*out)); return 0; }
    u256 e = F->p;
    u256 two = {{2, 0, 0, 0}};
    u256_sub(&e, &e, &two);
    mont_pow(out, a, &e, F);
    return 1;
}
