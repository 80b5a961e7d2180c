/*
 * oracle/fr.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * BN254 scalar field Fr (and base field Fq) for the CPU oracle: thin static-inline
 * wrappers binding mont256.h to the two BN254 moduli, plus the jolt-field specific
 * conversions restated from /root/reference/crates/jolt-field/src/bn254/{mod.rs,mont.rs}.
 */
#pragma once
#include "bn254_constants.h"
#include "mont256.h"
#include <stddef.h>

typedef u256 fr_t; /* Montgomery form, canonical; same bytes as jolt_field::Fr::inner_limbs() (mod.rs:37-43) */
typedef u256 fq_t;

static const mont_field FR = {{FR_P_LIMBS}, {FR_R_LIMBS}, {FR_R2_LIMBS}, FR_INV};
static const mont_field FQ = {{FQ_P_LIMBS}, {FQ_R_LIMBS}, {FQ_R2_LIMBS}, FQ_INV};

static inline void fr_add(fr_t *o, const fr_t *a, const fr_t *b) { mont_add(o, a, b, &FR); }
static inline void fr_sub(fr_t *o, const fr_t *a, const fr_t *b) { mont_sub(o, a, b, &FR); }
static inline void fr_mul(fr_t *o, const fr_t *a, const fr_t *b) { mont_mul(o, a, b, &FR); }
static inline void fr_neg(fr_t *o, const fr_t *a) { mont_neg(o, a, &FR); }
static inline void fr_sqr(fr_t *o, const fr_t *a) { mont_sqr(o, a, &FR); }
static inline int fr_inv(fr_t *o, const fr_t *a) { return mont_inv(o, a, &FR); }
static inline fr_t fr_zero(void) { fr_t z = {{0, 0, 0, 0}}; return z; }
static inline fr_t fr_one(void) { return FR.r; }
static inline int fr_eq(const fr_t *a, const fr_t *b) { return u256_eq(a, b); }
static inline int fr_is_zero(const fr_t *a) { return u256_is_zero(a); }

static inline void fq_add(fq_t *o, const fq_t *a, const fq_t *b) { mont_add(o, a, b, &FQ); }
static inline void fq_sub(fq_t *o, const fq_t *a, const fq_t *b) { mont_sub(o, a, b, &FQ); }
static inline void fq_mul(fq_t *o, const fq_t *a, const fq_t *b) { mont_mul(o, a, b, &FQ); }
static inline void fq_neg(fq_t *o, const fq_t *a) { mont_neg(o, a, &FQ); }
static inline void fq_sqr(fq_t *o, const fq_t *a) { mont_sqr(o, a, &FQ); }
static inline int fq_inv(fq_t *o, const fq_t *a) { return mont_inv(o, a, &FQ); }

/* value-returning conveniences used by the algorithm restatements */
static inline fr_t FADD(fr_t a, fr_t b) { fr_t o; fr_add(&o, &a, &b); return o; }
static inline fr_t FSUB(fr_t a, fr_t b) { fr_t o; fr_sub(&o, &a, &b); return o; }
static inline fr_t FMUL(fr_t a, fr_t b) { fr_t o; fr_mul(&o, &a, &b); return o; }
static inline fr_t FNEG(fr_t a) { fr_t o; fr_neg(&o, &a); return o; }

/* ---- jolt-field conversions (defined in fr.c) ---- */
fr_t fr_from_u64(uint64_t v);                       /* mont.rs:307-315 (value == ark Fr::from(v)) */
fr_t fr_from_u128(uint64_t lo, uint64_t hi);        /* mont.rs:317-325 */
fr_t fr_from_i64(int64_t v);                        /* mod.rs:271-278 */
fr_t fr_from_i128(uint64_t mag_lo, uint64_t mag_hi, int negative); /* mod.rs:285-292 */
fr_t fr_mul_u64(fr_t a, uint64_t b);                /* mont.rs:286-295 (Barrett path) */
fr_t fr_mul_u128(fr_t a, uint64_t lo, uint64_t hi); /* mont.rs:297-305 */
fr_t fr_mul_pow_2(fr_t a, unsigned k);              /* used by prove_batch padding, prover.rs:247 */
fr_t fr_from_bytes_le_reduced(const uint8_t *b, size_t n);      /* mod.rs:129-132 */
fr_t fr_from_challenge_bytes(const uint8_t *b, size_t n);       /* mod.rs:171-184, 254 */
fr_t fr_from_scalar_challenge_bytes(const uint8_t *b, size_t n);/* mod.rs:188-193 */
void fr_to_bytes_le(uint8_t out[32], fr_t a);                   /* mod.rs:116-123 */
fr_t fr_barrett_reduce_5(const uint64_t c[5]);                  /* mont.rs:154-181 */
fr_t fr_from_montgomery_reduce(const uint64_t *limbs, size_t L);/* mont.rs:204-238 */
