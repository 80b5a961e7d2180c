// jolt_amd/csrc/small_scalar.hip.h -- field x machine-integer products with deferred reduction, on the device.
//
// The device analogue of FrSmallScalarAccumulator / mul_u64 / mul_u128 (crates/jolt-field/src/bn254/mont.rs:286-305,343-427; trait
// crates/jolt-field/src/algebra.rs:362-433): witness columns are small integers (flags, u64 / i64 registers, a few i128), so
// sum_k a_k * z_k with a_k a field element and z_k an integer is accumulated as an UNREDUCED integer -- a 256 x 64 (or x 128) bit
// product is 16 (32) multiply-adds where a Montgomery product is 162 -- and reduced ONCE.  The reference keeps 5-limb positive /
// negative sums and Barrett-reduces; here the sum is 13 x 32 bits (headroom: p * 2^128 * 2^34 terms) and the single reduction is a
// Montgomery REDC, which divides by R = 2^256: the callers pre-scale one operand by R where that is free (the per-column weights of
// small_r1cs.hip) or multiply the handful of final values by R^2 (the eq-weighted evaluations).  Same canonical values as the
// reference's Barrett path: both are exact mod p.
#pragma once
#include "field.hip.h"

namespace jolt {

// sign-magnitude of one machine integer of a jolt_ints column (JOLT_INT_U64 / _I64 / _I128), little-endian 32-bit limbs
struct SmallInt {
    uint32_t m[4];
    uint32_t neg;
};
constexpr int kIntKindU64 = 0, kIntKindI64 = 1, kIntKindI128 = 2;  // = JOLT_INT_* of include/jolt_hip.h

JOLT_HD SmallInt load_small(const void* __restrict__ data, int kind, size_t i) {
    SmallInt s;
    uint64_t lo, hi = 0;
    bool negative = false;
    if (kind == kIntKindI128) {
        const uint64_t* p = reinterpret_cast<const uint64_t*>(data) + 2 * i;
        lo = p[0];
        hi = p[1];
        negative = (hi >> 63) != 0;
        if (negative) {
            lo = ~lo + 1;
            hi = ~hi + (lo == 0 ? 1 : 0);
        }
    } else {
        lo = reinterpret_cast<const uint64_t*>(data)[i];
        negative = kind == kIntKindI64 && (lo >> 63) != 0;
        if (negative) lo = ~lo + 1;
    }
    s.m[0] = (uint32_t)lo;
    s.m[1] = (uint32_t)(lo >> 32);
    s.m[2] = (uint32_t)hi;
    s.m[3] = (uint32_t)(hi >> 32);
    s.neg = negative ? 1u : 0u;
    return s;
}

constexpr int kSmallLimbs = 13;  // 416 bits
struct SmallAcc {
    uint32_t l[kSmallLimbs];
};
JOLT_HD SmallAcc small_zero() {
    SmallAcc a;
#pragma unroll
    for (int i = 0; i < kSmallLimbs; ++i) a.l[i] = 0;
    return a;
}
// acc += a * m, m = LIMBS 32-bit limbs of the magnitude (2 for the 64-bit kinds, 4 for i128)
template <int LIMBS, class PR>
JOLT_HD void small_fmadd(SmallAcc& acc, const Fp<PR>& a, const uint32_t (&m)[4]) {
#pragma unroll
    for (int i = 0; i < LIMBS; ++i) {
        uint64_t p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = (uint64_t)a.l[j] * m[i] + acc.l[i + j];
        uint32_t c = 0;
        acc.l[i] = (uint32_t)p[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) acc.l[i + j] = __builtin_addc((uint32_t)p[j], (uint32_t)(p[j - 1] >> 32), c, &c);
        acc.l[i + 8] = __builtin_addc(acc.l[i + 8], (uint32_t)(p[7] >> 32), c, &c);
#pragma unroll
        for (int k = i + 9; k < kSmallLimbs; ++k) acc.l[k] = __builtin_addc(acc.l[k], 0u, c, &c);
    }
}
// acc * 2^-256 mod p, canonical (one Montgomery REDC; the 13-limb input keeps the output below p + 2^160 < 2p)
template <class PR>
JOLT_HD Fp<PR> small_redc(const SmallAcc& acc) {
    uint32_t t[kSmallLimbs + 9];
#pragma unroll
    for (int i = 0; i < kSmallLimbs; ++i) t[i] = acc.l[i];
#pragma unroll
    for (int i = kSmallLimbs; i < kSmallLimbs + 9; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t mq = t[i] * PR::INV;
        uint64_t q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = (uint64_t)mq * (uint32_t)PR::P[j] + t[i + j];
        uint32_t c = 0;
#pragma unroll
        for (int j = 1; j < 8; ++j) t[i + j] = __builtin_addc((uint32_t)q[j], (uint32_t)(q[j - 1] >> 32), c, &c);
        t[i + 8] = __builtin_addc(t[i + 8], (uint32_t)(q[7] >> 32), c, &c);
#pragma unroll
        for (int k = i + 9; k < kSmallLimbs + 9; ++k) t[k] = __builtin_addc(t[k], 0u, c, &c);
    }
    Fp<PR> r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = t[8 + i];
    return reduce_once(r, 0u);
}

// ---- exact integer arithmetic of the uni-skip extension (S64 x S192 -> S256 of crates/jolt-field/src/signed) -------------------
struct U256 {
    uint32_t l[8];
};
JOLT_HD U256 u256_zero() {
    U256 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z.l[i] = 0;
    return z;
}
// acc += w * m (w: 64-bit magnitude, m: LIMBS x 32 bits), truncated to 256 bits
template <int LIMBS>
JOLT_HD void u256_fmadd(U256& acc, uint64_t w, const uint32_t (&m)[4]) {
    const uint32_t w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
    const uint32_t wl[2] = {w0, w1};
    uint32_t prod[LIMBS + 2];
#pragma unroll
    for (int i = 0; i < LIMBS + 2; ++i) prod[i] = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < LIMBS; ++j) {
            const uint64_t v = (uint64_t)wl[i] * m[j] + prod[i + j] + carry;
            prod[i + j] = (uint32_t)v;
            carry = v >> 32;
        }
        prod[i + LIMBS] = (uint32_t)carry;
    }
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc.l[i] = __builtin_addc(acc.l[i], i < LIMBS + 2 ? prod[i] : 0u, c, &c);
}
JOLT_HD bool u256_geq(const U256& a, const U256& b) {
#pragma unroll
    for (int i = 7; i >= 0; --i) {
        if (a.l[i] != b.l[i]) return a.l[i] > b.l[i];
    }
    return true;
}
JOLT_HD U256 u256_sub(const U256& a, const U256& b) {
    U256 r;
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = __builtin_subc(a.l[i], b.l[i], br, &br);
    return r;
}
// low 256 bits of a (128-bit magnitude as two u64) * b
JOLT_HD U256 u256_mul_u128(const U256& b, uint64_t a_lo, uint64_t a_hi) {
    const uint32_t a[4] = {(uint32_t)a_lo, (uint32_t)(a_lo >> 32), (uint32_t)a_hi, (uint32_t)(a_hi >> 32)};
    U256 r = u256_zero();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; i + j < 8; ++j) {
            const uint64_t v = (uint64_t)a[i] * b.l[j] + r.l[i + j] + carry;
            r.l[i + j] = (uint32_t)v;
            carry = v >> 32;
        }
    }
    return r;
}

}  // namespace jolt
