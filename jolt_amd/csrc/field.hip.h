// jolt_amd/csrc/field.hip.h -- BN254 Fr / Fq arithmetic for gfx950 (and for the host-side mirror code).
//
// Representation: 8 x u32 little-endian limbs holding a*R mod p, R = 2^256, always canonical (< p).  The bytes are
// identical to the reference's `Fr` (4 x u64 Montgomery limbs, crates/jolt-field/src/bn254/mod.rs:33-43), so tables
// cross the C ABI without conversion.  The products are built from v_mad_u64_u32 (32x32+64 -> 64): CDNA4 has no
// 64x64 multiplier, and no MFMA is used anywhere (integer prime-field work, not a dense contraction).
//
// Replaces: Fr add/sub/mul/neg (crates/jolt-field/src/bn254/mod.rs:76-83 -> ark-ff Fp), the Montgomery REDC
// (crates/jolt-field/src/bn254/mont.rs:186-238) and Fq for G1 coordinates (crates/jolt-crypto/src/ec/bn254/mod.rs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bn254_constants.hip.h"

#define JOLT_HD __host__ __device__ __forceinline__

namespace jolt {

struct FrParams {
    static constexpr uint32_t P[8] = FR32_P_LIMBS;
    static constexpr uint32_t R[8] = FR32_R_LIMBS;    // Montgomery one
    static constexpr uint32_t R2[8] = FR32_R2_LIMBS;  // R^2 mod p
    static constexpr uint32_t INV = FR32_INV;         // -p^-1 mod 2^32
};
struct FqParams {
    static constexpr uint32_t P[8] = FQ32_P_LIMBS;
    static constexpr uint32_t R[8] = FQ32_R_LIMBS;
    static constexpr uint32_t R2[8] = FQ32_R2_LIMBS;
    static constexpr uint32_t INV = FQ32_INV;
};

template <class PR>
struct alignas(16) Fp {
    uint32_t l[8];

    static JOLT_HD Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.l[i] = 0;
        return r;
    }
    static JOLT_HD Fp one() {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.l[i] = PR::R[i];
        return r;
    }
    static JOLT_HD Fp r2() {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.l[i] = PR::R2[i];
        return r;
    }
    JOLT_HD bool is_zero() const {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc |= l[i];
        return acc == 0;
    }
    JOLT_HD bool operator==(const Fp& o) const {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc |= l[i] ^ o.l[i];
        return acc == 0;
    }
    JOLT_HD bool operator!=(const Fp& o) const { return !(*this == o); }
};

// ---- 256-bit helpers ------------------------------------------------------------------------------------
// Carries go through __builtin_addc/__builtin_subc so that hipcc emits v_add_co/v_addc_co chains.
template <class PR>
JOLT_HD uint32_t add256(Fp<PR>& r, const Fp<PR>& a, const Fp<PR>& b) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = __builtin_addc(a.l[i], b.l[i], c, &c);
    return c;
}
// r = a - b, returns borrow-out (0/1)
template <class PR>
JOLT_HD uint32_t sub256(Fp<PR>& r, const Fp<PR>& a, const Fp<PR>& b) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = __builtin_subc(a.l[i], b.l[i], c, &c);
    return c;
}
// r = a - P, returns borrow (1 iff a < P)
template <class PR>
JOLT_HD uint32_t sub_p(Fp<PR>& r, const Fp<PR>& a) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = __builtin_subc(a.l[i], (uint32_t)PR::P[i], c, &c);
    return c;
}
// branch-free select (per limb v_cndmask)
template <class PR>
JOLT_HD Fp<PR> select(bool take_a, const Fp<PR>& a, const Fp<PR>& b) {
    Fp<PR> r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = take_a ? a.l[i] : b.l[i];
    return r;
}
// canonicalise a value known to be < 2P (carry = bit 256 of the value)
template <class PR>
JOLT_HD Fp<PR> reduce_once(const Fp<PR>& a, uint32_t carry = 0) {
    Fp<PR> d;
    uint32_t borrow = sub_p(d, a);
    return select(carry == 0 && borrow != 0, a, d);
}

template <class PR>
JOLT_HD Fp<PR> add(const Fp<PR>& a, const Fp<PR>& b) {
    Fp<PR> s;
    (void)add256(s, a, b);  // both moduli leave >= 2 spare bits in 256: the sum never carries out
    return reduce_once(s);
}
template <class PR>
JOLT_HD Fp<PR> sub(const Fp<PR>& a, const Fp<PR>& b) {
    Fp<PR> d, e;
    uint32_t borrow = sub256(d, a, b);
    uint32_t mask = 0u - borrow;  // add P back when a < b
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) e.l[i] = __builtin_addc(d.l[i], (uint32_t)PR::P[i] & mask, c, &c);
    return e;
}
template <class PR>
JOLT_HD Fp<PR> neg(const Fp<PR>& a) {
    Fp<PR> z = Fp<PR>::zero();
    return sub(z, a);
}
template <class PR>
JOLT_HD Fp<PR> dbl(const Fp<PR>& a) { return add(a, a); }

// ---- Montgomery multiplication ------------------------------------------------------------------------------
// CIOS over 32-bit limbs, arranged so that each row is 8 independent v_mad_u64_u32 (a_j*b_i + t_j, which cannot
// overflow 64 bits) followed by ONE v_addc carry chain that shifts the high halves up a limb; the reduction row
// (m*P_j + u_j) has the same shape.  NROWS = 8 is a full multiplication; NROWS = 4 serves operands whose four low
// limbs are zero (see mul_shifted).
template <class PR, int NROWS>
JOLT_HD Fp<PR> mont_rows(const Fp<PR>& a, const uint32_t* b) {
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < NROWS; ++i) {
        uint64_t p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = (uint64_t)a.l[j] * b[i] + t[j];
        uint32_t c = 0, u[10];
        u[0] = (uint32_t)p[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) u[j] = __builtin_addc((uint32_t)p[j], (uint32_t)(p[j - 1] >> 32), c, &c);
        u[8] = __builtin_addc(t[8], (uint32_t)(p[7] >> 32), c, &c);
        u[9] = c;
        uint32_t m = u[0] * PR::INV;
        uint64_t q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = (uint64_t)m * (uint32_t)PR::P[j] + u[j];
        c = 0;
#pragma unroll
        for (int j = 1; j < 8; ++j) t[j - 1] = __builtin_addc((uint32_t)q[j], (uint32_t)(q[j - 1] >> 32), c, &c);
        t[7] = __builtin_addc(u[8], (uint32_t)(q[7] >> 32), c, &c);
        t[8] = u[9] + c;
    }
    Fp<PR> r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = t[i];
    return reduce_once(r, t[8]);
}

#if !defined(__HIP_DEVICE_COMPILE__)
// Host-side multiplication (transcript/round-message assembly, the MSM's final Horner): the same Montgomery product over
// 4 x u64 limbs with 128-bit partial products -- x86 has a 64x64->128 multiplier, the GPU does not.  Same canonical result.
template <class PR>
constexpr uint64_t host_neg_inv64() {
    uint64_t p0 = (uint64_t)PR::P[0] | ((uint64_t)PR::P[1] << 32);
    uint64_t y = 1;  // Newton: y <- y * (2 - p0 * y) doubles the number of correct low bits
    for (int i = 0; i < 6; ++i) y *= 2 - p0 * y;
    return 0 - y;
}
template <class PR>
inline Fp<PR> host_mul64(const Fp<PR>& a, const Fp<PR>& b) {
    typedef unsigned __int128 u128;
    constexpr uint64_t NINV = host_neg_inv64<PR>();
    uint64_t A[4], B[4], P[4], t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        A[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
        B[i] = (uint64_t)b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << 32);
        P[i] = (uint64_t)PR::P[2 * i] | ((uint64_t)PR::P[2 * i + 1] << 32);
    }
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (u128)A[j] * B[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * NINV;
        c = ((u128)m * P[0] + t[0]) >> 64;
        for (int j = 1; j < 4; ++j) {
            c += (u128)m * P[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    Fp<PR> r;
    for (int i = 0; i < 4; ++i) {
        r.l[2 * i] = (uint32_t)t[i];
        r.l[2 * i + 1] = (uint32_t)(t[i] >> 32);
    }
    return reduce_once(r, (uint32_t)t[4]);
}
#endif

// ---- device multiplication: product scanning over nine 29-bit limbs ------------------------------------------------------
// The row-wise form above spends most of its instructions on carry handling: per 256-bit product 128 v_mad_u64_u32 but also
// ~140 v_addc, ~135 v_mov (zeroing the high half of every 64-bit addend) and ~150 s_nop.  In radix 2^29 a whole product-scanning
// column -- at most 9 partial products a_i*b_j plus 9 reduction products m_i*p_j, each below 2^58 -- fits the 64-bit accumulator
// of v_mad_u64_u32, so a column is a bare chain of mads through the accumulator: no carries, no moves (162 mads + ~110 other
// instructions; measured 1.22x the throughput of the row-wise form in a mul+add chain, same canonical result).
// The limb structure's Montgomery radix is 2^(9*29) = 2^261, not 2^256: feeding b << 5 squares that away
// (a * 32b * 2^-261 = a * b * 2^-256); 32b < 2^259 still fits nine limbs and the output stays below 2p.
constexpr uint32_t kMask29 = (1u << 29) - 1;
template <class PR>
struct Limbs29 {
    static constexpr uint32_t limb(int k) {  // bits [29k, 29k+29) of p
        const int bit = 29 * k, w = bit >> 5, s = bit & 31;
        uint64_t lo = w < 8 ? (uint64_t)PR::P[w] : 0, hi = w + 1 < 8 ? (uint64_t)PR::P[w + 1] : 0;
        return (uint32_t)(((lo | (hi << 32)) >> s) & kMask29);
    }
    static constexpr uint32_t neg_inv() {  // -p^-1 mod 2^29 (Newton on the low limb)
        uint32_t p0 = limb(0), y = 1;
        for (int i = 0; i < 5; ++i) y *= 2u - p0 * y;
        return (0u - y) & kMask29;
    }
};
// limbs of (x << SH) in radix 2^29 (x < 2^256, SH <= 5: nine limbs)
template <int SH>
JOLT_HD void to_limbs29(const uint32_t (&x)[8], uint32_t (&o)[9]) {
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int bit = 29 * k - SH;  // first source bit of limb k
        uint32_t v;
        if (bit < 0) {
            v = x[0] << (-bit);  // k = 0 with SH > 0
        } else {
            const int w = bit >> 5, s = bit & 31;
            const uint32_t lo = w < 8 ? x[w] : 0u, hi = w + 1 < 8 ? x[w + 1] : 0u;
            v = s == 0 ? lo : (uint32_t)((((uint64_t)hi << 32) | lo) >> s);  // one v_alignbit_b32
        }
        o[k] = v & kMask29;
    }
}
template <class PR>
JOLT_HD Fp<PR> mul_limbs29(const Fp<PR>& a, const Fp<PR>& b) {
    constexpr uint32_t P0 = Limbs29<PR>::limb(0), P1 = Limbs29<PR>::limb(1), P2 = Limbs29<PR>::limb(2), P3 = Limbs29<PR>::limb(3), P4 = Limbs29<PR>::limb(4),
                       P5 = Limbs29<PR>::limb(5), P6 = Limbs29<PR>::limb(6), P7 = Limbs29<PR>::limb(7), P8 = Limbs29<PR>::limb(8);
    constexpr uint32_t PL[9] = {P0, P1, P2, P3, P4, P5, P6, P7, P8};
    constexpr uint32_t NINV = Limbs29<PR>::neg_inv();
    uint32_t A[9], B[9], M[9], R[9];
    to_limbs29<0>(a.l, A);
    to_limbs29<5>(b.l, B);
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (uint64_t)A[i] * B[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)M[i] * PL[k - i];
        M[k] = ((uint32_t)acc * NINV) & kMask29;
        acc += (uint64_t)M[k] * PL[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 18; ++k) {
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc += (uint64_t)A[i] * B[k - i];
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc += (uint64_t)M[i] * PL[k - i];
        R[k - 9] = (uint32_t)acc & kMask29;
        acc >>= 29;
    }
    Fp<PR> out;  // nine 29-bit limbs of a value below 2p < 2^256 -> eight 32-bit words
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = (32 * j) / 29, s = 32 * j - 29 * k;  // word j starts inside limb k at bit s
        uint32_t v = R[k] >> s;
        v |= R[k + 1] << (29 - s);
        if (58 - s < 32 && k + 2 < 9) v |= R[k + 2] << (58 - s);
        out.l[j] = v;
    }
    return reduce_once(out, 0u);
}

template <class PR>
JOLT_HD Fp<PR> mul(const Fp<PR>& a, const Fp<PR>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return mul_limbs29<PR>(a, b);
#else
    return host_mul64(a, b);
#endif
}
template <class PR>
JOLT_HD Fp<PR> sqr(const Fp<PR>& a) { return mul(a, a); }

// a * c where c's Montgomery representation has its four low u32 limbs equal to zero -- the shape of every
// sumcheck challenge (crates/jolt-field/src/bn254/mod.rs:172-184,254: limbs [0,0,low,high]).
// With c = c' * 2^128:  a*c*2^-256 = a*c'*2^-128, so only an 8x4 product and FOUR reduction rows are needed:
// half the multiplies of `mul`, same canonical result.  `chi` = the four high limbs of c.
template <class PR>
JOLT_HD Fp<PR> mul_shifted(const Fp<PR>& a, const uint32_t chi[4]) { return mont_rows<PR, 4>(a, chi); }

// ---- deferred-reduction accumulator ---------------------------------------------------------------------------------------
// The device analogue of WideAccumulator / FrSignedProductAccumulator (crates/jolt-field/src/bn254/mont.rs:334-602, trait
// crates/jolt-field/src/algebra.rs:362-433): sum_k a_k*b_k is accumulated as an UNREDUCED 512-bit integer (fmadd = the 8x8 product
// rows only, half the multiply-adds of a Montgomery product) and reduced once (one REDC).  Same canonical value as the sum of the
// reduced products -- deferred reduction is exact mod p.  Headroom: every product is < p^2 < 2^508 and the REDC output is
// < (0.19 n + 1) p for n products, so `reduce` (four conditional subtractions) is valid for n <= kWideMaxProducts.
// Measured on gfx950: NOT a win inside the round kernels (the carry ripples of the 17-limb accumulator and its registers cost
// more than the REDC rows it saves -- v_mad_u64_u32 is cheap next to the carry chains), so the kernels keep plain field sums and
// this stays the tested restatement of the accumulator contract (jolt_host_fr_wide_dot).
constexpr int kWideMaxProducts = 20;
template <class PR>
struct WideAcc {
    uint32_t l[17];
};
template <class PR>
JOLT_HD WideAcc<PR> wide_zero() {
    WideAcc<PR> w;
#pragma unroll
    for (int i = 0; i < 17; ++i) w.l[i] = 0;
    return w;
}
template <class PR>
JOLT_HD void wide_fmadd(WideAcc<PR>& acc, const Fp<PR>& a, const Fp<PR>& b) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = (uint64_t)a.l[j] * b.l[i] + acc.l[i + j];
        uint32_t c = 0;
        acc.l[i] = (uint32_t)p[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) acc.l[i + j] = __builtin_addc((uint32_t)p[j], (uint32_t)(p[j - 1] >> 32), c, &c);
        acc.l[i + 8] = __builtin_addc(acc.l[i + 8], (uint32_t)(p[7] >> 32), c, &c);
#pragma unroll
        for (int k = i + 9; k < 17; ++k) acc.l[k] = __builtin_addc(acc.l[k], 0u, c, &c);
    }
}
template <class PR>
JOLT_HD Fp<PR> wide_reduce(const WideAcc<PR>& acc) {
    uint32_t t[18];
#pragma unroll
    for (int i = 0; i < 17; ++i) t[i] = acc.l[i];
    t[17] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t m = t[i] * PR::INV;
        uint64_t q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = (uint64_t)m * (uint32_t)PR::P[j] + t[i + j];
        uint32_t c = 0;
#pragma unroll
        for (int j = 1; j < 8; ++j) t[i + j] = __builtin_addc((uint32_t)q[j], (uint32_t)(q[j - 1] >> 32), c, &c);
        t[i + 8] = __builtin_addc(t[i + 8], (uint32_t)(q[7] >> 32), c, &c);
#pragma unroll
        for (int k = i + 9; k < 18; ++k) t[k] = __builtin_addc(t[k], 0u, c, &c);
    }
    Fp<PR> r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = t[8 + i];
    uint32_t top = t[16];  // bits above 2^256 of the REDC output (t[17] stays zero within the documented headroom)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        Fp<PR> d;
        const uint32_t borrow = sub_p(d, r);
        const bool take = top != 0 || borrow == 0;
        r = select(take, d, r);
        top = take ? top - borrow : top;
    }
    return r;
}

// Montgomery form -> canonical integer (REDC of the bare limbs) and back
template <class PR>
JOLT_HD Fp<PR> from_mont(const Fp<PR>& a) {
    Fp<PR> one_int = Fp<PR>::zero();
    one_int.l[0] = 1;
    return mul(a, one_int);
}
template <class PR>
JOLT_HD Fp<PR> to_mont(const Fp<PR>& a) { return mul(a, Fp<PR>::r2()); }

// a^e, e a canonical 256-bit integer (host-side use: inversions in round-message assembly)
template <class PR>
JOLT_HD Fp<PR> pow(const Fp<PR>& a, const uint32_t e[8]) {
    Fp<PR> acc = Fp<PR>::one();
    for (int i = 255; i >= 0; --i) {
        acc = sqr(acc);
        if ((e[i / 32] >> (i % 32)) & 1) acc = mul(acc, a);
    }
    return acc;
}
// Fermat inverse; zero maps to zero (the reference returns None: callers check is_zero first)
template <class PR>
JOLT_HD Fp<PR> inv(const Fp<PR>& a) {
    uint32_t e[8];
    // p - 2
    int64_t c = -2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c += (int64_t)PR::P[i];
        e[i] = (uint32_t)c;
        c >>= 32;
    }
    return pow(a, e);
}

using Fr = Fp<FrParams>;
using Fq = Fp<FqParams>;

// small-integer -> Fr (Montgomery form): crates/jolt-field/src/bn254/mont.rs:307-315 (value == ark Fr::from(n))
JOLT_HD Fr fr_from_u64(uint64_t n) {
    Fr v = Fr::zero();
    v.l[0] = (uint32_t)n;
    v.l[1] = (uint32_t)(n >> 32);
    return to_mont(v);
}

}  // namespace jolt
