// jolt_amd/csrc/fq_limb.hip.h -- Fq in "limb form" for long chains of dependent products (the bucket sums of the fixed-base MSM).
//
// `mul` of field.hip.h unpacks both operands into nine 29-bit limbs, runs the product-scanning columns, packs the result back into
// eight 32-bit words and conditionally subtracts p: of its ~272 instructions only the 162 multiply-adds are arithmetic.  PMC on the
// bucket kernel shows the VALU saturated (waves wait to ISSUE half of their cycles), so the only lever left is instruction count.
// Here a value STAYS in limb form between operations:
//   * FqL = nine 29-bit limbs of a representative in [0, 2p) (lazy reduction: no conditional subtraction after a product --
//     (a b + m p) / 2^261 < 1.03 p for a, b < 2p, so the range is closed under mulL);
//   * the Montgomery radix is the limb structure's own 2^261: x is held as x * 2^261 mod p ("L-form"), so neither operand needs the
//     5-bit pre-shift of mul_limbs29.  The L-form of x is the STANDARD Montgomery form of 32x -- tables are converted once, when they are
//     built (k_fx_to_lform), and the 32-byte words in memory look like any other Fq;
//   * squarings use the symmetry of their columns (45 instead of 81 partial products).
// mulL ~ 210 instructions, sqrL ~ 175, add / sub with full carry normalisation ~ 50.  Every function also compiles for the host
// (jolt_host_fq_limb_* / jolt_host_g1_sum_limb_form in host_mirror.hip), where the CPU suite pins it against the oracle.
#pragma once
#include "field.hip.h"

namespace jolt {

struct FqL {
    uint32_t l[9];
};

namespace fql {
constexpr uint32_t P(int k) { return Limbs29<FqParams>::limb(k); }
// limbs of 2p (p < 2^254: 2p fits nine limbs)
constexpr uint32_t P2(int k) {
    uint64_t carry = 0, v = 0;
    for (int i = 0; i <= k; ++i) {
        v = 2ull * P(i) + carry;
        carry = v >> 29;
        v &= kMask29;
    }
    return (uint32_t)v;
}
constexpr uint32_t NINV = Limbs29<FqParams>::neg_inv();
// limb k of m * p (m <= 9: below 2^258, the top limb keeps what is left)
constexpr uint32_t PM(int m, int k) {
    uint64_t carry = 0, v = 0;
    for (int i = 0; i <= k; ++i) {
        v = (uint64_t)m * P(i) + carry;
        carry = v >> 29;
        if (i < 8) v &= kMask29;
    }
    return (uint32_t)v;
}
}  // namespace fql
#define JOLT_FQL_MP(m) {fql::PM(m, 0), fql::PM(m, 1), fql::PM(m, 2), fql::PM(m, 3), fql::PM(m, 4), fql::PM(m, 5), fql::PM(m, 6), fql::PM(m, 7), fql::PM(m, 8)}
// function-local constexpr tables (indexed by unrolled constants: folded into immediates)
#define JOLT_FQL_P {fql::P(0), fql::P(1), fql::P(2), fql::P(3), fql::P(4), fql::P(5), fql::P(6), fql::P(7), fql::P(8)}
#define JOLT_FQL_2P {fql::P2(0), fql::P2(1), fql::P2(2), fql::P2(3), fql::P2(4), fql::P2(5), fql::P2(6), fql::P2(7), fql::P2(8)}

JOLT_HD FqL fql_from_words(const Fq& a) {  // the words already hold the value's L-form (or any value < 2^256 to be taken as is)
    FqL r;
    to_limbs29<0>(a.l, r.l);
    return r;
}
JOLT_HD FqL fql_zero() {
    FqL r;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.l[k] = 0;
    return r;
}
JOLT_HD bool fql_is_zero(const FqL& a) {  // a in [0, 2p): zero mod p iff a == 0 or a == p
    constexpr uint32_t PL[9] = JOLT_FQL_P;
    uint32_t any = 0, diff = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        any |= a.l[k];
        diff |= a.l[k] ^ PL[k];
    }
    return any == 0 || diff == 0;
}
// a + b, both < 2p, result < 2p
JOLT_HD FqL fql_add(const FqL& a, const FqL& b) {
    constexpr uint32_t P2L[9] = JOLT_FQL_2P;
    uint32_t s[9], d[9];
    uint32_t carry = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const uint32_t t = a.l[k] + b.l[k] + carry;
        s[k] = t & kMask29;
        carry = t >> 29;
    }
    int32_t borrow = 0;  // d = s - 2p
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int32_t t = (int32_t)s[k] - (int32_t)P2L[k] + borrow;
        d[k] = (uint32_t)t & kMask29;
        borrow = t >> 29;  // 0 or -1 (arithmetic shift)
    }
    FqL r;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.l[k] = borrow ? s[k] : d[k];
    return r;
}
// a - b, both < 2p, result < 2p
JOLT_HD FqL fql_sub(const FqL& a, const FqL& b) {
    constexpr uint32_t P2L[9] = JOLT_FQL_2P;
    uint32_t d[9], e[9];
    int32_t borrow = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int32_t t = (int32_t)a.l[k] - (int32_t)b.l[k] + borrow;
        d[k] = (uint32_t)t & kMask29;
        borrow = t >> 29;
    }
    uint32_t carry = 0;  // e = d + 2p (mod 2^261)
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const uint32_t t = d[k] + P2L[k] + carry;
        e[k] = t & kMask29;
        carry = t >> 29;
    }
    FqL r;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.l[k] = borrow ? e[k] : d[k];
    return r;
}
JOLT_HD FqL fql_dbl(const FqL& a) { return fql_add(a, a); }

// a * b * 2^-261 mod p; a, b < 2p in normalised limbs; result < 1.03 p, normalised
JOLT_HD FqL fql_mul(const FqL& a, const FqL& b) {
    constexpr uint32_t PL[9] = JOLT_FQL_P;
    uint32_t M[9];
    FqL r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)M[i] * PL[k - i];
        M[k] = ((uint32_t)acc * fql::NINV) & kMask29;
        acc += (uint64_t)M[k] * PL[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 18; ++k) {
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc += (uint64_t)M[i] * PL[k - i];
        r.l[k - 9] = (uint32_t)acc & kMask29;
        acc >>= 29;
    }
    return r;
}
// (a b + c d) * 2^-261: both products through the same columns and ONE reduction (81 multiply-adds saved against two fql_mul and a
// difference); a b + c d < 169 p^2 as for fql_mul.  A column holds <= 18 partial products < 2^58 and 9 reduction products: < 2^63.
JOLT_HD FqL fql_mul2(const FqL& a, const FqL& b, const FqL& c, const FqL& d) {
    constexpr uint32_t PL[9] = JOLT_FQL_P;
    uint32_t M[9];
    FqL r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) {
            acc += (uint64_t)a.l[i] * b.l[k - i];
            acc += (uint64_t)c.l[i] * d.l[k - i];
        }
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)M[i] * PL[k - i];
        M[k] = ((uint32_t)acc * fql::NINV) & kMask29;
        acc += (uint64_t)M[k] * PL[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 18; ++k) {
#pragma unroll
        for (int i = k - 8; i < 9; ++i) {
            acc += (uint64_t)a.l[i] * b.l[k - i];
            acc += (uint64_t)c.l[i] * d.l[k - i];
        }
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc += (uint64_t)M[i] * PL[k - i];
        r.l[k - 9] = (uint32_t)acc & kMask29;
        acc >>= 29;
    }
    return r;
}
// a^2 * 2^-261: the off-diagonal products once, against the doubled limbs (2 a_j < 2^30: a column of <= 5 products < 2^59 each still fits)
JOLT_HD FqL fql_sqr(const FqL& a) {
    constexpr uint32_t PL[9] = JOLT_FQL_P;
    uint32_t M[9], a2[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) a2[k] = a.l[k] << 1;
    FqL r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; 2 * i < k; ++i) acc += (uint64_t)a.l[i] * a2[k - i];
        if (k % 2 == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)M[i] * PL[k - i];
        M[k] = ((uint32_t)acc * fql::NINV) & kMask29;
        acc += (uint64_t)M[k] * PL[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 18; ++k) {
#pragma unroll
        for (int i = k - 8; 2 * i < k; ++i) acc += (uint64_t)a.l[i] * a2[k - i];
        if (k % 2 == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc += (uint64_t)M[i] * PL[k - i];
        r.l[k - 9] = (uint32_t)acc & kMask29;
        acc >>= 29;
    }
    return r;
}
// L-form -> the standard Montgomery words of the same field element, canonical: x 2^261 * (2^256 mod p) * 2^-261 = x 2^256.
// `r256` = the limbs of 2^256 mod p (= the words of Fq::one())
JOLT_HD Fq fql_to_std(const FqL& a, const FqL& r256) {
    const FqL v = fql_mul(a, r256);  // < 1.03 p
    Fq out;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = (32 * j) / 29, s = 32 * j - 29 * k;
        uint32_t w = v.l[k] >> s;
        w |= v.l[k + 1] << (29 - s);
        if (58 - s < 32 && k + 2 < 9) w |= v.l[k + 2] << (58 - s);
        out.l[j] = w;
    }
    return reduce_once(out, 0u);
}

// ---- lazily reduced differences: ONE carry pass, no comparison ------------------------------------------------------------------
// fql_mul / fql_sqr accept any operands a, b with a b < 169 p^2 (output < p (a b / (p 2^261) + 1) < 2p) as long as the limbs are
// normalised, so a difference does not have to come back into [0, 2p): a + M p - b with M p >= b is positive and simply a few p larger.
// g1xl_add_mixed tracks the ranges (comments there).  a + M p - b - 2 c in one pass (c may be absent).
template <int M, bool WITH_C>
JOLT_HD FqL fql_diff(const FqL& a, const FqL& b, const FqL& c) {
    constexpr uint32_t MP[9] = JOLT_FQL_MP(M);
    FqL r;
    int32_t carry = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        int32_t t = (int32_t)a.l[k] + (int32_t)MP[k] - (int32_t)b.l[k] + carry;  // every term below 2^29 (the top limbs far below): no overflow
        if (WITH_C) t -= (int32_t)(c.l[k] << 1);
        r.l[k] = k < 8 ? ((uint32_t)t & kMask29) : (uint32_t)t;  // the value is positive: the top limb takes what is left
        carry = t >> 29;
    }
    return r;
}
// is a == m p for some 1 <= m <= MAX (a is known to be below (MAX + 1) p and a multiple check is all that is needed)?
template <int MAX>
JOLT_HD bool fql_is_multiple_of_p(const FqL& a) {
    bool hit = false;
#pragma unroll
    for (int m = 1; m <= MAX; ++m) {
        uint32_t diff = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) diff |= a.l[k] ^ fql::PM(m, k);
        hit = hit || diff == 0;
    }
    return hit;
}

// ---- XYZZ accumulator in limb form (see g1.hip.h for the coordinates) ----------------------------------------------------------
struct G1XyzzL {
    FqL x, y, zz, zzz;
};
JOLT_HD G1XyzzL g1xl_identity() {
    G1XyzzL r;
    r.x = fql_zero();
    r.y = fql_zero();
    r.zz = fql_zero();
    r.zzz = fql_zero();
    return r;
}
JOLT_HD bool g1xl_is_identity(const G1XyzzL& p) { return fql_is_zero(p.zz); }
// 2 * (x, y) for an affine point in L-form (mdbl-2008-s, a = 0): only reached when a bucket receives the same point twice in a row
JOLT_HD G1XyzzL g1xl_double_affine(const FqL& x, const FqL& y) {
    const FqL U = fql_dbl(y);
    const FqL V = fql_sqr(U);
    const FqL W = fql_mul(U, V);
    const FqL S = fql_mul(x, V);
    const FqL xx = fql_sqr(x);
    const FqL Mm = fql_add(fql_dbl(xx), xx);
    G1XyzzL r;
    r.x = fql_sub(fql_sqr(Mm), fql_dbl(S));
    r.y = fql_sub(fql_mul(Mm, fql_sub(S, r.x)), fql_mul(W, y));
    r.zz = V;
    r.zzz = W;
    return r;
}
// madd-2008-s: 8M + 2S; `one` = the L-form of 1.  (qx, qy) must not be the point at infinity (the caller skips (0, 0)).
// Ranges (multiples of p; products are below 1.6 p for every operand pair that occurs): X < 7.6, Y < 3.6, ZZ, ZZZ < 1.6 on entry and on
// exit; P = U2 + 8p - X in (0.4, 9.6), R = S2 + 4p - Y in (0.4, 5.6), X3 = R^2 + 6p - PPP - 2Q in (1.2, 7.6), Q + 8p - X3 in (0.4, 9.6),
// Y3 = (R (Q - X3) + (4p - Y) PPP) 2^-261 < 1.4 (one reduction for both products); the largest product, P^2 < 92.2 p^2, stays below 1.55 p.
// P = 0 mod p (the same x: the point itself or its negative) shows as ZZ3 = ZZ PP = 0 mod p, tested on a product (0 or p) AFTER the
// common path instead of on the lazily reduced P before it.
JOLT_HD G1XyzzL g1xl_add_mixed(const G1XyzzL& p, const FqL& qx, const FqL& qy, const FqL& one) {
    if (g1xl_is_identity(p)) {
        G1XyzzL r;
        r.x = qx;
        r.y = qy;
        r.zz = one;
        r.zzz = one;
        return r;
    }
    const FqL U2 = fql_mul(qx, p.zz);
    const FqL S2 = fql_mul(qy, p.zzz);
    const FqL P = fql_diff<8, false>(U2, p.x, U2);
    const FqL R = fql_diff<4, false>(S2, p.y, S2);
    const FqL PP = fql_sqr(P);
    const FqL PPP = fql_mul(P, PP);
    const FqL Q = fql_mul(p.x, PP);
    G1XyzzL r;
    r.zz = fql_mul(p.zz, PP);
    if (fql_is_zero(r.zz)) {  // P = 0 mod p
        if (fql_is_multiple_of_p<5>(R)) return g1xl_double_affine(qx, qy);  // the same point again
        return g1xl_identity();                                             // its negative
    }
    r.x = fql_diff<6, true>(fql_sqr(R), PPP, Q);
    r.y = fql_mul2(R, fql_diff<8, false>(Q, r.x, Q), fql_diff<4, false>(fql_zero(), p.y, Q), PPP);  // R (Q - X3) + (4p - Y) PPP: 53.8 + 6.4 p^2, below 1.4 p
    r.zzz = fql_mul(p.zzz, PPP);
    return r;
}

}  // namespace jolt
