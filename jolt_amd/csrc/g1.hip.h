// jolt_amd/csrc/g1.hip.h -- BN254 G1 group law for device and host (y^2 = x^3 + 3 over Fq, a = 0).
//
// Jacobian (X, Y, Z) with ark's conventions: identity <=> Z == 0; layout = ark_bn254::G1Projective, which the
// reference wraps transparently (crates/jolt-crypto/src/ec/bn254/mod.rs:17-24).  Affine points carry (0,0) for
// infinity (not on the curve).  Bucket sums in the MSM hit P+P, P+(-P) and infinity (specs/clean-slate-prover.md:557-563),
// so every formula handles its special cases explicitly; results are the same POINT as the reference's, the
// projective representative is free (serialisation is compressed affine).
#pragma once
#include "field.hip.h"

namespace jolt {

struct G1Affine {
    Fq x, y;
};
struct G1Jac {
    Fq x, y, z;
};
static_assert(sizeof(G1Jac) == 96 && sizeof(G1Affine) == 64, "G1 layouts");

JOLT_HD bool g1_is_identity(const G1Jac& p) { return p.z.is_zero(); }
JOLT_HD bool g1_aff_is_inf(const G1Affine& p) { return p.x.is_zero() && p.y.is_zero(); }
JOLT_HD G1Jac g1_identity() {
    G1Jac r;
    r.x = Fq::one();
    r.y = Fq::one();
    r.z = Fq::zero();
    return r;
}
JOLT_HD G1Jac g1_from_affine(const G1Affine& a) {
    G1Jac r;
    if (g1_aff_is_inf(a)) return g1_identity();
    r.x = a.x;
    r.y = a.y;
    r.z = Fq::one();
    return r;
}
JOLT_HD G1Affine g1_aff_neg(const G1Affine& a) {
    G1Affine r;
    r.x = a.x;
    r.y = neg(a.y);
    return r;
}
JOLT_HD G1Jac g1_neg(const G1Jac& p) {
    G1Jac r = p;
    r.y = neg(p.y);
    return r;
}

// Y^2 = X^3 + 3 Z^6 with canonical coordinates (the identity Z = 0 passes): what a point that crosses the ABI from the caller must satisfy before it is absorbed
JOLT_HD bool g1_is_on_curve(const G1Jac& p) {
    Fq d;
    if (sub_p(d, p.x) == 0 || sub_p(d, p.y) == 0 || sub_p(d, p.z) == 0) return false;  // a coordinate >= q
    if (g1_is_identity(p)) return true;
    const Fq z2 = sqr(p.z), z6 = mul(sqr(z2), z2);
    const Fq three_z6 = add(dbl(z6), z6);
    const Fq lhs = sqr(p.y), rhs = add(mul(sqr(p.x), p.x), three_z6);
    return sub(lhs, rhs).is_zero();
}

// dbl-2009-l
JOLT_HD G1Jac g1_double(const G1Jac& p) {
    if (g1_is_identity(p)) return p;
    Fq A = sqr(p.x), B = sqr(p.y), C = sqr(B);
    Fq D = dbl(sub(sub(sqr(add(p.x, B)), A), C));
    Fq E = add(dbl(A), A);
    Fq F = sqr(E);
    G1Jac r;
    r.x = sub(F, dbl(D));
    r.z = dbl(mul(p.y, p.z));
    r.y = sub(mul(E, sub(D, r.x)), dbl(dbl(dbl(C))));
    return r;
}

// madd-2007-bl: Jacobian + affine
JOLT_HD G1Jac g1_add_mixed(const G1Jac& p, const G1Affine& q) {
    if (g1_aff_is_inf(q)) return p;
    if (g1_is_identity(p)) return g1_from_affine(q);
    Fq Z1Z1 = sqr(p.z);
    Fq U2 = mul(q.x, Z1Z1);
    Fq S2 = mul(mul(q.y, p.z), Z1Z1);
    if (p.x == U2) {
        if (p.y == S2) return g1_double(p);
        return g1_identity();
    }
    Fq H = sub(U2, p.x);
    Fq HH = sqr(H);
    Fq I = dbl(dbl(HH));
    Fq J = mul(H, I);
    Fq rr = dbl(sub(S2, p.y));
    Fq V = mul(p.x, I);
    G1Jac r;
    r.x = sub(sub(sqr(rr), J), dbl(V));
    r.y = sub(mul(rr, sub(V, r.x)), dbl(mul(p.y, J)));
    r.z = sub(sub(sqr(add(p.z, H)), Z1Z1), HH);
    return r;
}

// ---- XYZZ accumulator for long chains of mixed additions (bucket sums): x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2; identity: ZZ = 0 ----
// madd-2008-s costs 8M + 2S and 6 additions where madd-2007-bl (above) costs 7M + 4S and 14: one multiply and eight modular
// additions / doublings fewer per point, for one more coordinate in registers.  Only the accumulator of a bucket is kept in this form;
// it leaves as a Jacobian point (g1x_to_jac: 2M + 2S).
struct G1Xyzz {
    Fq x, y, zz, zzz;
};
JOLT_HD G1Xyzz g1x_identity() {
    G1Xyzz r;
    r.x = Fq::zero();
    r.y = Fq::zero();
    r.zz = Fq::zero();
    r.zzz = Fq::zero();
    return r;
}
JOLT_HD bool g1x_is_identity(const G1Xyzz& p) { return p.zz.is_zero(); }
JOLT_HD G1Xyzz g1x_from_jac(const G1Jac& p) {
    G1Xyzz r;
    r.x = p.x;
    r.y = p.y;
    r.zz = sqr(p.z);
    r.zzz = mul(r.zz, p.z);
    return r;
}
// (X, Y, ZZ, ZZZ) ~ Jacobian (X * ZZ^2, Y * ZZZ^2, ZZZ): scaling the true Z = ZZZ / ZZ by ZZ
JOLT_HD G1Jac g1x_to_jac(const G1Xyzz& p) {
    if (g1x_is_identity(p)) return g1_identity();
    G1Jac r;
    r.x = mul(p.x, sqr(p.zz));
    r.y = mul(p.y, sqr(p.zzz));
    r.z = p.zzz;
    return r;
}
JOLT_HD G1Xyzz g1x_add_mixed(const G1Xyzz& p, const G1Affine& q) {
    if (g1_aff_is_inf(q)) return p;
    if (g1x_is_identity(p)) {
        G1Xyzz r;
        r.x = q.x;
        r.y = q.y;
        r.zz = Fq::one();
        r.zzz = Fq::one();
        return r;
    }
    const Fq U2 = mul(q.x, p.zz);
    const Fq S2 = mul(q.y, p.zzz);
    const Fq P = sub(U2, p.x);
    const Fq R = sub(S2, p.y);
    if (P.is_zero()) {
        if (R.is_zero()) return g1x_from_jac(g1_double(g1_from_affine(q)));  // the same point twice
        return g1x_identity();                                            // P + (-P)
    }
    const Fq PP = sqr(P);
    const Fq PPP = mul(P, PP);
    const Fq Q = mul(p.x, PP);
    G1Xyzz r;
    r.x = sub(sub(sqr(R), PPP), dbl(Q));
    r.y = sub(mul(R, sub(Q, r.x)), mul(p.y, PPP));
    r.zz = mul(p.zz, PP);
    r.zzz = mul(p.zzz, PPP);
    return r;
}

// add-2007-bl: Jacobian + Jacobian
JOLT_HD G1Jac g1_add(const G1Jac& p, const G1Jac& q) {
    if (g1_is_identity(p)) return q;
    if (g1_is_identity(q)) return p;
    Fq Z1Z1 = sqr(p.z), Z2Z2 = sqr(q.z);
    Fq U1 = mul(p.x, Z2Z2), U2 = mul(q.x, Z1Z1);
    Fq S1 = mul(mul(p.y, q.z), Z2Z2), S2 = mul(mul(q.y, p.z), Z1Z1);
    if (U1 == U2) {
        if (S1 == S2) return g1_double(p);
        return g1_identity();
    }
    Fq H = sub(U2, U1);
    Fq I = sqr(dbl(H));
    Fq J = mul(H, I);
    Fq rr = dbl(sub(S2, S1));
    Fq V = mul(U1, I);
    G1Jac r;
    r.x = sub(sub(sqr(rr), J), dbl(V));
    r.y = sub(mul(rr, sub(V, r.x)), dbl(mul(S1, J)));
    r.z = mul(sub(sub(sqr(add(p.z, q.z)), Z1Z1), Z2Z2), H);
    return r;
}

JOLT_HD G1Affine g1_to_affine(const G1Jac& p) {
    G1Affine a;
    if (g1_is_identity(p)) {
        a.x = Fq::zero();
        a.y = Fq::zero();
        return a;
    }
    Fq zi = inv(p.z);
    Fq zi2 = sqr(zi);
    a.x = mul(p.x, zi2);
    a.y = mul(p.y, mul(zi2, zi));
    return a;
}

// equality as group elements
JOLT_HD bool g1_eq(const G1Jac& p, const G1Jac& q) {
    bool pi = g1_is_identity(p), qi = g1_is_identity(q);
    if (pi || qi) return pi && qi;
    Fq Z1Z1 = sqr(p.z), Z2Z2 = sqr(q.z);
    if (mul(p.x, Z2Z2) != mul(q.x, Z1Z1)) return false;
    return mul(p.y, mul(Z2Z2, q.z)) == mul(q.y, mul(Z1Z1, p.z));
}

// k * p for a small non-negative integer k (bucket weights in the window reduction)
JOLT_HD G1Jac g1_mul_small(const G1Jac& p, uint32_t k) {
    G1Jac acc = g1_identity();
    if (k == 0) return acc;
    int top = 31;
    while (!((k >> top) & 1)) --top;
    for (int i = top; i >= 0; --i) {
        acc = g1_double(acc);
        if ((k >> i) & 1) acc = g1_add(acc, p);
    }
    return acc;
}

// scalar * p, scalar given as a canonical 256-bit integer (8 x u32), MSB-first double-and-add
// (JoltGroup::scalar_mul, crates/jolt-crypto/src/ec/bn254/mod.rs:190-193)
JOLT_HD G1Jac g1_mul_canonical(const G1Jac& p, const uint32_t k[8]) {
    G1Jac acc = g1_identity();
    for (int i = 255; i >= 0; --i) {
        acc = g1_double(acc);
        if ((k[i / 32] >> (i % 32)) & 1) acc = g1_add(acc, p);
    }
    return acc;
}

}  // namespace jolt
