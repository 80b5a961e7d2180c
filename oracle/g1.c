/*
 * oracle/g1.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of BN254 G1 as the reference uses it through JoltGroup
 * (/root/reference/crates/jolt-crypto/src/ec/group.rs:27-71, ec/bn254/mod.rs:17-24,176-212).
 * The group law itself lives in the un-vendored arkworks fork (ark-ec 0.5.0 @ a16z/arkworks-algebra
 * dev/twist-shout 76bb3a45, Cargo.lock:801); this file restates the public short-Weierstrass Jacobian
 * formulas (a = 0, b = 3, generator (1,2)) with ark's layout: Projective {x,y,z} of Montgomery Fq limbs,
 * identity <=> z == 0.
 *
 * PARITY UNPINNED by vectors: the reference's tests hold no golden G1 points (group_laws.rs
 * checks msm == sum s_i P_i only).  Pinned here by: on-curve checks, group laws, msm_pippenger == naive
 * sum (the reference's own property, crates/jolt-crypto/tests/group_laws.rs:69-78,135-146) and
 * [k]G == known small multiples computed with Python big-int affine arithmetic, plus public known answers -- 2G as in the
 * EIP-196 alt_bn128 vectors, (r-1)G = -G, the compressed form of G (tests/test_oracle_g1.py::test_public_known_answers).
 */
#include "fr.h"
#include <stdlib.h>

#define EXPORT __attribute__((visibility("default")))

typedef struct { fq_t x, y, z; } g1_t;           /* Jacobian, 96 bytes == ark_bn254::G1Projective */
typedef struct { fq_t x, y; } g1_affine_t;       /* affine, (0,0) encodes infinity (not on the curve) */

static inline fq_t QADD(fq_t a, fq_t b) { fq_t o; fq_add(&o, &a, &b); return o; }
static inline fq_t QSUB(fq_t a, fq_t b) { fq_t o; fq_sub(&o, &a, &b); return o; }
static inline fq_t QMUL(fq_t a, fq_t b) { fq_t o; fq_mul(&o, &a, &b); return o; }
static inline fq_t QSQR(fq_t a) { fq_t o; fq_sqr(&o, &a); return o; }
static inline fq_t QDBL(fq_t a) { return QADD(a, a); }
static inline fq_t QNEG(fq_t a) { fq_t o; fq_neg(&o, &a); return o; }

static g1_t g1_identity(void) { g1_t p; p.x = FQ.r; p.y = FQ.r; memset(&p.z, 0, sizeof p.z); return p; }
static int g1_is_identity(const g1_t *p) { return u256_is_zero(&p->z); }
static int aff_is_inf(const g1_affine_t *p) { return u256_is_zero(&p->x) && u256_is_zero(&p->y); }

/* dbl-2009-l (a = 0) */
static g1_t g1_double(const g1_t *p) {
    if (g1_is_identity(p)) return *p;
    fq_t A = QSQR(p->x), B = QSQR(p->y), C = QSQR(B);
    fq_t D = QDBL(QSUB(QSUB(QSQR(QADD(p->x, B)), A), C));
    fq_t E = QADD(QDBL(A), A);
    fq_t F = QSQR(E);
    g1_t r;
    r.x = QSUB(F, QDBL(D));
    r.z = QDBL(QMUL(p->y, p->z));
    fq_t C8 = QDBL(QDBL(QDBL(C)));
    r.y = QSUB(QMUL(E, QSUB(D, r.x)), C8);
    return r;
}

/* add-2007-bl with the special cases made explicit */
static g1_t g1_add(const g1_t *p, const g1_t *q) {
    if (g1_is_identity(p)) return *q;
    if (g1_is_identity(q)) return *p;
    fq_t Z1Z1 = QSQR(p->z), Z2Z2 = QSQR(q->z);
    fq_t U1 = QMUL(p->x, Z2Z2), U2 = QMUL(q->x, Z1Z1);
    fq_t S1 = QMUL(QMUL(p->y, q->z), Z2Z2), S2 = QMUL(QMUL(q->y, p->z), Z1Z1);
    if (u256_eq(&U1, &U2)) {
        if (u256_eq(&S1, &S2)) return g1_double(p);
        return g1_identity();
    }
    fq_t H = QSUB(U2, U1);
    fq_t I = QSQR(QDBL(H));
    fq_t J = QMUL(H, I);
    fq_t rr = QDBL(QSUB(S2, S1));
    fq_t V = QMUL(U1, I);
    g1_t r;
    r.x = QSUB(QSUB(QSQR(rr), J), QDBL(V));
    r.y = QSUB(QMUL(rr, QSUB(V, r.x)), QDBL(QMUL(S1, J)));
    r.z = QMUL(QSUB(QSUB(QSQR(QADD(p->z, q->z)), Z1Z1), Z2Z2), H);
    return r;
}

/* madd-2007-bl (q affine) */
static g1_t g1_add_mixed(const g1_t *p, const g1_affine_t *q) {
    if (aff_is_inf(q)) return *p;
    if (g1_is_identity(p)) { g1_t r; r.x = q->x; r.y = q->y; r.z = FQ.r; return r; }
    fq_t Z1Z1 = QSQR(p->z);
    fq_t U2 = QMUL(q->x, Z1Z1);
    fq_t S2 = QMUL(QMUL(q->y, p->z), Z1Z1);
    if (u256_eq(&p->x, &U2)) {
        if (u256_eq(&p->y, &S2)) return g1_double(p);
        return g1_identity();
    }
    fq_t H = QSUB(U2, p->x);
    fq_t HH = QSQR(H);
    fq_t I = QDBL(QDBL(HH));
    fq_t J = QMUL(H, I);
    fq_t rr = QDBL(QSUB(S2, p->y));
    fq_t V = QMUL(p->x, I);
    g1_t r;
    r.x = QSUB(QSUB(QSQR(rr), J), QDBL(V));
    r.y = QSUB(QMUL(rr, QSUB(V, r.x)), QDBL(QMUL(p->y, J)));
    r.z = QSUB(QSUB(QSQR(QADD(p->z, H)), Z1Z1), HH);
    return r;
}

static g1_t g1_neg(const g1_t *p) { g1_t r = *p; r.y = QNEG(p->y); return r; }

static g1_affine_t g1_to_affine(const g1_t *p) {
    g1_affine_t a;
    if (g1_is_identity(p)) { memset(&a, 0, sizeof a); return a; }
    fq_t zinv;
    fq_inv(&zinv, &p->z);
    fq_t zinv2 = QSQR(zinv);
    a.x = QMUL(p->x, zinv2);
    a.y = QMUL(p->y, QMUL(zinv2, zinv));
    return a;
}

/* equality as group elements (ark Projective PartialEq): cross-multiplied coordinates */
static int g1_eq(const g1_t *p, const g1_t *q) {
    int pi = g1_is_identity(p), qi = g1_is_identity(q);
    if (pi || qi) return pi && qi;
    fq_t Z1Z1 = QSQR(p->z), Z2Z2 = QSQR(q->z);
    fq_t a = QMUL(p->x, Z2Z2), b = QMUL(q->x, Z1Z1);
    if (!u256_eq(&a, &b)) return 0;
    fq_t c = QMUL(p->y, QMUL(Z2Z2, q->z)), d = QMUL(q->y, QMUL(Z1Z1, p->z));
    return u256_eq(&c, &d);
}

/* JoltGroup::scalar_mul (bn254/mod.rs:190-193): scalar is an Fr in Montgomery form; double-and-add over
 * its canonical integer, MSB first */
static g1_t g1_scalar_mul(const g1_t *p, const fr_t *scalar) {
    u256 k;
    mont_to_canonical(&k, scalar, &FR);
    g1_t acc = g1_identity();
    for (int i = 255; i >= 0; --i) {
        acc = g1_double(&acc);
        if ((k.l[i / 64] >> (i % 64)) & 1) acc = g1_add(&acc, p);
    }
    return acc;
}

EXPORT void orc_g1_generator(g1_t *out) {
    fq_t one = FQ.r;
    out->x = one;
    out->y = QADD(one, one);
    out->z = one;
}
EXPORT void orc_g1_identity(g1_t *out) { *out = g1_identity(); }
EXPORT void orc_g1_add(const g1_t *p, const g1_t *q, g1_t *out) { *out = g1_add(p, q); }
EXPORT void orc_g1_double(const g1_t *p, g1_t *out) { *out = g1_double(p); }
EXPORT void orc_g1_neg(const g1_t *p, g1_t *out) { *out = g1_neg(p); }
EXPORT void orc_g1_add_mixed(const g1_t *p, const g1_affine_t *q, g1_t *out) { *out = g1_add_mixed(p, q); }
EXPORT void orc_g1_scalar_mul(const g1_t *p, const fr_t *s, g1_t *out) { *out = g1_scalar_mul(p, s); }
EXPORT int orc_g1_eq(const g1_t *p, const g1_t *q) { return g1_eq(p, q); }
EXPORT int orc_g1_is_identity(const g1_t *p) { return g1_is_identity(p); }
EXPORT void orc_g1_to_affine_vec(const g1_t *p, g1_affine_t *out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = g1_to_affine(&p[i]); }
EXPORT int orc_g1_on_curve(const g1_t *p) {
    if (g1_is_identity(p)) return 1;
    g1_affine_t a = g1_to_affine(p);
    fq_t three = {{3, 0, 0, 0}}, b;
    mont_from_canonical(&b, &three, &FQ);
    fq_t lhs = QSQR(a.y), rhs = QADD(QMUL(QSQR(a.x), a.x), b);
    return u256_eq(&lhs, &rhs);
}

/* Compressed serialization as the reference appends points to transcripts and proofs
 * (bn254/mod.rs:139-171 -> ark-serialize short-Weierstrass compressed): 32-byte LE x with flags in the top
 * two bits of the last byte: 0x80 = y is the lexicographically larger root (y > -y), 0x40 = infinity. */
EXPORT void orc_g1_serialize_compressed(const g1_t *p, uint8_t out[32]) {
    memset(out, 0, 32);
    if (g1_is_identity(p)) { out[31] |= 0x40; return; }
    g1_affine_t a = g1_to_affine(p);
    u256 xc, yc, nyc;
    mont_to_canonical(&xc, &a.x, &FQ);
    mont_to_canonical(&yc, &a.y, &FQ);
    fq_t ny = QNEG(a.y);
    mont_to_canonical(&nyc, &ny, &FQ);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) out[8 * i + j] = (uint8_t)(xc.l[i] >> (8 * j));
    int y_is_negative = !(u256_geq(&nyc, &yc)); /* y <= -y -> positive */
    if (y_is_negative) out[31] |= 0x80;
}

/* JoltGroup::msm contract (group.rs:63-70) evaluated the slow, obviously-correct way */
EXPORT void orc_g1_msm_naive(const g1_t *bases, const fr_t *scalars, size_t n, g1_t *out) {
    g1_t acc = g1_identity();
    for (size_t i = 0; i < n; ++i) {
        g1_t t = g1_scalar_mul(&bases[i], &scalars[i]);
        acc = g1_add(&acc, &t);
    }
    *out = acc;
}

/* Bucket-method MSM standing in for ark_ec::VariableBaseMSM::msm_bigint (bn254/mod.rs:195-212 converts every
 * base to affine and every scalar to its canonical integer first, as done here).  Window size follows ark's
 * rule c = 3 (n < 32) else floor(log2(n)*69/100) + 2.  Same group element as orc_g1_msm_naive. */
EXPORT void orc_g1_msm_pippenger(const g1_t *bases, const fr_t *scalars, size_t n, g1_t *out) {
    if (n == 0) { *out = g1_identity(); return; }
    g1_affine_t *aff = (g1_affine_t *)malloc(n * sizeof(g1_affine_t));
    u256 *ks = (u256 *)malloc(n * sizeof(u256));
    for (size_t i = 0; i < n; ++i) { aff[i] = g1_to_affine(&bases[i]); mont_to_canonical(&ks[i], &scalars[i], &FR); }
    unsigned log2n = 0;
    while (((size_t)2 << log2n) <= n) log2n++;
    unsigned c = n < 32 ? 3 : (log2n * 69 / 100) + 2;
    size_t n_buckets = ((size_t)1 << c) - 1;
    g1_t *buckets = (g1_t *)malloc(n_buckets * sizeof(g1_t));
    unsigned num_bits = 254;
    unsigned n_windows = (num_bits + c - 1) / c;
    g1_t total = g1_identity();
    for (int w = (int)n_windows - 1; w >= 0; --w) {
        for (unsigned k = 0; k < c; ++k) total = g1_double(&total);
        for (size_t b = 0; b < n_buckets; ++b) buckets[b] = g1_identity();
        unsigned start = (unsigned)w * c;
        for (size_t i = 0; i < n; ++i) {
            uint64_t digit = 0;
            for (unsigned k = 0; k < c; ++k) {
                unsigned bit = start + k;
                if (bit < 256) digit |= ((ks[i].l[bit / 64] >> (bit % 64)) & 1) << k;
            }
            if (digit) buckets[digit - 1] = g1_add_mixed(&buckets[digit - 1], &aff[i]);
        }
        g1_t running = g1_identity(), wsum = g1_identity();
        for (size_t b = n_buckets; b-- > 0;) {
            running = g1_add(&running, &buckets[b]);
            wsum = g1_add(&wsum, &running);
        }
        total = g1_add(&total, &wsum);
    }
    free(buckets);
    free(aff);
    free(ks);
    *out = total;
}

/* ---- cpu_baseline leg only: the same bucket method on all host cores ------------------------------------------------------
 * ark's VariableBaseMSM runs its windows on rayon threads; here tasks = (window, chunk of the points), each with its own bucket
 * array, merged by the group law (order-free).  The bases are converted to affine ONCE per base array (cached by address and
 * length) -- the reference converts per call with a batch inversion (bn254/mod.rs:205), cheaper than the n single inversions a
 * per-call conversion would cost here, so leaving it out of the timed calls only flatters the CPU side. */
#include <omp.h>
static const g1_t *g_aff_src = NULL;
static size_t g_aff_n = 0;
static g1_affine_t *g_aff = NULL;
EXPORT void orc_baseline_prepare_bases(const g1_t *bases, size_t n) {
    free(g_aff);
    g_aff = (g1_affine_t *)malloc((n ? n : 1) * sizeof(g1_affine_t));
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) g_aff[i] = g1_to_affine(&bases[i]);
    g_aff_src = bases;
    g_aff_n = n;
}
/* XYZZ accumulators (x = X / ZZ, y = Y / ZZZ, identity <=> ZZ == 0): madd-2008-s costs 8M + 2S against the 7M + 4S of the Jacobian
 * mixed addition and has no doubling-shaped dependency chain; this is the accumulator GPU and CPU bucket MSMs converged on. */
typedef struct { fq_t x, y, zz, zzz; } g1_xyzz_t;
static inline void xyzz_set_identity(g1_xyzz_t *p) { memset(p, 0, sizeof *p); }
static inline int xyzz_is_identity(const g1_xyzz_t *p) { return u256_is_zero(&p->zz); }
/* acc += (qx, +-qy), q affine and not infinity */
static inline void xyzz_add_mixed(g1_xyzz_t *acc, const fq_t *qx, const fq_t *qy_in, int negate) {
    fq_t qy = negate ? QNEG(*qy_in) : *qy_in;
    if (xyzz_is_identity(acc)) { acc->x = *qx; acc->y = qy; acc->zz = FQ.r; acc->zzz = FQ.r; return; }
    fq_t U2 = QMUL(*qx, acc->zz), S2 = QMUL(qy, acc->zzz);
    fq_t P = QSUB(U2, acc->x), R = QSUB(S2, acc->y);
    if (u256_is_zero(&P)) {
        if (!u256_is_zero(&R)) { xyzz_set_identity(acc); return; }
        /* mdbl-2008-s-1: doubling of the affine point */
        fq_t U = QDBL(qy), V = QSQR(U), W = QMUL(U, V), S = QMUL(*qx, V);
        fq_t X2 = QSQR(*qx), M = QADD(QDBL(X2), X2);
        acc->x = QSUB(QSQR(M), QDBL(S));
        acc->y = QSUB(QMUL(M, QSUB(S, acc->x)), QMUL(W, qy));
        acc->zz = V; acc->zzz = W;
        return;
    }
    fq_t PP = QSQR(P), PPP = QMUL(P, PP), Q = QMUL(acc->x, PP);
    fq_t X3 = QSUB(QSUB(QSQR(R), PPP), QDBL(Q));
    acc->y = QSUB(QMUL(R, QSUB(Q, X3)), QMUL(acc->y, PPP));
    acc->x = X3;
    acc->zz = QMUL(acc->zz, PP);
    acc->zzz = QMUL(acc->zzz, PPP);
}
/* acc += q (add-2008-s, 12M + 2S) */
static inline void xyzz_add(g1_xyzz_t *acc, const g1_xyzz_t *q) {
    if (xyzz_is_identity(q)) return;
    if (xyzz_is_identity(acc)) { *acc = *q; return; }
    fq_t U1 = QMUL(acc->x, q->zz), U2 = QMUL(q->x, acc->zz), S1 = QMUL(acc->y, q->zzz), S2 = QMUL(q->y, acc->zzz);
    fq_t P = QSUB(U2, U1), R = QSUB(S2, S1);
    if (u256_is_zero(&P)) {
        if (!u256_is_zero(&R)) { xyzz_set_identity(acc); return; }
        /* dbl-2008-s-1 */
        fq_t U = QDBL(acc->y), V = QSQR(U), W = QMUL(U, V), S = QMUL(acc->x, V);
        fq_t X2 = QSQR(acc->x), M = QADD(QDBL(X2), X2);
        fq_t X3 = QSUB(QSQR(M), QDBL(S));
        acc->y = QSUB(QMUL(M, QSUB(S, X3)), QMUL(W, acc->y));
        acc->x = X3;
        acc->zz = QMUL(V, acc->zz); acc->zzz = QMUL(W, acc->zzz);
        return;
    }
    fq_t PP = QSQR(P), PPP = QMUL(P, PP), Q = QMUL(U1, PP);
    fq_t X3 = QSUB(QSUB(QSQR(R), PPP), QDBL(Q));
    acc->y = QSUB(QMUL(R, QSUB(Q, X3)), QMUL(S1, PPP));
    acc->x = X3;
    acc->zz = QMUL(QMUL(acc->zz, q->zz), PP);
    acc->zzz = QMUL(QMUL(acc->zzz, q->zzz), PPP);
}
/* (X, Y, ZZ, ZZZ) -> Jacobian with Z = ZZ: X' = X ZZ, Y' = Y ZZZ (ZZ^3 = ZZZ^2) */
static inline g1_t xyzz_to_jacobian(const g1_xyzz_t *p) {
    if (xyzz_is_identity(p)) return g1_identity();
    g1_t r; r.x = QMUL(p->x, p->zz); r.y = QMUL(p->y, p->zzz); r.z = p->zz; return r;
}

/* Several MSMs over prefixes of the prepared bases as ONE pool of (msm, window, chunk) tasks scheduled dynamically over the host
 * threads -- what the reference gets from nested rayon parallelism (scheme.rs:141-145 commits the levels with par_iter; arkworks'
 * msm runs its windows in parallel under that).  Per MSM: ark's window rule capped at 16 bits, SIGNED digits (window value minus
 * 2^(c-1) of s + sum_w 2^(c-1) 2^(cw): carry-free per window, 2^(c-1) buckets), XYZZ buckets, windows above the scalars' top bit
 * skipped (64-bit witness columns cost 5 windows, not 16), chunks sized so that the pool has >= 4 tasks per thread while a chunk
 * still holds >= 8 points per bucket.  Same group elements as orc_g1_msm_pippenger (tests/test_oracle_g1.py). */
typedef struct { unsigned msm, window, chunk; } msm_task;
static unsigned g_baseline_max_c = 16; /* 2^15 XYZZ buckets = 4 MiB per thread; bench.py's calibration may lower it (orc_baseline_set_max_window) */
EXPORT void orc_baseline_set_max_window(unsigned c) { g_baseline_max_c = c < 4 ? 4 : (c > 16 ? 16 : c); }
EXPORT void orc_baseline_msm_many(const g1_t *bases, const fr_t *const *scalars, const size_t *lens, size_t count, g1_t *out) {
    if (count == 0) return;
    size_t max_n = 0;
    for (size_t i = 0; i < count; ++i) if (lens[i] > max_n) max_n = lens[i];
    if (!(g_aff && bases >= g_aff_src && bases + max_n <= g_aff_src + g_aff_n)) { /* bases not prepared: the serial restatement */
        for (size_t i = 0; i < count; ++i) orc_g1_msm_pippenger(bases, scalars[i], lens[i], &out[i]);
        return;
    }
    const g1_affine_t *aff = g_aff + (bases - g_aff_src);
    const int threads = omp_get_max_threads();
    unsigned *cs = (unsigned *)malloc(count * sizeof(unsigned)), *ws = (unsigned *)malloc(count * sizeof(unsigned)), *chs = (unsigned *)malloc(count * sizeof(unsigned));
    u256 **ks = (u256 **)malloc(count * sizeof(u256 *));
    size_t total_work = 0;
    for (size_t i = 0; i < count; ++i) {
        const size_t n = lens[i];
        ks[i] = (u256 *)malloc((n ? n : 1) * sizeof(u256));
        unsigned log2n = 0;
        while (((size_t)2 << log2n) <= n) log2n++;
        unsigned c = n < 32 ? 3 : (log2n * 69 / 100) + 2; /* ark's window rule, as in orc_g1_msm_pippenger */
        if (c > g_baseline_max_c) c = g_baseline_max_c;
        cs[i] = c;
        /* canonical scalars + the half-window offsets; top bit over the MSM */
        u256 half;
        memset(&half, 0, sizeof half);
        for (unsigned bit = c - 1; bit < 256; bit += c) half.l[bit / 64] |= (uint64_t)1 << (bit % 64);
        unsigned top = 0;
#pragma omp parallel for schedule(static) reduction(max : top)
        for (size_t j = 0; j < n; ++j) {
            u256 k;
            mont_to_canonical(&k, &scalars[i][j], &FR);
            unsigned b = 0;
            for (int l = 3; l >= 0; --l) if (k.l[l]) { b = 64 * (unsigned)l + 64 - (unsigned)__builtin_clzll(k.l[l]); break; }
            if (b > top) top = b;
            unsigned __int128 carry = 0;
            for (int l = 0; l < 4; ++l) { carry += (unsigned __int128)k.l[l] + half.l[l]; k.l[l] = (uint64_t)carry; carry >>= 64; }
            ks[i][j] = k;
        }
        unsigned full = (256 + c - 1) / c;            /* windows covering s + half < 2^256 */
        unsigned need = top / c + 2;                   /* the window holding the top bit and the one its carry may reach */
        ws[i] = need < full ? need : full;
        total_work += n * ws[i];
    }
    /* chunks: aim at 4 tasks per thread over the pool, keep >= 8 points per bucket in a chunk */
    size_t n_tasks = 0;
    const size_t target = total_work / ((size_t)threads * 4) + 1; /* point-window units per task */
    for (size_t i = 0; i < count; ++i) {
        const size_t buckets = (size_t)1 << (cs[i] - 1);
        size_t ch = lens[i] / (target > 8 * buckets ? target : 8 * buckets);
        if (ch < 1) ch = 1;
        chs[i] = (unsigned)ch;
        n_tasks += (size_t)ws[i] * ch;
    }
    msm_task *tasks = (msm_task *)malloc((n_tasks ? n_tasks : 1) * sizeof(msm_task));
    g1_xyzz_t *partial = (g1_xyzz_t *)malloc((n_tasks ? n_tasks : 1) * sizeof(g1_xyzz_t));
    size_t *first_task = (size_t *)malloc(count * sizeof(size_t));
    size_t t = 0;
    for (size_t i = 0; i < count; ++i) { /* long MSMs first: their tasks are the longest */
        first_task[i] = t;
        for (unsigned w = 0; w < ws[i]; ++w)
            for (unsigned ch = 0; ch < chs[i]; ++ch) { tasks[t].msm = (unsigned)i; tasks[t].window = w; tasks[t].chunk = ch; ++t; }
    }
#pragma omp parallel
    {
        g1_xyzz_t *buckets = (g1_xyzz_t *)malloc(((size_t)1 << 15) * sizeof(g1_xyzz_t));
#pragma omp for schedule(dynamic, 1)
        for (size_t k = 0; k < n_tasks; ++k) {
            const unsigned i = tasks[k].msm, w = tasks[k].window, c = cs[i];
            const size_t n = lens[i], nb = (size_t)1 << (c - 1);
            const size_t lo = (size_t)tasks[k].chunk * n / chs[i], hi = (size_t)(tasks[k].chunk + 1) * n / chs[i];
            for (size_t b = 0; b < nb; ++b) xyzz_set_identity(&buckets[b]);
            const unsigned start = w * c, limb = start / 64, shift = start % 64;
            const uint64_t mask = ((uint64_t)1 << c) - 1;
            const u256 *k256 = ks[i];
            const int64_t offset = start + c - 1 < 256 ? (int64_t)nb : 0;
            for (size_t j = lo; j < hi; ++j) {
                if (j + 8 < hi) { /* the bucket of the point 8 ahead: at c = 16 a thread's 4 MiB of buckets live in L3 / DRAM, not in its L2 */
                    uint64_t vp = k256[j + 8].l[limb] >> shift;
                    if (shift + c > 64 && limb < 3) vp |= k256[j + 8].l[limb + 1] << (64 - shift);
                    const int64_t dp = (int64_t)(vp & mask) - offset;
                    if (dp) {
                        const char *bp = (const char *)&buckets[(dp > 0 ? dp : -dp) - 1];
                        __builtin_prefetch(bp, 1, 1);
                        __builtin_prefetch(bp + 64, 1, 1);
                    }
                }
                uint64_t v = k256[j].l[limb] >> shift;
                if (shift + c > 64 && limb < 3) v |= k256[j].l[limb + 1] << (64 - shift);
                /* a top window whose half-window bit would lie beyond bit 255 carries no offset: its value (< 2^(256 - start) <= nb) is the digit */
                const int64_t d = (int64_t)(v & mask) - offset;
                if (d == 0) continue;
                const g1_affine_t *q = &aff[j];
                if (aff_is_inf(q)) continue;
                if (d > 0) xyzz_add_mixed(&buckets[d - 1], &q->x, &q->y, 0);
                else xyzz_add_mixed(&buckets[-d - 1], &q->x, &q->y, 1);
            }
            g1_xyzz_t running, sum;
            xyzz_set_identity(&running);
            xyzz_set_identity(&sum);
            for (size_t b = nb; b-- > 0;) {
                xyzz_add(&running, &buckets[b]);
                xyzz_add(&sum, &running);
            }
            partial[k] = sum;
        }
        free(buckets);
    }
    for (size_t i = 0; i < count; ++i) {
        g1_t total = g1_identity();
        for (int w = (int)ws[i] - 1; w >= 0; --w) {
            for (unsigned k = 0; k < cs[i]; ++k) total = g1_double(&total);
            g1_xyzz_t wsum;
            xyzz_set_identity(&wsum);
            for (unsigned ch = 0; ch < chs[i]; ++ch) xyzz_add(&wsum, &partial[first_task[i] + (size_t)w * chs[i] + ch]);
            g1_t wj = xyzz_to_jacobian(&wsum);
            total = g1_add(&total, &wj);
        }
        out[i] = total;
        free(ks[i]);
    }
    free(first_task); free(partial); free(tasks); free(ks); free(chs); free(ws); free(cs);
}
EXPORT void orc_baseline_msm(const g1_t *bases, const fr_t *scalars, size_t n, g1_t *out) {
    if (n == 0) { *out = g1_identity(); return; }
    orc_baseline_msm_many(bases, &scalars, &n, 1, out);
}
/* sums of bases[idx[p][j] * cycles + j] over the hot cycles of each column p: the one-hot columns' commitments on the K x T grid
 * (additions only -- the reference's tier-1 one-hot path, crates/jolt-dory/src/streaming.rs:160-205), all columns as one task pool */
EXPORT void orc_baseline_grid_onehot_sums(const g1_t *bases, const uint8_t *idx, size_t n_polys, size_t cycles, g1_t *out) {
    const g1_affine_t *aff = (g_aff && bases == g_aff_src) ? g_aff : NULL;
    if (!aff) { for (size_t p = 0; p < n_polys; ++p) out[p] = g1_identity(); return; }
    const int threads = omp_get_max_threads();
    size_t chunks = ((size_t)threads * 4 + n_polys - 1) / (n_polys ? n_polys : 1);
    if (chunks < 1) chunks = 1;
    while (chunks > 1 && cycles / chunks < 1024) chunks /= 2;
    g1_xyzz_t *partial = (g1_xyzz_t *)malloc(n_polys * chunks * sizeof(g1_xyzz_t));
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t k = 0; k < n_polys * chunks; ++k) {
        const size_t p = k / chunks, ch = k % chunks, lo = ch * cycles / chunks, hi = (ch + 1) * cycles / chunks;
        const uint8_t *col = idx + p * cycles;
        g1_xyzz_t acc;
        xyzz_set_identity(&acc);
        for (size_t j = lo; j < hi; ++j) {
            if (col[j] == 0xFF) continue;
            const g1_affine_t *q = &aff[(size_t)col[j] * cycles + j];
            if (!aff_is_inf(q)) xyzz_add_mixed(&acc, &q->x, &q->y, 0);
        }
        partial[k] = acc;
    }
    for (size_t p = 0; p < n_polys; ++p) {
        g1_xyzz_t acc;
        xyzz_set_identity(&acc);
        for (size_t ch = 0; ch < chunks; ++ch) xyzz_add(&acc, &partial[p * chunks + ch]);
        out[p] = xyzz_to_jacobian(&acc);
    }
    free(partial);
}
/* The MSM the HyperKZG restatement calls: the serial bucket method above by default; bench.py's cpu_baseline leg switches it to
 * the OpenMP form of oracle/baseline.c (same point, computed on all host cores). */
void (*orc_msm_impl)(const g1_t *, const fr_t *, size_t, g1_t *) = orc_g1_msm_pippenger;
static void msm_many_serial(const g1_t *bases, const fr_t *const *scalars, const size_t *lens, size_t count, g1_t *out) {
    for (size_t i = 0; i < count; ++i) orc_g1_msm_pippenger(bases, scalars[i], lens[i], &out[i]);
}
/* several independent MSMs over prefixes of one base array (the level commitments, the three witness commitments) */
void (*orc_msm_many_impl)(const g1_t *, const fr_t *const *, const size_t *, size_t, g1_t *) = msm_many_serial;
EXPORT void orc_baseline_use_parallel_msm(int on) {
    orc_msm_impl = on ? orc_baseline_msm : orc_g1_msm_pippenger;
    orc_msm_many_impl = on ? orc_baseline_msm_many : msm_many_serial;
}

/* HyperKZGScheme::setup_from_secret (crates/jolt-hyperkzg/src/scheme.rs:54-73): g1_powers[i] = beta^i * g1,
 * built by repeated scalar_mul exactly as the reference does (max_degree + 1 entries). */
EXPORT void orc_srs_setup_from_secret(const fr_t *beta, size_t count, g1_t *out) {
    g1_t cur;
    orc_g1_generator(&cur);
    for (size_t i = 0; i < count; ++i) {
        out[i] = cur;
        cur = g1_scalar_mul(&cur, beta);
    }
}

/* ---- Dory tier-1 (G1) streaming row commitments: crates/jolt-dory/src/streaming.rs ------------------------------------
 * feed_u64 / feed_i128 / feed_i128_rows_with (:115-205): row r of the batch commits to
 *     sum_j values[r*row_width + j] * bases[j]      (ark msm_u64 / msm_i128; a negative value contributes -(|v| * G_j)),
 * evaluated here the slow way: MSB-first double-and-add over the 128-bit magnitude of each value.
 * kind: 0 = u64, 1 = i64, 2 = i128 (two u64, low first, two's complement) -- the encodings of include/jolt_hip.h.
 * The reference pins this path only through commit -> open -> verify round trips (crates/jolt-dory/tests); there are no
 * golden points, so tests pin this function against orc_g1_msm_naive on the same integers lifted to Fr. */
EXPORT void orc_dory_commit_rows(const g1_t *bases, const void *values, int kind, size_t count, size_t row_width, g1_t *out) {
    size_t rows = row_width ? count / row_width : 0;
    for (size_t r = 0; r < rows; ++r) {
        g1_t acc = g1_identity();
        for (size_t j = 0; j < row_width; ++j) {
            size_t i = r * row_width + j;
            uint64_t lo, hi = 0;
            int negative = 0;
            if (kind == 2) {
                lo = ((const uint64_t *)values)[2 * i];
                hi = ((const uint64_t *)values)[2 * i + 1];
                negative = (int)(hi >> 63);
                if (negative) { lo = ~lo + 1; hi = ~hi + (lo == 0); }
            } else {
                lo = ((const uint64_t *)values)[i];
                negative = kind == 1 ? (int)(lo >> 63) : 0;
                if (negative) lo = ~lo + 1;
            }
            if (!lo && !hi) continue;
            g1_t t = g1_identity();
            for (int b = 127; b >= 0; --b) {
                t = g1_double(&t);
                uint64_t limb = b >= 64 ? hi : lo;
                if ((limb >> (b & 63)) & 1) t = g1_add(&t, &bases[j]);
            }
            if (negative) t = g1_neg(&t);
            acc = g1_add(&acc, &t);
        }
        out[r] = acc;
    }
}

/* one_hot_chunk_commitments (:366-419): indices_per_k[hot_row].push(column) for every hot column, then the per-row sums of
 * bases (batch_g1_additions_multi_affine); rows nobody hits stay Bn254G1::default() = identity.  idx: one byte per column,
 * 0xFF = None.  out: k points for this chunk. */
EXPORT void orc_dory_onehot_chunk(const g1_t *bases, const uint8_t *idx, size_t one_hot_k, size_t chunk_width, g1_t *out) {
    for (size_t k = 0; k < one_hot_k; ++k) out[k] = g1_identity();
    for (size_t col = 0; col < chunk_width; ++col) {
        uint8_t hot = idx[col];
        if (hot == 0xFF) continue;
        out[hot] = g1_add(&out[hot], &bases[col]);
    }
}
