/*
 * oracle/hyperkzg.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the HyperKZG prover path, paths relative to
 * /root/reference/crates/jolt-hyperkzg/src/:
 *   kzg_commit                 kzg.rs:15-27
 *   compute_witness_polynomial kzg.rs:34-46
 *   eval_univariate            kzg.rs:51-59
 *   kzg_open_batch             kzg.rs:69-126
 *   challenge_powers           kzg.rs:213-221
 *   fold_polynomials           scheme.rs:88-114
 *   open                       scheme.rs:122-158
 *
 * PARITY UNPINNED by vectors: the reference tests only round-trip commit->open->verify; there
 * are no golden commitments.  Pinned by algebraic identities in tests/test_oracle_g1.py:
 * commit(p) == p(beta)*G, witness-polynomial division identity (kzg.rs:229-264), fold == multilinear
 * partial evaluation, and the verifier's folding-consistency relation (scheme.rs:226-240) on (v, point, eval).
 */
#include "fr.h"
#include "mock_transcript.h"
#include <omp.h>
#include <stdlib.h>

#define EXPORT __attribute__((visibility("default")))

typedef struct { fq_t x, y, z; } g1_t;
void orc_g1_msm_pippenger(const g1_t *bases, const fr_t *scalars, size_t n, g1_t *out);
extern void (*orc_msm_impl)(const g1_t *, const fr_t *, size_t, g1_t *); /* g1.c: serial Pippenger unless the cpu_baseline leg swapped it */
extern void (*orc_msm_many_impl)(const g1_t *, const fr_t *const *, const size_t *, size_t, g1_t *); /* g1.c: the same, several independent MSMs */
void orc_g1_serialize_compressed(const g1_t *p, uint8_t out[32]);
void orc_bind_low_to_high(const fr_t *t, size_t len, const fr_t *r, fr_t *out);

/* kzg.rs:15-27 */
EXPORT int orc_kzg_commit(const fr_t *coeffs, size_t n, const g1_t *g1_powers, size_t srs_len, g1_t *out) {
    if (n > srs_len) return -1; /* HyperKZGError::SrsTooSmall */
    orc_msm_impl(g1_powers, coeffs, n, out);
    return 0;
}

/* kzg.rs:34-46: h = f / (x - u), h[i-1] = f[i] + h[i]*u ; output length d-1 */
EXPORT void orc_kzg_witness_polynomial(const fr_t *f, size_t d, const fr_t *u, fr_t *h) {
    if (d <= 1) return;
    fr_t acc = fr_zero();
    for (size_t i = d - 1; i >= 1; --i) {
        acc = FADD(f[i], FMUL(acc, *u));
        h[i - 1] = acc;
    }
}

/* kzg.rs:51-59 */
static fr_t fr_pow_u64(fr_t base, uint64_t e) {
    fr_t acc = fr_one();
    for (; e; e >>= 1) {
        if (e & 1) acc = FMUL(acc, base);
        base = FMUL(base, base);
    }
    return acc;
}
EXPORT void orc_kzg_eval_univariate(const fr_t *coeffs, size_t n, const fr_t *u, fr_t *out) {
    if (n < 65536) { /* the reference's loop verbatim */
        fr_t result = fr_zero(), power = fr_one();
        for (size_t i = 0; i < n; ++i) {
            result = FADD(result, FMUL(coeffs[i], power));
            power = FMUL(power, *u);
        }
        *out = result;
        return;
    }
    /* same sum, split into chunks that start from u^lo (exact arithmetic: the value does not depend on the split) */
    const size_t chunk = 16384, chunks = (n + chunk - 1) / chunk;
    fr_t result = fr_zero();
#pragma omp parallel
    {
        fr_t local = fr_zero();
#pragma omp for schedule(static) nowait
        for (size_t c = 0; c < chunks; ++c) {
            size_t lo = c * chunk, hi = lo + chunk < n ? lo + chunk : n;
            fr_t power = fr_pow_u64(*u, lo);
            for (size_t i = lo; i < hi; ++i) {
                local = FADD(local, FMUL(coeffs[i], power));
                power = FMUL(power, *u);
            }
        }
#pragma omp critical
        result = FADD(result, local);
    }
    *out = result;
}

/* scheme.rs:88-114: polys[0] = evals; fold i uses point[1..] back-to-front; polys laid out back to back
 * in `out` (total 2^ell + 2^(ell-1) + ... + 2 entries); returns the number of levels (= ell). */
EXPORT size_t orc_hyperkzg_fold_polynomials(const fr_t *evals, size_t ell, const fr_t *point, fr_t *out) {
    size_t n = (size_t)1 << ell;
    memcpy(out, evals, n * sizeof(fr_t));
    const fr_t *prev = out;
    fr_t *next = out + n;
    size_t len = n;
    for (size_t k = ell; k-- > 1;) { /* point[ell-1], ..., point[1] */
        orc_bind_low_to_high(prev, len, &point[k], next);
        prev = next;
        len /= 2;
        next += len;
    }
    return ell;
}

static void mt_append_g1(mock_transcript *t, const g1_t *p) {
    uint8_t b[32];
    orc_g1_serialize_compressed(p, b);
    mt_append_bytes(t, b, 32);
}

/* scheme.rs:122-158 open + kzg.rs:69-126 kzg_open_batch.
 * Outputs: com[ell-1], w[3], v[3][ell] (row-major), plus the three transcript challenges (r, q, d_0). */
EXPORT int orc_hyperkzg_open(const g1_t *g1_powers, size_t srs_len, const fr_t *evals, size_t ell, const fr_t *point,
                             uint64_t transcript_label, g1_t *com, g1_t *w, fr_t *v, fr_t *challenges_out) {
    if (ell == 0) return -2; /* HyperKZGError::EmptyPoint */
    size_t n = (size_t)1 << ell;
    if (n > srs_len) return -1;
    mock_transcript tr;
    mt_init(&tr, transcript_label);
    fr_t *polys = (fr_t *)malloc(2 * n * sizeof(fr_t));
    orc_hyperkzg_fold_polynomials(evals, ell, point, polys);
    /* offsets of each level */
    size_t off[64], len[64];
    off[0] = 0; len[0] = n;
    for (size_t i = 1; i < ell; ++i) { off[i] = off[i - 1] + len[i - 1]; len[i] = len[i - 1] / 2; }
    /* scheme.rs:141-145 */
    if (ell > 1) { /* ell - 1 independent kzg_commit calls (the reference maps them with par_iter) */
        const fr_t *ptrs[64];
        for (size_t i = 1; i < ell; ++i) { if (len[i] > srs_len) { free(polys); return -1; } ptrs[i - 1] = polys + off[i]; }
        orc_msm_many_impl(g1_powers, ptrs, len + 1, ell - 1, com);
    }
    /* scheme.rs:148-152 */
    for (size_t i = 1; i < ell; ++i) mt_append_g1(&tr, &com[i - 1]);
    fr_t r = mt_challenge(&tr);
    fr_t u[3] = {r, FNEG(r), FMUL(r, r)};
    /* kzg.rs:84-92 */
    for (int t = 0; t < 3; ++t)
        for (size_t j = 0; j < ell; ++j) orc_kzg_eval_univariate(polys + off[j], len[j], &u[t], &v[t * ell + j]);
    for (int t = 0; t < 3; ++t)
        for (size_t j = 0; j < ell; ++j) mt_append_fr(&tr, &v[t * ell + j]);
    /* kzg.rs:95-105 */
    fr_t q = mt_challenge(&tr);
    fr_t *b_poly = (fr_t *)malloc(n * sizeof(fr_t));
    for (size_t i = 0; i < n; ++i) b_poly[i] = fr_zero();
    fr_t qj = fr_one();
    for (size_t j = 0; j < ell; ++j) {
        const fr_t *pj = polys + off[j];
#pragma omp parallel for schedule(static) if (len[j] >= 65536)
        for (size_t i = 0; i < len[j]; ++i) b_poly[i] = FADD(b_poly[i], FMUL(qj, pj[i]));
        qj = FMUL(qj, q);
    }
    /* kzg.rs:108-116 */
    fr_t *h = (fr_t *)malloc(3 * n * sizeof(fr_t));
    {
        const fr_t *ptrs[3] = {h, h + n, h + 2 * n};
        const size_t lens[3] = {n - 1, n - 1, n - 1};
#pragma omp parallel for schedule(static) if (n >= 65536)
        for (int t = 0; t < 3; ++t) orc_kzg_witness_polynomial(b_poly, n, &u[t], h + (size_t)t * n);
        orc_msm_many_impl(g1_powers, ptrs, lens, 3, w);
    }
    /* kzg.rs:118-124 */
    for (int t = 0; t < 3; ++t) mt_append_g1(&tr, &w[t]);
    fr_t d0 = mt_challenge(&tr);
    if (challenges_out) { challenges_out[0] = r; challenges_out[1] = q; challenges_out[2] = d0; }
    free(h);
    free(b_poly);
    free(polys);
    return 0;
}

EXPORT void orc_mt_append_g1(mock_transcript *t, const g1_t *p) { mt_append_g1(t, p); }
