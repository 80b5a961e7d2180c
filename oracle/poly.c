/*
 * oracle/poly.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the jolt-poly pieces on the hot path:
 * dense multilinear bind, eq-table expansion, split-eq (Gruen) bookkeeping, LT / eq+1 tables and
 * univariate interpolation.  Each function cites the reference lines it follows
 * (paths relative to /root/reference/).
 *
 * Parity: the reference holds no golden vectors for these (properties only), so this file is
 * pinned by (a) the Fr layer's golden vectors and (b) the reference tests' own algebraic identities,
 * re-run in tests/test_oracle_poly.py (bind == evaluate, eq-table == product formula, split-eq == dense eq,
 * LT boolean truth table, eq+1 boolean truth table).
 */
#include "fr.h"
#include <stdlib.h>

#define EXPORT __attribute__((visibility("default")))

/* ---------------------------------------------------------------------------------------------
 * a3: Polynomial::bind  (crates/jolt-poly/src/dense.rs)
 * ------------------------------------------------------------------------------------------- */

/* dense.rs:188-220 bind_high_to_low: t[i] <- t[i] + r*(t[i+half]-t[i]); in place, result = first half */
EXPORT void orc_bind_high_to_low(fr_t *t, size_t len, const fr_t *r) {
    size_t half = len / 2;
    for (size_t i = 0; i < half; ++i) {
        fr_t lo = t[i], hi = t[i + half];
        t[i] = FADD(lo, FMUL(*r, FSUB(hi, lo)));
    }
}

/* dense.rs:223-263 bind_low_to_high: out[i] <- t[2i] + r*(t[2i+1]-t[2i]); `out` may alias `t` */
EXPORT void orc_bind_low_to_high(const fr_t *t, size_t len, const fr_t *r, fr_t *out) {
    size_t half = len / 2;
    /* out of place and large: outputs are independent (dense.rs:270-303 par_iter); in place it must stay sequential */
#pragma omp parallel for schedule(static) if (half >= 32768 && (out + half <= t || t + len <= out))
    for (size_t i = 0; i < half; ++i) {
        fr_t lo = t[2 * i], hi = t[2 * i + 1];
        out[i] = FADD(lo, FMUL(*r, FSUB(hi, lo)));
    }
}

/* dense.rs:129-142 bind_to_field for small-scalar tables (u64 promoted with From<u64>): HighToLow pairing */
EXPORT void orc_bind_to_field_u64(const uint64_t *t, size_t len, const fr_t *r, fr_t *out) {
    size_t half = len / 2;
    for (size_t i = 0; i < half; ++i) {
        fr_t lo = fr_from_u64(t[i]), hi = fr_from_u64(t[i + half]);
        out[i] = FADD(lo, FMUL(*r, FSUB(hi, lo)));
    }
}

/* ---------------------------------------------------------------------------------------------
 * a7: EqPolynomial  (crates/jolt-poly/src/eq.rs)
 * ------------------------------------------------------------------------------------------- */

/* eq.rs:299-315 evals_serial (== evals / evals_parallel values, eq.rs:221-231,374-458): big-endian,
 * r[0] pairs the index MSB; optional scale (NULL = one). out has 2^n entries. */
static void eq_evals_serial(const fr_t *r, size_t n, const fr_t *scale, fr_t *out) {
    size_t total = (size_t)1 << n;
    fr_t s = scale ? *scale : fr_one();
    for (size_t i = 0; i < total; ++i) out[i] = s;
    size_t size = 1;
    for (size_t j = 0; j < n; ++j) {
        size *= 2;
        for (size_t i = size; i-- > 0;) {
            if ((i & 1) == 0) continue; /* (0..size).rev().step_by(2): i = size-1, size-3, ... */
            fr_t scalar = out[i / 2];
            out[i] = FMUL(scalar, r[j]);
            out[i - 1] = FSUB(scalar, out[i]);
        }
    }
}
EXPORT void orc_eq_evals(const fr_t *r, size_t n, const fr_t *scale, fr_t *out) {
    if (n < 16) { eq_evals_serial(r, n, scale, out); return; }
    /* large tables (eq.rs:374-458 evals_parallel: same values): the table of the top 8 variables, then every aligned block of
     * 2^(n-8) entries expanded independently from its prefix value (the evals_for_aligned_block identity, eq.rs:238-263) */
    fr_t prefix[256];
    eq_evals_serial(r, 8, scale, prefix);
    const size_t block = (size_t)1 << (n - 8);
#pragma omp parallel for schedule(static)
    for (size_t b = 0; b < 256; ++b) eq_evals_serial(r + 8, n - 8, &prefix[b], out + b * block);
}

/* eq.rs:50-98 evaluations(): the interleaved doubling form; same table as orc_eq_evals(scale=1) */
EXPORT void orc_eq_evaluations(const fr_t *r, size_t n, fr_t *out) {
    out[0] = fr_one();
    size_t len = 1;
    for (size_t k = 0; k < n; ++k) {
        fr_t one_minus = FSUB(fr_one(), r[k]);
        for (size_t j = len; j-- > 0;) {
            fr_t base = out[j];
            out[2 * j] = FMUL(base, one_minus);
            out[2 * j + 1] = FMUL(base, r[k]);
        }
        len *= 2;
    }
}

/* eq.rs:238-263 evals_for_aligned_block: eq(r,k) for k in [start, start+block), block = 2^b aligned */
EXPORT void orc_eq_evals_aligned_block(const fr_t *r, size_t n, size_t start, size_t block, fr_t *out) {
    size_t block_vars = 0;
    while (((size_t)1 << block_vars) < block) block_vars++;
    size_t prefix_len = n - block_vars;
    size_t prefix_value = start >> block_vars;
    fr_t prefix_scale = fr_one();
    for (size_t pos = 0; pos < prefix_len; ++pos) {
        size_t shift = prefix_len - 1 - pos;
        int bit = (int)((prefix_value >> shift) & 1);
        fr_t f = bit ? r[pos] : FSUB(fr_one(), r[pos]);
        prefix_scale = FMUL(prefix_scale, f);
    }
    orc_eq_evals(r + prefix_len, block_vars, &prefix_scale, out);
}

/* eq.rs:100-118 evaluate: prod_i (r_i p_i + (1-r_i)(1-p_i)) */
EXPORT void orc_eq_mle(const fr_t *x, const fr_t *y, size_t n, fr_t *out) {
    fr_t acc = fr_one();
    for (size_t i = 0; i < n; ++i) {
        fr_t a = FMUL(x[i], y[i]);
        fr_t b = FMUL(FSUB(fr_one(), x[i]), FSUB(fr_one(), y[i]));
        acc = FMUL(acc, FADD(a, b));
    }
    *out = acc;
}

/* dense.rs:340-366 evaluate: sum_x f(x) eq(x, point) */
EXPORT void orc_poly_evaluate(const fr_t *evals, size_t n_vars, const fr_t *point, fr_t *out) {
    size_t total = (size_t)1 << n_vars;
    fr_t *eq = (fr_t *)malloc(total * sizeof(fr_t));
    orc_eq_evaluations(point, n_vars, eq);
    fr_t acc = fr_zero();
    for (size_t i = 0; i < total; ++i) acc = FADD(acc, FMUL(evals[i], eq[i]));
    free(eq);
    *out = acc;
}

/* ---------------------------------------------------------------------------------------------
 * a14: LT and eq+1 tables
 * ------------------------------------------------------------------------------------------- */

/* crates/jolt-poly/src/lt.rs:144-156 lt_evals */
EXPORT void orc_lt_evals(const fr_t *r, size_t n, fr_t *out) {
    size_t total = (size_t)1 << n;
    for (size_t i = 0; i < total; ++i) out[i] = fr_zero();
    for (size_t i = 0; i < n; ++i) {
        fr_t ri = r[n - 1 - i];
        size_t half = (size_t)1 << i;
#pragma omp parallel for schedule(static) if (half >= 32768) /* entry k touches only k and half + k */
        for (size_t k = 0; k < half; ++k) {
            fr_t x = out[k];
            fr_t y = FMUL(x, ri);
            out[half + k] = y;
            out[k] = FADD(x, FSUB(ri, y));
        }
    }
}

/* crates/jolt-poly/src/eq_plus_one.rs:71-130 evals: (eq table, eq+1 table), big-endian */
EXPORT void orc_eq_plus_one_evals(const fr_t *r, size_t ell, const fr_t *scale, fr_t *eq_out, fr_t *eqp1_out) {
    size_t size = (size_t)1 << ell;
#pragma omp parallel for schedule(static) if (size >= 65536)
    for (size_t i = 0; i < size; ++i) { eq_out[i] = fr_zero(); eqp1_out[i] = fr_zero(); }
    eq_out[0] = scale ? *scale : fr_one();
    for (size_t i = 0; i < ell; ++i) {
        size_t step = (size_t)1 << (ell - i);
        size_t half_step = step / 2;
        fr_t r_lower = fr_one();
        for (size_t j = i + 1; j < ell; ++j) r_lower = FMUL(r_lower, r[j]);
        r_lower = FMUL(r_lower, FSUB(fr_one(), r[i]));
#pragma omp parallel for schedule(static) if (size / step >= 32768)
        for (size_t idx = half_step; idx < size; idx += step) eqp1_out[idx] = FMUL(eq_out[idx - half_step], r_lower);
        size_t eq_step = (size_t)1 << (ell - i - 1);
#pragma omp parallel for schedule(static) if (size / (eq_step * 2) >= 32768)
        for (size_t k = 0; k < size; k += eq_step * 2) {
            fr_t val = FMUL(eq_out[k], r[i]);
            eq_out[k + eq_step] = val;
            eq_out[k] = FSUB(eq_out[k], val);
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * UnivariatePoly (crates/jolt-poly/src/univariate.rs)
 * ------------------------------------------------------------------------------------------- */

/* univariate.rs:198-202 from_evals: coefficients of the unique degree <= n-1 polynomial through
 * (0,e0),(1,e1),...,(n-1,e_{n-1}).  The reference solves the Vandermonde system; the solution is unique,
 * here obtained with Newton forward differences (exact in the field). */
EXPORT void orc_univariate_from_evals(const fr_t *evals, size_t n, fr_t *coeffs) {
    /* Newton divided differences on integer nodes: d_k = Delta^k e_0 / k! */
    fr_t d[16], c[16], basis[16];
    if (n > 16) n = 16;
    for (size_t i = 0; i < n; ++i) d[i] = evals[i];
    for (size_t k = 1; k < n; ++k) {
        fr_t kinv, kf = fr_from_u64(k);
        fr_inv(&kinv, &kf);
        for (size_t i = n - 1; i >= k; --i) {
            d[i] = FMUL(FSUB(d[i], d[i - 1]), kinv);
        }
    }
    /* p(x) = sum_k d_k * prod_{j<k} (x - j): expand */
    for (size_t i = 0; i < n; ++i) { c[i] = fr_zero(); basis[i] = fr_zero(); }
    basis[0] = fr_one();
    size_t blen = 1;
    for (size_t k = 0; k < n; ++k) {
        for (size_t i = 0; i < blen; ++i) c[i] = FADD(c[i], FMUL(d[k], basis[i]));
        if (k + 1 < n) { /* basis *= (x - k) */
            fr_t kf = fr_from_u64(k);
            fr_t nb[16];
            for (size_t i = 0; i <= blen; ++i) nb[i] = fr_zero();
            for (size_t i = 0; i < blen; ++i) {
                nb[i + 1] = FADD(nb[i + 1], basis[i]);
                nb[i] = FSUB(nb[i], FMUL(basis[i], kf));
            }
            blen += 1;
            for (size_t i = 0; i < blen; ++i) basis[i] = nb[i];
        }
    }
    for (size_t i = 0; i < n; ++i) coeffs[i] = c[i];
}

/* Horner evaluation of a coefficient vector (UnivariatePoly::evaluate) */
EXPORT void orc_univariate_evaluate(const fr_t *coeffs, size_t n, const fr_t *x, fr_t *out) {
    fr_t acc = fr_zero();
    for (size_t i = n; i-- > 0;) acc = FADD(FMUL(acc, *x), coeffs[i]);
    *out = acc;
}

/* ---------------------------------------------------------------------------------------------
 * a6: split-eq  (crates/jolt-poly/src/split_eq.rs), LowToHigh order only (every
 * T-scale bind in the kernels is LowToHigh)
 * ------------------------------------------------------------------------------------------- */

/* split_eq.rs:214-236: the two cached-table points of GruenSplitEqPolynomial::new(LowToHigh):
 * head = point[..n-1], split = n/2, out_point = head[..min(split,len)], in_point = rest.
 * After `bound` binds the current tables are the prefix tables selected by the pop rule
 * (split_eq.rs:339-350).  Returns in_bits/out_bits of the *current* e_in/e_out tables. */
EXPORT void orc_split_eq_current_dims(size_t n, size_t bound, size_t *out_bits, size_t *in_bits) {
    if (n == 0) { *out_bits = 0; *in_bits = 0; return; }
    size_t split = n / 2;
    size_t head_len = n - 1;
    size_t out_len = split < head_len ? split : head_len;
    size_t in_len = head_len - out_len;
    size_t e_out = out_len, e_in = in_len; /* cached vectors hold tables for prefixes 0..=len; current = last */
    size_t current_index = n;
    for (size_t b = 0; b < bound; ++b) {
        current_index -= 1;
        if (n / 2 < current_index && e_in > 0) e_in -= 1;
        else if (0 < current_index && e_out > 0) e_out -= 1;
    }
    *out_bits = e_out;
    *in_bits = e_in;
}

/* split_eq.rs:334-337: scalar update of bind(): s *= 1 - p - c + 2pc */
EXPORT void orc_split_eq_bind_scalar(const fr_t *scalar, const fr_t *point_i, const fr_t *challenge, fr_t *out) {
    fr_t prod = FMUL(*point_i, *challenge);
    fr_t f = FADD(FADD(FSUB(FSUB(fr_one(), *point_i), *challenge), prod), prod);
    *out = FMUL(*scalar, f);
}

/* split_eq.rs:383-417 gruen_poly_deg_3: cubic round message from (q(0), q(inf), claim). coeffs[4]. */
EXPORT int orc_gruen_poly_deg_3(const fr_t *current_scalar, const fr_t *point_i, const fr_t *q_constant,
                                const fr_t *q_quadratic, const fr_t *s0_plus_s1, fr_t *coeffs) {
    fr_t eq1 = FMUL(*current_scalar, *point_i);
    fr_t eq0 = FSUB(*current_scalar, eq1);
    fr_t eqm = FSUB(eq1, eq0);
    fr_t eq2 = FADD(eq1, eqm);
    fr_t eq3 = FADD(eq2, eqm);
    fr_t quad0 = *q_constant;
    fr_t cubic0 = FMUL(eq0, quad0);
    fr_t cubic1 = FSUB(*s0_plus_s1, cubic0);
    fr_t eq1_inv;
    if (!fr_inv(&eq1_inv, &eq1)) return -1;
    fr_t quad1 = FMUL(cubic1, eq1_inv);
    fr_t e2 = FADD(*q_quadratic, *q_quadratic);
    fr_t quad2 = FADD(FSUB(FADD(quad1, quad1), quad0), e2);
    fr_t quad3 = FADD(FADD(FSUB(FADD(quad2, quad1), quad0), e2), e2);
    fr_t evals[4] = {cubic0, cubic1, FMUL(eq2, quad2), FMUL(eq3, quad3)};
    orc_univariate_from_evals(evals, 4, coeffs);
    return 0;
}
