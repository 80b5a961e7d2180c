/*
 * oracle/sumcheck.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the sumcheck members and the batched round loop (SURVEY.md section 8 rows a4, a5,
 * a6, a9).  Paths relative to /root/reference/:
 *   generic member (semantic definition)  crates/jolt-kernels/src/reference/naive.rs:211-318
 *   summand semantics (Expr)              crates/jolt-claims/src/claims.rs:17-46,101-130
 *   skipped-evals assembly                crates/jolt-kernels/src/optimized/support.rs:450-459
 *   split-eq product member               crates/jolt-kernels/src/optimized/support.rs:391-411,
 *                                         crates/jolt-poly/src/split_eq.rs:334-417,449-512,
 *                                         crates/jolt-kernels/src/optimized/ram_hamming_booleanity.rs:111-135
 *   batched round loop                    crates/jolt-sumcheck/src/prover.rs:193-362, batch.rs:23-72,
 *                                         recorder.rs:118-130
 *
 * Parity: no golden vectors exist in the reference for round polynomials; pinned by the
 * reference's own identities re-run in tests/test_oracle_sumcheck.py: s(0)+s(1)==claim every round,
 * final claim == summand at the bound point (evaluate via eq tables), skipped-evals == direct evals,
 * split-eq member == dense eq member, DenseMember::with_sum fixtures
 * (crates/jolt-sumcheck/src/tests.rs:1129-1135).
 */
#include "fr.h"
#include "mock_transcript.h"
#include <omp.h>
#include <stdlib.h>

/* The sweeps over the hypercube are split over OpenMP threads (per-thread partial sums merged by addition: field addition is
 * exact, so the merge order cannot change a value -- the reference's rayon fold/reduce makes the same argument,
 * crates/jolt-kernels/src/reference/naive.rs:262-296).  That is what lets the GPU parity tests run this restatement at the
 * benchmark's trace lengths (T = 2^20 .. 2^22) instead of toy sizes. */
#define ORC_PAR_MIN 4096

#define EXPORT __attribute__((visibility("default")))

void orc_bind_low_to_high(const fr_t *t, size_t len, const fr_t *r, fr_t *out);
void orc_bind_high_to_low(fr_t *t, size_t len, const fr_t *r);
void orc_eq_evals(const fr_t *r, size_t n, const fr_t *scale, fr_t *out);
void orc_univariate_from_evals(const fr_t *evals, size_t n, fr_t *coeffs);
void orc_univariate_evaluate(const fr_t *coeffs, size_t n, const fr_t *x, fr_t *out);
void orc_split_eq_current_dims(size_t n, size_t bound, size_t *out_bits, size_t *in_bits);
void orc_split_eq_bind_scalar(const fr_t *scalar, const fr_t *point_i, const fr_t *challenge, fr_t *out);
int orc_gruen_poly_deg_3(const fr_t *current_scalar, const fr_t *point_i, const fr_t *q_constant,
                         const fr_t *q_quadratic, const fr_t *s0_plus_s1, fr_t *coeffs);

enum { ORC_ORDER_LOW_TO_HIGH = 0, ORC_ORDER_HIGH_TO_LOW = 1 };
enum { ORC_KIND_EXPR = 0, ORC_KIND_GRUEN_PRODUCT = 1 };
enum { ORC_OK = 0, ORC_ERR_ROUND_CHECK = -1, ORC_ERR_ARG = -2, ORC_ERR_DEGREE = -3, ORC_ERR_NOT_INVERTIBLE = -4 };

typedef struct {
    int kind;
    size_t rounds, rounds_bound;
    size_t len; /* current table length */
    uint32_t n_tables;
    fr_t **tables; /* owned copies */
    /* EXPR */
    uint32_t n_terms, degree;
    int order, skip_one;
    uint32_t *term_offsets, *factors;
    fr_t *coeffs;
    /* GRUEN_PRODUCT: eq(w, j) * tables[0](j) * tables[1](j), LowToHigh */
    fr_t *w; /* point, n = rounds */
    fr_t current_scalar;
} orc_member;

/* naive.rs:136-205 NaiveSumcheckProver::new -- tables are copied (the kernel owns its tables).
 * Challenge leaves are pre-folded into `coeffs` by the caller (exact: coefficient * prod challenge). */
EXPORT orc_member *orc_member_create_expr(const fr_t *const *tables, uint32_t n_tables, size_t len, uint32_t n_terms,
                                          const uint32_t *term_offsets, const uint32_t *factors, const fr_t *coeffs,
                                          uint32_t degree, int order, int skip_one) {
    orc_member *m = (orc_member *)calloc(1, sizeof(orc_member));
    m->kind = ORC_KIND_EXPR;
    m->len = len;
    m->rounds = 0;
    while (((size_t)1 << m->rounds) < len) m->rounds++;
    m->n_tables = n_tables;
    m->tables = (fr_t **)calloc(n_tables, sizeof(fr_t *));
    for (uint32_t i = 0; i < n_tables; ++i) {
        m->tables[i] = (fr_t *)malloc(len * sizeof(fr_t));
        memcpy(m->tables[i], tables[i], len * sizeof(fr_t));
    }
    m->n_terms = n_terms;
    m->degree = degree;
    m->order = order;
    m->skip_one = skip_one;
    m->term_offsets = (uint32_t *)malloc((n_terms + 1) * sizeof(uint32_t));
    memcpy(m->term_offsets, term_offsets, (n_terms + 1) * sizeof(uint32_t));
    uint32_t nf = term_offsets[n_terms];
    m->factors = (uint32_t *)malloc((nf ? nf : 1) * sizeof(uint32_t));
    memcpy(m->factors, factors, nf * sizeof(uint32_t));
    m->coeffs = (fr_t *)malloc(n_terms * sizeof(fr_t));
    memcpy(m->coeffs, coeffs, n_terms * sizeof(fr_t));
    return m;
}

/* eq(w,.) * a * b member served from split tables; scale = optional initial scalar
 * (GruenSplitEqPolynomial::new_with_scaling, split_eq.rs:187-260) */
EXPORT orc_member *orc_member_create_gruen_product(const fr_t *a, const fr_t *b, size_t len, const fr_t *w,
                                                   const fr_t *scale) {
    orc_member *m = (orc_member *)calloc(1, sizeof(orc_member));
    m->kind = ORC_KIND_GRUEN_PRODUCT;
    m->len = len;
    m->rounds = 0;
    while (((size_t)1 << m->rounds) < len) m->rounds++;
    m->n_tables = 2;
    m->tables = (fr_t **)calloc(2, sizeof(fr_t *));
    m->tables[0] = (fr_t *)malloc(len * sizeof(fr_t));
    m->tables[1] = (fr_t *)malloc(len * sizeof(fr_t));
    memcpy(m->tables[0], a, len * sizeof(fr_t));
    memcpy(m->tables[1], b, len * sizeof(fr_t));
    m->degree = 3;
    m->w = (fr_t *)malloc((m->rounds ? m->rounds : 1) * sizeof(fr_t));
    memcpy(m->w, w, m->rounds * sizeof(fr_t));
    m->current_scalar = scale ? *scale : fr_one();
    return m;
}

EXPORT void orc_member_destroy(orc_member *m) {
    if (!m) return;
    for (uint32_t i = 0; i < m->n_tables; ++i) free(m->tables[i]);
    free(m->tables);
    free(m->term_offsets);
    free(m->factors);
    free(m->coeffs);
    free(m->w);
    free(m);
}

EXPORT size_t orc_member_num_rounds(const orc_member *m) { return m->rounds; }
EXPORT uint32_t orc_member_degree(const orc_member *m) { return m->degree; }

/* naive.rs:211-219 bind_tables */
static void member_bind(orc_member *m, const fr_t *challenge) {
    if (m->kind == ORC_KIND_GRUEN_PRODUCT) {
        /* split_eq.rs:334-337 (LowToHigh): point = w[current_index-1], current_index = rounds - bound */
        size_t current_index = m->rounds - m->rounds_bound;
        orc_split_eq_bind_scalar(&m->current_scalar, &m->w[current_index - 1], challenge, &m->current_scalar);
    }
    for (uint32_t i = 0; i < m->n_tables; ++i) {
        if (m->kind == ORC_KIND_GRUEN_PRODUCT || m->order == ORC_ORDER_LOW_TO_HIGH) {
            if (m->len >= ORC_PAR_MIN) { /* out of place so that the outputs can be computed in parallel (dense.rs:270-303) */
                size_t half = m->len / 2;
                fr_t *src = m->tables[i], *dst = (fr_t *)malloc(half * sizeof(fr_t));
                const fr_t r = *challenge;
#pragma omp parallel for schedule(static)
                for (size_t y = 0; y < half; ++y) {
                    fr_t lo = src[2 * y], hi = src[2 * y + 1];
                    dst[y] = FADD(lo, FMUL(r, FSUB(hi, lo)));
                }
                free(src);
                m->tables[i] = dst;
            } else {
                orc_bind_low_to_high(m->tables[i], m->len, challenge, m->tables[i]);
            }
        } else if (m->len >= ORC_PAR_MIN) {
            size_t half = m->len / 2;
            fr_t *t = m->tables[i];
            const fr_t r = *challenge;
#pragma omp parallel for schedule(static)
            for (size_t y = 0; y < half; ++y) t[y] = FADD(t[y], FMUL(r, FSUB(t[y + half], t[y])));
        } else {
            orc_bind_high_to_low(m->tables[i], m->len, challenge);
        }
    }
    m->len /= 2;
    m->rounds_bound += 1;
}

/* dense.rs:309-320 sumcheck_eval_pair */
static inline void eval_pair(const orc_member *m, uint32_t table, size_t y, fr_t *lo, fr_t *hi) {
    const fr_t *t = m->tables[table];
    if (m->order == ORC_ORDER_LOW_TO_HIGH) { *lo = t[2 * y]; *hi = t[2 * y + 1]; }
    else { *lo = t[y]; *hi = t[y + m->len / 2]; }
}

/* naive.rs:241-310: msg(t) = sum_y Expr(leaves at lo + t*(hi-lo)), t = 0..=degree (t=1 skipped when
 * skip_one and recovered from the claim, support.rs:450-459). Returns degree+1 coefficients. */
static int expr_round(orc_member *m, const fr_t *previous_claim, fr_t *coeffs_out) {
    size_t half = m->len / 2;
    uint32_t d = m->degree;
    fr_t evals[16];
    if (d + 1 > 16) return ORC_ERR_DEGREE;
    for (uint32_t t = 0; t <= d; ++t) {
        if (m->skip_one && t == 1) continue;
        fr_t point = fr_from_u64(t);
        fr_t sum = fr_zero();
#pragma omp parallel if (half >= ORC_PAR_MIN)
        {
            fr_t local = fr_zero();
#pragma omp for schedule(static) nowait
            for (size_t y = 0; y < half; ++y) {
                fr_t result = fr_zero();
                for (uint32_t k = 0; k < m->n_terms; ++k) {
                    fr_t value = m->coeffs[k];
                    for (uint32_t f = m->term_offsets[k]; f < m->term_offsets[k + 1]; ++f) {
                        fr_t lo, hi;
                        eval_pair(m, m->factors[f], y, &lo, &hi);
                        /* dense.rs:328-337: lo + point * (hi - lo) */
                        fr_t v = FADD(lo, FMUL(point, FSUB(hi, lo)));
                        value = FMUL(value, v);
                    }
                    result = FADD(result, value);
                }
                local = FADD(local, result);
            }
#pragma omp critical
            sum = FADD(sum, local);
        }
        evals[t] = sum;
    }
    if (m->skip_one) {
        evals[1] = FSUB(*previous_claim, evals[0]);
    } else {
        fr_t round_sum = FADD(evals[0], evals[1]);
        if (!fr_eq(&round_sum, previous_claim)) return ORC_ERR_ROUND_CHECK;
    }
    orc_univariate_from_evals(evals, d + 1, coeffs_out);
    return ORC_OK;
}

/* support.rs:391-411 product_endpoints over split_eq.rs:449-512 par_fold_out_in, then
 * split_eq.rs:383-417 gruen_poly_deg_3 */
static int gruen_round(orc_member *m, const fr_t *previous_claim, fr_t *coeffs_out) {
    size_t n = m->rounds;
    size_t out_bits, in_bits;
    orc_split_eq_current_dims(n, m->rounds_bound, &out_bits, &in_bits);
    /* split_eq.rs:214-216: head = w[..n-1]; out_point = head[..split], in_point = head[split..] */
    size_t split = n / 2;
    size_t head_len = n ? n - 1 : 0;
    size_t out_len = split < head_len ? split : head_len;
    size_t e_out_n = (size_t)1 << out_bits, e_in_n = (size_t)1 << in_bits;
    fr_t *e_out = (fr_t *)malloc(e_out_n * sizeof(fr_t));
    fr_t *e_in = (fr_t *)malloc(e_in_n * sizeof(fr_t));
    orc_eq_evals(m->w, out_bits, NULL, e_out);           /* evals_cached(out_point)[out_bits] */
    orc_eq_evals(m->w + out_len, in_bits, NULL, e_in);   /* evals_cached(in_point)[in_bits]  */
    const fr_t *a = m->tables[0], *b = m->tables[1];
    fr_t zero = fr_zero(), infinity = fr_zero();
#pragma omp parallel if (m->len >= ORC_PAR_MIN)
    {
        fr_t lz = fr_zero(), li = fr_zero();
#pragma omp for schedule(static) nowait
        for (size_t x_out = 0; x_out < e_out_n; ++x_out) {
            fr_t acc0 = fr_zero(), acc1 = fr_zero();
            for (size_t x_in = 0; x_in < e_in_n; ++x_in) {
                size_t row = (x_out << in_bits) | x_in;
                fr_t a_low = a[2 * row], a_high = a[2 * row + 1];
                fr_t b_low = b[2 * row], b_high = b[2 * row + 1];
                acc0 = FADD(acc0, FMUL(e_in[x_in], FMUL(a_low, b_low)));
                acc1 = FADD(acc1, FMUL(e_in[x_in], FMUL(FSUB(a_high, a_low), FSUB(b_high, b_low))));
            }
            lz = FADD(lz, FMUL(e_out[x_out], acc0));
            li = FADD(li, FMUL(e_out[x_out], acc1));
        }
#pragma omp critical
        {
            zero = FADD(zero, lz);
            infinity = FADD(infinity, li);
        }
    }
    free(e_out);
    free(e_in);
    size_t current_index = n - m->rounds_bound;
    if (orc_gruen_poly_deg_3(&m->current_scalar, &m->w[current_index - 1], &zero, &infinity, previous_claim, coeffs_out))
        return ORC_ERR_NOT_INVERTIBLE;
    return ORC_OK;
}

/* ProveRounds::prove_round (jolt-sumcheck/src/prover.rs:57-66): bind (NULL on first active round), then
 * the round polynomial's degree+1 coefficients. */
EXPORT int orc_member_prove_round(orc_member *m, const fr_t *bind, const fr_t *previous_claim, fr_t *coeffs_out) {
    if (bind) member_bind(m, bind);
    if (m->len < 2) return ORC_ERR_ARG;
    return m->kind == ORC_KIND_EXPR ? expr_round(m, previous_claim, coeffs_out) : gruen_round(m, previous_claim, coeffs_out);
}

/* Raw round sums without the round check or interpolation -- what ONE SHARD of a hypercube-sharded member contributes
 * (its sums do not add up to the global claim by themselves).  EXPR: out[t] = s(t), t = 0..degree (naive.rs:262-296);
 * GRUEN: out = {q(0), q(inf)} (support.rs:391-411) times `shard_scale` (= eq(w_hi, rank); NULL = one). */
EXPORT int orc_member_round_sums(orc_member *m, const fr_t *bind, const fr_t *shard_scale, fr_t *out) {
    if (bind) member_bind(m, bind);
    if (m->len < 2) return ORC_ERR_ARG;
    size_t half = m->len / 2;
    if (m->kind == ORC_KIND_EXPR) {
        for (uint32_t t = 0; t <= m->degree; ++t) {
            fr_t point = fr_from_u64(t), sum = fr_zero();
            for (size_t y = 0; y < half; ++y)
                for (uint32_t k = 0; k < m->n_terms; ++k) {
                    fr_t value = m->coeffs[k];
                    for (uint32_t f = m->term_offsets[k]; f < m->term_offsets[k + 1]; ++f) {
                        fr_t lo, hi;
                        eval_pair(m, m->factors[f], y, &lo, &hi);
                        value = FMUL(value, FADD(lo, FMUL(point, FSUB(hi, lo))));
                    }
                    sum = FADD(sum, value);
                }
            out[t] = sum;
        }
        return ORC_OK;
    }
    size_t n = m->rounds, out_bits, in_bits;
    orc_split_eq_current_dims(n, m->rounds_bound, &out_bits, &in_bits);
    size_t split = n / 2, head_len = n ? n - 1 : 0;
    size_t out_len = split < head_len ? split : head_len;
    size_t e_out_n = (size_t)1 << out_bits, e_in_n = (size_t)1 << in_bits;
    fr_t *e_out = (fr_t *)malloc(e_out_n * sizeof(fr_t)), *e_in = (fr_t *)malloc(e_in_n * sizeof(fr_t));
    orc_eq_evals(m->w, out_bits, shard_scale, e_out);
    orc_eq_evals(m->w + out_len, in_bits, NULL, e_in);
    fr_t zero = fr_zero(), infinity = fr_zero();
    for (size_t x_out = 0; x_out < e_out_n; ++x_out)
        for (size_t x_in = 0; x_in < e_in_n; ++x_in) {
            size_t row = (x_out << in_bits) | x_in;
            fr_t e = FMUL(e_out[x_out], e_in[x_in]);
            fr_t al = m->tables[0][2 * row], ah = m->tables[0][2 * row + 1], bl = m->tables[1][2 * row], bh = m->tables[1][2 * row + 1];
            zero = FADD(zero, FMUL(e, FMUL(al, bl)));
            infinity = FADD(infinity, FMUL(e, FMUL(FSUB(ah, al), FSUB(bh, bl))));
        }
    free(e_out);
    free(e_in);
    out[0] = zero;
    out[1] = infinity;
    return ORC_OK;
}

/* ProveRounds::finish_rounds (prover.rs:68-71) */
EXPORT int orc_member_finish_rounds(orc_member *m, const fr_t *bind) {
    member_bind(m, bind);
    return ORC_OK;
}

/* SumcheckKernel::output_claims (naive.rs:331-347): each table's fully bound value t[0];
 * for the gruen member out[n_tables] additionally receives the bound eq scalar. */
EXPORT int orc_member_final_values(const orc_member *m, fr_t *out) {
    if (m->rounds_bound != m->rounds) return ORC_ERR_ARG;
    for (uint32_t i = 0; i < m->n_tables; ++i) out[i] = m->tables[i][0];
    if (m->kind == ORC_KIND_GRUEN_PRODUCT) out[m->n_tables] = m->current_scalar;
    return ORC_OK;
}

/* Current (partially bound) evaluations of table t: `len >> rounds_bound` values (test hook for the sharded tail hand-over). */
EXPORT size_t orc_member_current_len(const orc_member *m) { return m->len; }
EXPORT int orc_member_copy_table(const orc_member *m, uint32_t t, fr_t *out, size_t cap) {
    if (t >= m->n_tables || cap < m->len) return ORC_ERR_ARG;
    memcpy(out, m->tables[t], m->len * sizeof(fr_t));
    return ORC_OK;
}

/* The member's input claim = sum over the hypercube of its summand (what the stage would consume).
 * For EXPR: sum_x Expr(x); for GRUEN: sum_x scale*eq(w,x) a(x) b(x). */
EXPORT void orc_member_input_claim(const orc_member *m, fr_t *out) {
    fr_t sum = fr_zero();
    fr_t *eq = NULL;
    if (m->kind != ORC_KIND_EXPR) {
        eq = (fr_t *)malloc(m->len * sizeof(fr_t));
        orc_eq_evals(m->w, m->rounds, &m->current_scalar, eq);
    }
#pragma omp parallel if (m->len >= ORC_PAR_MIN)
    {
        fr_t local = fr_zero();
#pragma omp for schedule(static) nowait
        for (size_t x = 0; x < m->len; ++x) {
            if (m->kind == ORC_KIND_EXPR) {
                for (uint32_t k = 0; k < m->n_terms; ++k) {
                    fr_t value = m->coeffs[k];
                    for (uint32_t f = m->term_offsets[k]; f < m->term_offsets[k + 1]; ++f) value = FMUL(value, m->tables[m->factors[f]][x]);
                    local = FADD(local, value);
                }
            } else {
                local = FADD(local, FMUL(eq[x], FMUL(m->tables[0][x], m->tables[1][x])));
            }
        }
#pragma omp critical
        sum = FADD(sum, local);
    }
    free(eq);
    *out = sum;
}

/* ---------------------------------------------------------------------------------------------
 * a9: prove_batch  (crates/jolt-sumcheck/src/prover.rs:193-362)
 * ------------------------------------------------------------------------------------------- */

/* out_polys: max_num_vars rows of (max_degree+1) coefficients (batched round polynomials, trailing zeros kept);
 * out_challenges: max_num_vars; out_member_claims: n_members; out_final_claim: 1.
 * `challenge_mode` 0 = 125-bit challenges (mt_challenge), 1 = full-width (mt_challenge_scalar). */
EXPORT int orc_prove_batch(orc_member **members, size_t n_members, const fr_t *input_claims, const fr_t *coefficients,
                           const size_t *offsets, size_t max_num_vars, size_t max_degree, uint64_t transcript_label,
                           int challenge_mode, fr_t *out_polys, fr_t *out_challenges, fr_t *out_member_claims,
                           fr_t *out_final_claim) {
    mock_transcript tr;
    mt_init(&tr, transcript_label);
    fr_t two = fr_from_u64(2), two_inv;
    fr_inv(&two_inv, &two);
    size_t stride = max_degree + 1;
    fr_t *member_claims = (fr_t *)malloc((n_members ? n_members : 1) * sizeof(fr_t));
    fr_t *pending = (fr_t *)malloc((n_members ? n_members : 1) * sizeof(fr_t));
    int *has_pending = (int *)calloc(n_members ? n_members : 1, sizeof(int));
    int rc = ORC_OK;
    /* batch.rs:57-71 claimed_sum; prover.rs:244-249 padded member claims */
    fr_t running_claim = fr_zero();
    for (size_t i = 0; i < n_members; ++i) {
        size_t rounds = members[i]->rounds;
        if (offsets[i] + rounds > max_num_vars) { rc = ORC_ERR_ARG; goto done; }
        member_claims[i] = fr_mul_pow_2(input_claims[i], (unsigned)(max_num_vars - rounds));
        running_claim = FADD(running_claim, FMUL(coefficients[i], member_claims[i]));
    }
    for (size_t round = 0; round < max_num_vars; ++round) {
        fr_t *batched = &out_polys[round * stride];
        for (size_t k = 0; k < stride; ++k) batched[k] = fr_zero();
        fr_t polys[64][16];
        int active[64];
        if (n_members > 64 || stride > 16) { rc = ORC_ERR_ARG; goto done; }
        for (size_t i = 0; i < n_members; ++i) {
            size_t rounds = members[i]->rounds;
            active[i] = round >= offsets[i] && round < offsets[i] + rounds;
            if (!active[i]) {
                /* prover.rs:273-282: inactive member contributes the constant claim/2 */
                member_claims[i] = FMUL(member_claims[i], two_inv);
                batched[0] = FADD(batched[0], FMUL(coefficients[i], member_claims[i]));
                continue;
            }
            uint32_t d = members[i]->degree;
            if (d > max_degree) { rc = ORC_ERR_DEGREE; goto done; }
            for (size_t k = 0; k < stride; ++k) polys[i][k] = fr_zero();
            rc = orc_member_prove_round(members[i], has_pending[i] ? &pending[i] : NULL, &member_claims[i], polys[i]);
            has_pending[i] = 0;
            if (rc) goto done;
            for (size_t k = 0; k <= d; ++k) batched[k] = FADD(batched[k], FMUL(coefficients[i], polys[i][k]));
        }
        /* prover.rs:316-324 round check: s(0)+s(1) == running claim */
        fr_t s0 = batched[0], s1 = fr_zero();
        for (size_t k = 0; k < stride; ++k) s1 = FADD(s1, batched[k]);
        fr_t round_sum = FADD(s0, s1);
        if (!fr_eq(&round_sum, &running_claim)) { rc = ORC_ERR_ROUND_CHECK; goto done; }
        /* recorder.rs:118-130: append the compressed poly (linear term omitted, univariate.rs:180-189),
         * trailing zeros trimmed down to degree 1 (prover.rs:168-177), then squeeze the challenge */
        size_t ncoef = stride;
        while (ncoef > 2 && fr_is_zero(&batched[ncoef - 1])) ncoef--;
        mt_append_round_poly(&tr, "sumcheck_poly", batched, ncoef); /* CompressedLabeledRoundPoly::sumcheck (round_proof.rs:115-143) */
        fr_t challenge = challenge_mode ? mt_challenge_scalar(&tr) : mt_challenge(&tr);
        out_challenges[round] = challenge;
        orc_univariate_evaluate(batched, stride, &challenge, &running_claim);
        for (size_t i = 0; i < n_members; ++i) {
            if (!active[i]) continue;
            orc_univariate_evaluate(polys[i], members[i]->degree + 1, &challenge, &member_claims[i]);
            pending[i] = challenge;
            has_pending[i] = 1;
        }
    }
    /* prover.rs:343-355 finish_rounds for every ever-active member */
    for (size_t i = 0; i < n_members; ++i)
        if (has_pending[i]) orc_member_finish_rounds(members[i], &pending[i]);
    for (size_t i = 0; i < n_members; ++i) out_member_claims[i] = member_claims[i];
    *out_final_claim = running_claim;
done:
    free(member_claims);
    free(pending);
    free(has_pending);
    return rc;
}

/* Mock-transcript access for tests that drive the product side with the same challenge stream */
EXPORT size_t orc_mt_sizeof(void) { return sizeof(mock_transcript); }
EXPORT void orc_mt_init(mock_transcript *t, uint64_t label) { mt_init(t, label); }
EXPORT int orc_mt_init_bytes(mock_transcript *t, uint64_t kind, const uint8_t *label, size_t n) { return mt_init_bytes(t, kind, label, n); }
EXPORT void orc_mt_append_label(mock_transcript *t, const char *label) { mt_append_label(t, label); }
EXPORT void orc_mt_append_label_with_count(mock_transcript *t, const char *label, uint64_t count) { mt_append_label_with_count(t, label, count); }
EXPORT void orc_mt_append_u64_word(mock_transcript *t, uint64_t v) { mt_append_u64_word(t, v); }
EXPORT void orc_mt_append_round_poly(mock_transcript *t, const char *label, const fr_t *coeffs, size_t n) { mt_append_round_poly(t, label, coeffs, n); }
EXPORT void orc_mt_state(const mock_transcript *t, uint8_t out[32]) { /* Transcript::state (digest.rs:191-193; legacy.rs:302-304: 32 bytes squeezed from a clone) */
    if (t->kind == MT_KIND_BLAKE2B_LEGACY) memcpy(out, t->state, 32);
    else if (t->kind == MT_KIND_KECCAK_SPONGE) { orc_keccak_duplex c = t->sponge; orc_duplex_squeeze(&c, out, 32); }
    else if (t->kind == MT_KIND_BLAKE2B_SPONGE) { mock_transcript c = *t; mt_bridge_squeeze(&c, out, 32); }
    else memcpy(out, t->s, 32);
}
EXPORT void orc_blake2b_digest(const uint8_t *in, size_t n, size_t outlen, uint8_t *out) {
    orc_blake2b h;
    orc_blake2b_init(&h, outlen);
    orc_blake2b_update(&h, in, n);
    orc_blake2b_final(&h, out);
}
EXPORT void orc_keccak_f1600_permute(uint8_t st[200]) { orc_keccak_f1600(st); }
EXPORT void orc_mt_append_bytes(mock_transcript *t, const uint8_t *b, size_t n) { mt_append_bytes(t, b, n); }
EXPORT void orc_mt_append_fr(mock_transcript *t, const fr_t *a) { mt_append_fr(t, a); }
EXPORT void orc_mt_challenge(mock_transcript *t, fr_t *out) { *out = mt_challenge(t); }
EXPORT void orc_mt_challenge_scalar(mock_transcript *t, fr_t *out) { *out = mt_challenge_scalar(t); }

/* ---------------------------------------------------------------------------------------------
 * a5: optimized dense forms used as a CPU-baseline workload and as extra parity surfaces
 * ------------------------------------------------------------------------------------------- */

/* support.rs:474-516 triple_product_round_evals: s(t) at t in {0,2,3} of sum_y a b c over LowToHigh pairs */
EXPORT void orc_triple_product_round_evals(const fr_t *a, const fr_t *b, const fr_t *c, size_t len, fr_t out[3]) {
    size_t half = len / 2;
    fr_t acc[3] = {fr_zero(), fr_zero(), fr_zero()};
    for (size_t y = 0; y < half; ++y) {
        fr_t a0 = a[2 * y], a1 = a[2 * y + 1], b0 = b[2 * y], b1 = b[2 * y + 1], c0 = c[2 * y], c1 = c[2 * y + 1];
        fr_t am = FSUB(a1, a0), bm = FSUB(b1, b0), cm = FSUB(c1, c0);
        fr_t a2 = FADD(a1, am), b2 = FADD(b1, bm), c2 = FADD(c1, cm);
        acc[0] = FADD(acc[0], FMUL(FMUL(a0, b0), c0));
        acc[1] = FADD(acc[1], FMUL(FMUL(a2, b2), c2));
        acc[2] = FADD(acc[2], FMUL(FMUL(FADD(a2, am), FADD(b2, bm)), FADD(c2, cm)));
    }
    out[0] = acc[0]; out[1] = acc[1]; out[2] = acc[2];
}
