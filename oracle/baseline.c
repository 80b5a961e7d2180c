/*
 * oracle/baseline.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * OpenMP-parallel drivers over the oracle's restatements, used ONLY by bench.py's `cpu_baseline` leg
 * (kind = "port": the reference is Rust + rayon and cannot be built in this image).  They mirror the shape
 * of the reference's rayon loops: bind in parallel over output indices (crates/jolt-poly/src/dense.rs:223-303),
 * round sums as a parallel fold/reduce (crates/jolt-kernels/src/optimized/support.rs:521-561).
 */
#include "fr.h"
#include <omp.h>
#include <stdlib.h>

#define EXPORT __attribute__((visibility("default")))

EXPORT int orc_baseline_threads(void) { return omp_get_max_threads(); }
EXPORT void orc_baseline_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }

/* parallel LowToHigh bind into a scratch buffer (dense.rs:270-303 bind_low_to_high_reusing_scratch) */
EXPORT void orc_baseline_bind_low_to_high(const fr_t *t, size_t len, const fr_t *r, fr_t *out) {
    size_t half = len / 2;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < half; ++i) {
        fr_t lo = t[2 * i], hi = t[2 * i + 1];
        out[i] = FADD(lo, FMUL(*r, FSUB(hi, lo)));
    }
}

/* One fused sumcheck round of a sum-of-products member, the way the optimized tier walks it
 * (support.rs:521-561 par_sum_pair_groups): evaluations at t in {0,2,..,degree} over LowToHigh pairs.
 * evals_out[0] = s(0), evals_out[k] = s(k+1) for k >= 1 (degree entries in total). */
EXPORT void orc_baseline_round_evals(const fr_t *const *tables, uint32_t n_tables, size_t len, uint32_t n_terms,
                                     const uint32_t *term_offsets, const uint32_t *factors, const fr_t *coeffs,
                                     uint32_t degree, fr_t *evals_out) {
    size_t half = len / 2;
    int nt = omp_get_max_threads();
    fr_t *partial = (fr_t *)calloc((size_t)nt * 16, sizeof(fr_t));
#pragma omp parallel
    {
        int tid = omp_get_thread_num();
        fr_t acc[16];
        for (uint32_t k = 0; k < 16; ++k) acc[k] = fr_zero();
#pragma omp for schedule(static)
        for (size_t y = 0; y < half; ++y) {
            fr_t lo[64], step[64], cur[64];
            for (uint32_t i = 0; i < n_tables && i < 64; ++i) {
                lo[i] = tables[i][2 * y];
                step[i] = FSUB(tables[i][2 * y + 1], lo[i]);
                cur[i] = lo[i];
            }
            uint32_t slot = 0;
            for (uint32_t t = 0; t <= degree; ++t) {
                if (t == 1) { for (uint32_t i = 0; i < n_tables; ++i) cur[i] = FADD(cur[i], step[i]); continue; }
                fr_t result = fr_zero();
                for (uint32_t k = 0; k < n_terms; ++k) {
                    fr_t value = coeffs[k];
                    for (uint32_t f = term_offsets[k]; f < term_offsets[k + 1]; ++f) value = FMUL(value, cur[factors[f]]);
                    result = FADD(result, value);
                }
                acc[slot] = FADD(acc[slot], result);
                slot++;
                for (uint32_t i = 0; i < n_tables; ++i) cur[i] = FADD(cur[i], step[i]);
            }
        }
        for (uint32_t k = 0; k < degree; ++k) partial[(size_t)tid * 16 + k] = acc[k];
    }
    for (uint32_t k = 0; k < degree; ++k) {
        fr_t s = fr_zero();
        for (int t = 0; t < nt; ++t) s = FADD(s, partial[(size_t)t * 16 + k]);
        evals_out[k] = s;
    }
    free(partial);
}

/* Full sumcheck of one fused (product-of-linear-combinations) member on the CPU, the way the reference's optimized
 * rayon tier runs it: per round, bind every table low-to-high in parallel (dense.rs:270-303) and accumulate the
 * round sums at t in {0,2,..,degree} with per-thread accumulators merged at the end (support.rs:521-561).
 * Challenges are supplied (no transcript).  Descriptor arrays mirror jolt_member_lc_desc of include/jolt_hip.h.
 * out_evals receives the LAST round's `degree` sums (sanity output). */
EXPORT void orc_baseline_member_sumcheck(const fr_t *const *tables_in, uint32_t n_tables, size_t len, uint32_t n_groups,
                                         const uint32_t *grp_off, const uint32_t *fac_off, const fr_t *fac_const,
                                         const uint32_t *fac_has_const, const uint32_t *lc_tab, const fr_t *lc_coeff,
                                         const uint32_t *lc_one, uint32_t degree, const fr_t *challenges, size_t n_rounds,
                                         fr_t *out_evals) {
    fr_t **cur = (fr_t **)malloc(n_tables * sizeof(fr_t *));
    fr_t **alt = (fr_t **)malloc(n_tables * sizeof(fr_t *));
    for (uint32_t i = 0; i < n_tables; ++i) {
        cur[i] = (fr_t *)malloc(len * sizeof(fr_t));
        alt[i] = (fr_t *)malloc((len / 2 + 1) * sizeof(fr_t));
        memcpy(cur[i], tables_in[i], len * sizeof(fr_t));
    }
    int nt = omp_get_max_threads();
    fr_t *partial = (fr_t *)malloc((size_t)nt * 8 * sizeof(fr_t));
    for (size_t round = 0; round < n_rounds; ++round) {
        if (round > 0) {
            size_t half = len / 2;
            for (uint32_t i = 0; i < n_tables; ++i) {
                const fr_t *src = cur[i];
                fr_t *dst = alt[i];
                const fr_t r = challenges[round - 1];
#pragma omp parallel for schedule(static)
                for (size_t y = 0; y < half; ++y) {
                    fr_t lo = src[2 * y], hi = src[2 * y + 1];
                    dst[y] = FADD(lo, FMUL(r, FSUB(hi, lo)));
                }
                fr_t *tmp = cur[i]; cur[i] = alt[i]; alt[i] = tmp;
            }
            len = half;
        }
        size_t half = len / 2;
#pragma omp parallel
        {
            int tid = omp_get_thread_num();
            fr_t acc[8];
            for (uint32_t k = 0; k < 8; ++k) acc[k] = fr_zero();
#pragma omp for schedule(static)
            for (size_t y = 0; y < half; ++y) {
                for (uint32_t g = 0; g < n_groups; ++g) {
                    fr_t prod[8];
                    for (uint32_t f = grp_off[g]; f < grp_off[g + 1]; ++f) {
                        fr_t lo = fac_has_const[f] ? fac_const[f] : fr_zero(), hi = lo;
                        for (uint32_t k = fac_off[f]; k < fac_off[f + 1]; ++k) {
                            fr_t a = cur[lc_tab[k]][2 * y], b = cur[lc_tab[k]][2 * y + 1];
                            if (!lc_one[k]) { a = FMUL(a, lc_coeff[k]); b = FMUL(b, lc_coeff[k]); }
                            lo = FADD(lo, a);
                            hi = FADD(hi, b);
                        }
                        fr_t step = FSUB(hi, lo), v = lo;
                        int first = f == grp_off[g];
                        prod[0] = first ? v : FMUL(prod[0], v);
                        v = FADD(v, step); /* t = 1 skipped */
                        for (uint32_t t = 1; t < degree; ++t) {
                            v = FADD(v, step);
                            prod[t] = first ? v : FMUL(prod[t], v);
                        }
                    }
                    for (uint32_t t = 0; t < degree; ++t) acc[t] = FADD(acc[t], prod[t]);
                }
            }
            for (uint32_t t = 0; t < degree; ++t) partial[(size_t)tid * 8 + t] = acc[t];
        }
        for (uint32_t t = 0; t < degree; ++t) {
            fr_t s = fr_zero();
            for (int th = 0; th < nt; ++th) s = FADD(s, partial[(size_t)th * 8 + t]);
            out_evals[t] = s;
        }
    }
    for (uint32_t i = 0; i < n_tables; ++i) { free(cur[i]); free(alt[i]); }
    free(cur); free(alt); free(partial);
}

/* Joint polynomial of the stage-8 batch opening on the K x T commitment grid, cycle-major placement (index = k*T + j; dense
 * columns at k = 0) -- RlcSource::to_dense over the grid-embedded polynomials (crates/jolt-poly/src/multilinear.rs:159-170,
 * placement crates/jolt-kernels/src/optimized/opening.rs:340-372).  idx: n_polys columns of `cycles` hot indices (0xFF = cold). */
EXPORT void orc_baseline_grid_joint(const uint8_t *idx, uint32_t n_polys, size_t cycles, uint32_t k_grid, const fr_t *scalars,
                                    const fr_t *const *dense, uint32_t n_dense, const fr_t *dense_scalars, fr_t *out) {
#pragma omp parallel for schedule(static)
    for (size_t j = 0; j < cycles; ++j) {
        for (uint32_t k = 0; k < k_grid; ++k) out[(size_t)k * cycles + j] = fr_zero();
        for (uint32_t p = 0; p < n_polys; ++p) {
            uint8_t a = idx[(size_t)p * cycles + j];
            if (a == 0xFF) continue;
            fr_t *slot = &out[(size_t)a * cycles + j];
            *slot = FADD(*slot, scalars[p]);
        }
        for (uint32_t d = 0; d < n_dense; ++d) out[j] = FADD(out[j], FMUL(dense[d][j], dense_scalars[d]));
    }
}
