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
