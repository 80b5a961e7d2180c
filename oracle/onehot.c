/* oracle/onehot.c -- TEST INFRASTRUCTURE (CPU restatement, never shipped): one-hot selector columns kept as hot indices.
 *
 * Follows crates/jolt-kernels/src/optimized/lazy_ra.rs:
 *   gather            :184-209   value(i,j) = sum_{off<width} table[off*stride + index(i, j*width+off)]
 *   double_branches   :211-231   next = [(1-c)*table ; c*table]
 *   materialize       :233-268   dense[j] = gather(.., branches, j), j < cycles/branches
 * and the pushforward tables of crates/jolt-kernels/src/optimized/booleanity.rs:24-31 (G_i[k] = sum_j w_j ra_i(k,j)).
 * The reference has no golden vectors for these (its own parity statement, lazy_ra.rs:26-32, is "same value as the iterated
 * dense bind"), so tests/test_oracle_onehot.py pins this file against the oracle's dense bind chain.
 * Index encoding: one byte per (polynomial, cycle), 0xFF = cold cycle (ChunkIndexSource::index -> None, lazy_ra.rs:47-50). */
#include <stdlib.h>
#include <string.h>

#include "fr.h"

#define EXPORT __attribute__((visibility("default")))
#define ORC_COLD 0xFF

static fr_t gather(const fr_t *table, size_t width, size_t k_entries, const uint8_t *col, size_t j) {
    fr_t sum = fr_zero();
    for (size_t off = 0; off < width; ++off) {
        uint8_t k = col[j * width + off];
        if (k != ORC_COLD) sum = FADD(sum, table[off * k_entries + k]);
    }
    return sum;
}

/* out[j] = value(j) at branch width `width`, for j < cycles / width */
EXPORT void orc_onehot_values(const fr_t *table, size_t width, size_t k_entries, const uint8_t *col, size_t cycles, fr_t *out) {
    for (size_t j = 0; j < cycles / width; ++j) out[j] = gather(table, width, k_entries, col, j);
}

/* in: width*k_entries, out: 2*width*k_entries */
EXPORT void orc_onehot_double_branches(const fr_t *in, size_t entries, const fr_t *challenge, fr_t *out) {
    fr_t one_minus = FSUB(fr_one(), *challenge);
    for (size_t e = 0; e < entries; ++e) {
        out[e] = FMUL(one_minus, in[e]);
        out[entries + e] = FMUL(*challenge, in[e]);
    }
}

/* G[k] = sum_j w[j] * [col[j] == k] */
EXPORT void orc_onehot_pushforward(const uint8_t *col, size_t cycles, size_t k_entries, const fr_t *w, fr_t *out) {
    for (size_t k = 0; k < k_entries; ++k) out[k] = fr_zero();
    for (size_t j = 0; j < cycles; ++j)
        if (col[j] != ORC_COLD) out[col[j]] = FADD(out[col[j]], w[j]);
}

/* ---- booleanity address phase (stage 6a): crates/jolt-kernels/src/optimized/booleanity.rs:283-427 ------------------------------------
 * OptimizedBooleanityAddressKernel over the pushforward masses G_i (linear = squared = G_i at the start; gamma weights g^(2i)):
 *   prove_round :349-398  s(c), c = 0..3, over low-to-high pairs:  sum_y eq_t(y) * sum_i w_i * ( (1-c)^2 sq_i[2y] + c^2 sq_i[2y+1] - lin_i,t(y) )
 *   bind        :320-341  linear and eq_address as multilinears, squared with the weights (1-r)^2 / r^2 (one-hot columns: cross terms vanish)
 *   output      :413-427  intermediate = eq[0] * sum_i w_i (sq_i[0] - lin_i[0])
 * Tables are n_polys rows of `len` entries each (row-major), bound in place (the first len / 2 entries of a row remain). */
EXPORT void orc_booleanity_address_round(const fr_t *linear, const fr_t *squared, size_t n_polys, size_t stride, size_t len, const fr_t *weights, const fr_t *eq_address,
                                         fr_t evals[4]) {
    const size_t half = len / 2;
    for (uint64_t c = 0; c < 4; ++c) {
        const fr_t point = fr_from_u64(c), point_sqr = FMUL(point, point), om = FSUB(fr_one(), point), one_minus_sqr = FMUL(om, om);
        fr_t sum = fr_zero();
        for (size_t y = 0; y < half; ++y) {
            fr_t inner = fr_zero();
            for (size_t i = 0; i < n_polys; ++i) {
                const fr_t *sq = squared + i * stride, *lin = linear + i * stride;
                const fr_t squared_ext = FADD(FMUL(one_minus_sqr, sq[2 * y]), FMUL(point_sqr, sq[2 * y + 1]));
                const fr_t linear_ext = FADD(lin[2 * y], FMUL(point, FSUB(lin[2 * y + 1], lin[2 * y])));
                inner = FADD(inner, FMUL(weights[i], FSUB(squared_ext, linear_ext)));
            }
            const fr_t eq_ext = FADD(eq_address[2 * y], FMUL(point, FSUB(eq_address[2 * y + 1], eq_address[2 * y])));
            sum = FADD(sum, FMUL(eq_ext, inner));
        }
        evals[c] = sum;
    }
}
EXPORT void orc_booleanity_address_bind(fr_t *linear, fr_t *squared, size_t n_polys, size_t stride, size_t len, fr_t *eq_address, const fr_t *r) {
    const size_t half = len / 2;
    const fr_t om = FSUB(fr_one(), *r), one_minus_sqr = FMUL(om, om), challenge_sqr = FMUL(*r, *r);
    for (size_t i = 0; i < n_polys; ++i) {
        fr_t *lin = linear + i * stride, *sq = squared + i * stride;
        for (size_t k = 0; k < half; ++k) {
            lin[k] = FADD(lin[2 * k], FMUL(*r, FSUB(lin[2 * k + 1], lin[2 * k])));
            sq[k] = FADD(FMUL(one_minus_sqr, sq[2 * k]), FMUL(challenge_sqr, sq[2 * k + 1]));
        }
    }
    for (size_t k = 0; k < half; ++k) eq_address[k] = FADD(eq_address[2 * k], FMUL(*r, FSUB(eq_address[2 * k + 1], eq_address[2 * k])));
}

/* Hamming-weight claim reduction (stage 7), crates/jolt-kernels/src/optimized/hamming_weight_claim_reduction.rs: the combined weights
 * W_i(k) = g^(3i) + g^(3i+1) eq_bool(k) + g^(3i+2) eq_virt_i(k) (:187-207, gamma_powers = 1, g, g^2, ..), the round sums of sum_i G_i W_i at t = 0 and t = 2
 * (group_evals :255-266) with the plain sum as a third value, and the bind of every table (:243-252).  eq tables are passed in (orc_eq_evals). */
EXPORT void orc_hamming_weights(const fr_t *gamma, const fr_t *eq_bool, const fr_t *eq_virt /* n_polys * k */, size_t n_polys, size_t k_entries, fr_t *out) {
    fr_t power = fr_one();
    for (size_t i = 0; i < n_polys; ++i) {
        fr_t g0 = power, g1 = FMUL(g0, *gamma), g2 = FMUL(g1, *gamma);
        power = FMUL(g2, *gamma);
        for (size_t k = 0; k < k_entries; ++k)
            out[i * k_entries + k] = FADD(g0, FADD(FMUL(g1, eq_bool[k]), FMUL(g2, eq_virt[i * k_entries + k])));
    }
}
EXPORT void orc_pair_tables_round(const fr_t *g, const fr_t *w, size_t n_polys, size_t stride, size_t len, fr_t *out /* 3 */) {
    fr_t s0 = fr_zero(), s2 = fr_zero(), total = fr_zero();
    for (size_t i = 0; i < n_polys; ++i) {
        const fr_t *gi = g + i * stride, *wi = w + i * stride;
        for (size_t y = 0; y < len / 2; ++y) {
            fr_t g_lo = gi[2 * y], g_hi = gi[2 * y + 1], w_lo = wi[2 * y], w_hi = wi[2 * y + 1];
            s0 = FADD(s0, FMUL(g_lo, w_lo));
            s2 = FADD(s2, FMUL(FSUB(FADD(g_hi, g_hi), g_lo), FSUB(FADD(w_hi, w_hi), w_lo)));
        }
        for (size_t k = 0; k < len; ++k) total = FADD(total, FMUL(gi[k], wi[k]));
    }
    out[0] = s0;
    out[1] = s2;
    out[2] = total;
}
EXPORT void orc_pair_tables_bind(fr_t *g, fr_t *w, size_t n_polys, size_t stride, size_t len, const fr_t *r) {
    for (size_t i = 0; i < n_polys; ++i) {
        fr_t *tabs[2] = {g + i * stride, w + i * stride};
        for (int t = 0; t < 2; ++t)
            for (size_t k = 0; k < len / 2; ++k) tabs[t][k] = FADD(tabs[t][2 * k], FMUL(*r, FSUB(tabs[t][2 * k + 1], tabs[t][2 * k])));
    }
}
