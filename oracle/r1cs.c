/*
 * oracle/r1cs.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the T-scale sums of the Spartan outer (stage 1) kernels, paths relative to
 * /root/reference/crates/jolt-kernels/src/reference/spartan_outer.rs:
 *   row_value_tables            :318-349   az_rows[r][t] = sum_{(v,a) in A_r} a * z_t[v], z_t[0] = 1, z_t[1+k] = input k
 *   uniskip_first_round_poly    :172-221   t1(node) = sum_t sum_s eq[(t << 1) | s] * Az(node,s,t) * Bz(node,s,t),
 *                                          Az(node,s,t) = sum_r w(node,s)[r] * az_rows[r][t]
 *   into_remainder              :236-300   Az / Bz of the remainder member as linear forms in the inputs with per-stream column weights
 *                                          (ConstraintMatrices::weighted_columns / public_column_contributions)
 * The Lagrange machinery (centered_lagrange_evals, interpolate_to_coeffs, spartan_outer_row_weights) and the constraint list
 * itself (crates/jolt-r1cs/src/constraints/jolt.rs) stay with the caller: they are O(rows) host work; what is restated here is what
 * touches every cycle.  The row-weight form (the reference's loop) and the column-weight form (what the device kernels take) are both
 * here; tests/test_oracle_r1cs.py checks they agree and that the materialised Az / Bz reproduce the reference's dense remainder member.
 *
 * PARITY UNPINNED by vectors (the reference holds none for stage 1 below whole-proof byte equality, tests/dory_byte_diff.rs).
 */
#include "fr.h"
#include <stdlib.h>

#define EXPORT __attribute__((visibility("default")))

/* rows[r][t] = sum_{e in [offsets[r], offsets[r+1])} coeffs[e] * z_t[cols[e]]   (spartan_outer.rs:318-349) */
EXPORT void orc_r1cs_row_values(const fr_t *const *inputs, size_t cycles, uint32_t n_rows, const uint32_t *offsets, const uint32_t *cols,
                                const fr_t *coeffs, fr_t *rows_out /* n_rows * cycles */) {
    for (uint32_t r = 0; r < n_rows; ++r)
        for (size_t t = 0; t < cycles; ++t) {
            fr_t acc = fr_zero();
            for (uint32_t e = offsets[r]; e < offsets[r + 1]; ++e)
                acc = FADD(acc, cols[e] == 0 ? coeffs[e] : FMUL(coeffs[e], inputs[cols[e] - 1][t]));
            rows_out[(size_t)r * cycles + t] = acc;
        }
}

/* the reference's loop verbatim (spartan_outer.rs:186-216): row_weights[(node * 2 + s) * n_rows + r] */
EXPORT void orc_r1cs_uniskip_sums_rows(const fr_t *az_rows, const fr_t *bz_rows, uint32_t n_rows, size_t cycles, const fr_t *eq /* 2 * cycles */,
                                       const fr_t *row_weights, uint32_t n_nodes, fr_t *out) {
    for (uint32_t n = 0; n < n_nodes; ++n) {
        fr_t sum = fr_zero();
        for (size_t t = 0; t < cycles; ++t)
            for (uint32_t s = 0; s < 2; ++s) {
                const fr_t *w = row_weights + ((size_t)n * 2 + s) * n_rows;
                fr_t az = fr_zero(), bz = fr_zero();
                for (uint32_t r = 0; r < n_rows; ++r) {
                    az = FADD(az, FMUL(w[r], az_rows[(size_t)r * cycles + t]));
                    bz = FADD(bz, FMUL(w[r], bz_rows[(size_t)r * cycles + t]));
                }
                sum = FADD(sum, FMUL(eq[(t << 1) | s], FMUL(az, bz)));
            }
        out[n] = sum;
    }
}

/* the same sums with the row weights folded into per-column weights (weighted_columns, :246-256):
 * a_weights[((node * 2 + s) * (1 + n_inputs)) + c], column 0 = the constant */
EXPORT void orc_r1cs_uniskip_sums(const fr_t *const *inputs, uint32_t n_inputs, size_t cycles, const fr_t *eq, const fr_t *a_weights, const fr_t *b_weights,
                                  uint32_t n_nodes, fr_t *out) {
    const size_t stride = 1 + (size_t)n_inputs;
    for (uint32_t n = 0; n < n_nodes; ++n) {
        fr_t sum = fr_zero();
        for (size_t t = 0; t < cycles; ++t)
            for (uint32_t s = 0; s < 2; ++s) {
                const fr_t *wa = a_weights + ((size_t)n * 2 + s) * stride, *wb = b_weights + ((size_t)n * 2 + s) * stride;
                fr_t az = wa[0], bz = wb[0];
                for (uint32_t v = 0; v < n_inputs; ++v) {
                    az = FADD(az, FMUL(wa[1 + v], inputs[v][t]));
                    bz = FADD(bz, FMUL(wb[1 + v], inputs[v][t]));
                }
                sum = FADD(sum, FMUL(eq[(t << 1) | s], FMUL(az, bz)));
            }
        out[n] = sum;
    }
}

/* az_out[(t << 1) | s] = a_weights[s][0] + sum_v a_weights[s][1 + v] * z_v(t); likewise bz (the remainder member's two linear forms
 * over the joint (cycle || stream) domain, into_remainder :258-300 multiplied out) */
EXPORT void orc_r1cs_materialize(const fr_t *const *inputs, uint32_t n_inputs, size_t cycles, const fr_t *a_weights /* 2 * (1 + n_inputs) */,
                                 const fr_t *b_weights, fr_t *az_out, fr_t *bz_out) {
    const size_t stride = 1 + (size_t)n_inputs;
    for (size_t t = 0; t < cycles; ++t)
        for (uint32_t s = 0; s < 2; ++s) {
            fr_t az = a_weights[s * stride], bz = b_weights[s * stride];
            for (uint32_t v = 0; v < n_inputs; ++v) {
                az = FADD(az, FMUL(a_weights[s * stride + 1 + v], inputs[v][t]));
                bz = FADD(bz, FMUL(b_weights[s * stride + 1 + v], inputs[v][t]));
            }
            az_out[(t << 1) | s] = az;
            bz_out[(t << 1) | s] = bz;
        }
}
