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
#pragma omp parallel for schedule(dynamic, 1) if (cycles >= 65536) /* the nodes are independent sums */
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
#pragma omp parallel for schedule(static) if (cycles >= 65536)
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

/* ---- Spartan product virtualization (stage 2): crates/jolt-kernels/src/optimized/spartan_product.rs ------------------------------
 * Three factor lanes on a 3-node centered uni-skip window {-1, 0, 1}, extended to {-2 .. 2}.  Per cycle (extended_products :113-141):
 *   left(node)  = c0 * left_instruction_input + c1 * lookup_output + c2 * jump_flag
 *   right(node) = c0 * right_instruction_input + c1 * branch_flag + c2 * (1 - next_is_noop)
 * with c = extension_coefficients()[node] (:86-105: the integer Lagrange basis of the window at the node; 0/1 selectors inside the
 * window).  The reference computes these as exact integers (S128 x S192 -> S256); here they are taken mod p first -- the same
 * field values (ring homomorphism), which is the relation its own parity test asserts. */
#define PRODUCT_DOMAIN 3
#define PRODUCT_EXTENDED 5
static void product_extension_coefficients(int64_t out[PRODUCT_EXTENDED][PRODUCT_DOMAIN]) {
    const int64_t domain_start = -((PRODUCT_DOMAIN - 1) / 2), extended_start = -((PRODUCT_EXTENDED - 1) / 2);
    for (int position = 0; position < PRODUCT_EXTENDED; ++position) {
        const int64_t node = extended_start + position;
        for (int i = 0; i < PRODUCT_DOMAIN; ++i) {
            int64_t numerator = 1, denominator = 1;
            for (int j = 0; j < PRODUCT_DOMAIN; ++j) {
                if (j == i) continue;
                numerator *= node - (domain_start + j);
                denominator *= (int64_t)i - j;
            }
            out[position][i] = numerator / denominator; /* exact: consecutive-integer domain */
        }
    }
}
EXPORT void orc_spartan_product_extension_coefficients(int64_t *out /* 5 x 3 */) {
    int64_t c[PRODUCT_EXTENDED][PRODUCT_DOMAIN];
    product_extension_coefficients(c);
    for (int p = 0; p < PRODUCT_EXTENDED; ++p)
        for (int i = 0; i < PRODUCT_DOMAIN; ++i) out[p * PRODUCT_DOMAIN + i] = c[p][i];
}
static void product_lanes(const uint64_t *left_input, const uint64_t *lookup_output, const uint8_t *jump, const uint64_t *right_input /* i128 two's complement: lo, hi */,
                          const uint8_t *branch, const uint8_t *next_is_noop, size_t j, fr_t left[3], fr_t right[3]) {
    left[0] = fr_from_u64(left_input[j]);
    left[1] = fr_from_u64(lookup_output[j]);
    left[2] = fr_from_u64(jump[j]);
    uint64_t lo = right_input[2 * j], hi = right_input[2 * j + 1];
    const int negative = (int)(hi >> 63);
    if (negative) {
        lo = ~lo + 1;
        hi = ~hi + (lo == 0 ? 1 : 0);
    }
    right[0] = fr_from_i128(lo, hi, negative);
    right[1] = fr_from_u64(branch[j]);
    right[2] = fr_from_u64(1 - (uint64_t)next_is_noop[j]);
}
/* extended_t1_values (:203-230): t1(node) = sum_j eq(tau_low, j) * left_node(j) * right_node(j) for all 5 extended nodes */
EXPORT void orc_spartan_product_t1(const uint64_t *left_input, const uint64_t *lookup_output, const uint8_t *jump, const uint64_t *right_input, const uint8_t *branch,
                                   const uint8_t *next_is_noop, size_t cycles, const fr_t *eq, fr_t *out /* 5 */) {
    int64_t c[PRODUCT_EXTENDED][PRODUCT_DOMAIN];
    product_extension_coefficients(c);
    for (int p = 0; p < PRODUCT_EXTENDED; ++p) out[p] = fr_zero();
    for (size_t j = 0; j < cycles; ++j) {
        fr_t l[3], r[3];
        product_lanes(left_input, lookup_output, jump, right_input, branch, next_is_noop, j, l, r);
        for (int p = 0; p < PRODUCT_EXTENDED; ++p) {
            fr_t left = fr_zero(), right = fr_zero();
            for (int i = 0; i < PRODUCT_DOMAIN; ++i) {
                const fr_t ci = fr_from_i64(c[p][i]);
                left = FADD(left, FMUL(ci, l[i]));
                right = FADD(right, FMUL(ci, r[i]));
            }
            out[p] = FADD(out[p], FMUL(eq[j], FMUL(left, right)));
        }
    }
}
/* ProductRemainderKernel::prepare's cell() (:358-378): the remainder's left / right tables under the uni-skip challenge's Lagrange weights */
EXPORT void orc_spartan_product_tables(const uint64_t *left_input, const uint64_t *lookup_output, const uint8_t *jump, const uint64_t *right_input, const uint8_t *branch,
                                       const uint8_t *next_is_noop, size_t cycles, const fr_t *weights /* 3 */, fr_t *left, fr_t *right) {
    for (size_t j = 0; j < cycles; ++j) {
        fr_t l[3], r[3];
        product_lanes(left_input, lookup_output, jump, right_input, branch, next_is_noop, j, l, r);
        left[j] = fr_zero();
        right[j] = fr_zero();
        for (int i = 0; i < 3; ++i) {
            left[j] = FADD(left[j], FMUL(weights[i], l[i]));
            right[j] = FADD(right[j], FMUL(weights[i], r[i]));
        }
    }
}
