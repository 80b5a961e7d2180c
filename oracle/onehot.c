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
