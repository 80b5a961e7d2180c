/* oracle/address_ops.c -- TEST INFRASTRUCTURE (CPU restatement, never shipped): the T-scale halves of the joint-domain relations whose rounds
 * run over K-sized address tables.
 *
 * Follows
 *   crates/jolt-kernels/src/optimized/bytecode_read_raf.rs:152-237  stage_pushforwards: the five per-stage cycle-eq pushforwards onto the bytecode
 *       address domain, all stages in one trace walk over the split-eq two-table decomposition eq(r, j) = E_hi[j_hi] * E_lo[j_lo],
 *       j = (j_hi << lo_bits) | j_lo, lo_bits = log_t / 2: per j_hi block the inner sums are additions of E_lo entries per touched PC (membership by an
 *       epoch marker, :170-196), then partial[k] += E_hi[j_hi] * inner[k] over the touched PCs (:197-205)
 *   crates/jolt-kernels/src/optimized/ram_trace.rs:150-162          fold_cycles: out[k] = sum_{j : addresses[j] = k} eq_cycle[j] (NO_ACCESS skipped)
 *   the witness oracle's ram_val_final column (optimized/ram_output_check.rs:91: dense_view(witness, ram_val_final())): the word an address holds
 *       after its last access, the initial word where it is never accessed
 * The reference has no golden vectors for these; its own statement (bytecode_read_raf.rs:726-760, the membership test) is equality with the naive
 * pushforward over the full eq tables, which tests/test_oracle_address.py re-runs. */
#include <stdlib.h>
#include <string.h>

#include "fr.h"

#define EXPORT __attribute__((visibility("default")))

void orc_eq_evals(const fr_t *r, size_t n, const fr_t *scale, fr_t *out);

/* fold_cycles: keys >= k_entries are cold cycles (NO_ACCESS) */
EXPORT void orc_fold_cycles(const uint64_t *keys, size_t cycles, size_t k_entries, const fr_t *w, fr_t *out) {
    for (size_t k = 0; k < k_entries; ++k) out[k] = fr_zero();
    for (size_t j = 0; j < cycles; ++j)
        if (keys[j] < k_entries) out[keys[j]] = FADD(out[keys[j]], w[j]);
}

/* stage_pushforwards for n_stages cycle points of log_t big-endian coordinates each (points: n_stages * log_t); every pc < k_entries
 * (the reference rejects anything else, :283-291); out: n_stages * k_entries */
EXPORT int orc_stage_pushforwards(const fr_t *points, size_t n_stages, size_t log_t, const uint64_t *pcs, size_t k_entries, fr_t *out) {
    const size_t lo_bits = log_t / 2, hi_bits = log_t - lo_bits;
    const size_t in_len = (size_t)1 << lo_bits, out_len = (size_t)1 << hi_bits;
    fr_t *e_hi = malloc(n_stages * out_len * sizeof(fr_t)), *e_lo = malloc(n_stages * in_len * sizeof(fr_t));
    fr_t *inner = malloc(n_stages * k_entries * sizeof(fr_t));
    size_t *touched = malloc(in_len * sizeof(size_t));
    uint32_t *seen = calloc(k_entries, sizeof(uint32_t));
    if (!e_hi || !e_lo || !inner || !touched || !seen) { free(e_hi); free(e_lo); free(inner); free(touched); free(seen); return -1; }
    for (size_t s = 0; s < n_stages; ++s) {
        orc_eq_evals(points + s * log_t, hi_bits, NULL, e_hi + s * out_len);
        orc_eq_evals(points + s * log_t + hi_bits, lo_bits, NULL, e_lo + s * in_len);
    }
    for (size_t i = 0; i < n_stages * k_entries; ++i) { out[i] = fr_zero(); inner[i] = fr_zero(); }
    size_t n_touched = 0;
    uint32_t epoch = 0;
    int bad = 0;
    for (size_t j_hi = 0; j_hi < out_len && !bad; ++j_hi) {
        for (size_t t = 0; t < n_touched; ++t)
            for (size_t s = 0; s < n_stages; ++s) inner[s * k_entries + touched[t]] = fr_zero();
        n_touched = 0;
        epoch += 1;
        const size_t base = j_hi << lo_bits;
        for (size_t j_lo = 0; j_lo < in_len; ++j_lo) {
            const uint64_t pc = pcs[base + j_lo];
            if (pc >= k_entries) { bad = 1; break; }
            if (seen[pc] != epoch) {
                seen[pc] = epoch;
                touched[n_touched++] = (size_t)pc;
            }
            for (size_t s = 0; s < n_stages; ++s) inner[s * k_entries + pc] = FADD(inner[s * k_entries + pc], e_lo[s * in_len + j_lo]);
        }
        for (size_t t = 0; t < n_touched; ++t)
            for (size_t s = 0; s < n_stages; ++s) {
                const size_t k = touched[t];
                out[s * k_entries + k] = FADD(out[s * k_entries + k], FMUL(e_hi[s * out_len + j_hi], inner[s * k_entries + k]));
            }
    }
    free(e_hi); free(e_lo); free(inner); free(touched); free(seen);
    return bad ? -2 : 0;
}

/* out[k] = the word written by the LAST access to k (post value), init[k] if k is never accessed */
EXPORT void orc_last_value(const uint64_t *keys, const uint64_t *post, size_t cycles, size_t k_entries, const fr_t *init, fr_t *out) {
    for (size_t k = 0; k < k_entries; ++k) out[k] = init[k];
    for (size_t j = 0; j < cycles; ++j)
        if (keys[j] < k_entries) out[keys[j]] = fr_from_u64(post[j]);
}
