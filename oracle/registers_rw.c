/*
 * oracle/registers_rw.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the optimized registers read/write-checking kernel (stage 4), paths relative to
 * /root/reference/crates/jolt-kernels/src/optimized/registers_read_write/:
 *   RegisterCycleRow::entries (<= 3 cells per cycle, rs2 folds into rs1's cell, rd into either read's)   sparse.rs:406-466
 *   SparseEntry::{bind, accumulate_pair_evals, merge_fill}                                             sparse.rs:228-380
 *   bind_sparse_entries / sparse_quadratic / SparseEntries::into_dense                                  sparse.rs:566-823
 *   ReadWriteKernel::{cycle_round_message, address_round_message, bind, one_hot_operand_claims}         mod.rs:205-372
 * Summand (reference/registers_read_write.rs:3-10):
 *   eq(r_cycle, j) * ( rd_wa * (rd_inc + val) + gamma * rs1_ra * val + gamma^2 * rs2_ra * val )(k, j)
 * over (register || cycle), bound low-to-high, the log T cycle variables first.  The coefficient columns are held as FIELD values (the
 * reference's `Direct` representation); its u16 lookup-table representation of the first rounds computes the same field values by
 * construction (sparse.rs:66-70) and is a memory layout, not a different result.
 *
 * PARITY UNPINNED by vectors: the reference pins this kernel by lock-step equality with its dense member
 * (registers_read_write/tests.rs via parity.rs:79-118); tests/test_oracle_registers.py re-runs that identity here (this restatement ==
 * the oracle's dense naive member over the materialised K x T grids, round for round, and the operand claims == the grids' evaluations).
 */
#include "fr.h"
#include <stdlib.h>

#define EXPORT __attribute__((visibility("default")))
#define REG_NONE 0xFF

typedef struct {
    uint64_t row;
    uint8_t col;
    uint64_t prev_u, next_u; /* register value just before / after this cell's row slice (sparse.rs:206-209) */
    fr_t val, ra, wa;        /* bound Val (value BEFORE the access), gamma * rs1_ra + gamma^2 * rs2_ra, rd_wa */
} reg_entry;

typedef struct {
    reg_entry *e;
    size_t n;
} reg_matrix;

static fr_t lerp(fr_t a, fr_t b, fr_t r) { return FADD(a, FMUL(r, FSUB(b, a))); }

/* sparse.rs:406-466: the cells of one cycle, sorted by register; ra seeds [0, g, g^2, g + g^2], wa seeds [0, 1] (mod.rs:139-143) */
EXPORT reg_matrix *orc_regrw_create(const uint8_t *rs1, const uint64_t *rs1_val, const uint8_t *rs2, const uint64_t *rs2_val, const uint8_t *rd, const uint64_t *rd_pre,
                                    const uint64_t *rd_post, size_t cycles, const fr_t *gamma) {
    reg_matrix *m = (reg_matrix *)calloc(1, sizeof(reg_matrix));
    m->e = (reg_entry *)calloc(3 * (cycles ? cycles : 1), sizeof(reg_entry));
    const fr_t g = *gamma, g2 = FMUL(g, g);
    for (size_t j = 0; j < cycles; ++j) {
        reg_entry out[3];
        size_t len = 0;
        memset(out, 0, sizeof out);
        if (rs1[j] != REG_NONE) {
            out[len].col = rs1[j]; out[len].prev_u = rs1_val[j]; out[len].next_u = rs1_val[j]; out[len].val = fr_from_u64(rs1_val[j]); out[len].ra = g;
            len++;
        }
        if (rs2[j] != REG_NONE) {
            size_t f = 0;
            while (f < len && out[f].col != rs2[j]) f++;
            if (f < len) out[f].ra = FADD(g, g2);
            else { out[len].col = rs2[j]; out[len].prev_u = rs2_val[j]; out[len].next_u = rs2_val[j]; out[len].val = fr_from_u64(rs2_val[j]); out[len].ra = g2; len++; }
        }
        if (rd[j] != REG_NONE) {
            size_t f = 0;
            while (f < len && out[f].col != rd[j]) f++;
            if (f < len) { out[f].wa = fr_one(); out[f].next_u = rd_post[j]; }
            else { out[len].col = rd[j]; out[len].prev_u = rd_pre[j]; out[len].next_u = rd_post[j]; out[len].val = fr_from_u64(rd_pre[j]); out[len].wa = fr_one(); len++; }
        }
        for (size_t a = 0; a < len; ++a) /* sort by column; len <= 3 */
            for (size_t b = a + 1; b < len; ++b)
                if (out[b].col < out[a].col) { reg_entry t = out[a]; out[a] = out[b]; out[b] = t; }
        for (size_t a = 0; a < len; ++a) { out[a].row = j; m->e[m->n++] = out[a]; }
    }
    return m;
}
EXPORT void orc_regrw_destroy(reg_matrix *m) { if (m) { free(m->e); free(m); } }
EXPORT size_t orc_regrw_len(const reg_matrix *m) { return m->n; }
EXPORT void orc_regrw_export(const reg_matrix *m, uint64_t *rows, uint64_t *cols, fr_t *val, fr_t *ra, fr_t *wa, uint64_t *prev, uint64_t *next) {
    for (size_t i = 0; i < m->n; ++i) {
        rows[i] = m->e[i].row; cols[i] = m->e[i].col; val[i] = m->e[i].val; ra[i] = m->e[i].ra; wa[i] = m->e[i].wa; prev[i] = m->e[i].prev_u; next[i] = m->e[i].next_u;
    }
}

/* SparseEntry::bind (sparse.rs:228-279); a missing side: Val = the neighbour's raw boundary value, ra = wa = 0 */
static reg_entry reg_bind(const reg_entry *even, const reg_entry *odd, fr_t r) {
    reg_entry o;
    memset(&o, 0, sizeof o);
    if (even && odd) {
        o.val = lerp(even->val, odd->val, r); o.ra = lerp(even->ra, odd->ra, r); o.wa = lerp(even->wa, odd->wa, r);
        o.prev_u = even->prev_u; o.next_u = odd->next_u; o.row = even->row / 2; o.col = even->col;
    } else if (even) {
        o.val = lerp(even->val, fr_from_u64(even->next_u), r); o.ra = FMUL(FSUB(fr_one(), r), even->ra); o.wa = FMUL(FSUB(fr_one(), r), even->wa);
        o.prev_u = even->prev_u; o.next_u = even->next_u; o.row = even->row / 2; o.col = even->col;
    } else {
        o.val = lerp(fr_from_u64(odd->prev_u), odd->val, r); o.ra = FMUL(r, odd->ra); o.wa = FMUL(r, odd->wa);
        o.prev_u = odd->prev_u; o.next_u = odd->next_u; o.row = odd->row / 2; o.col = odd->col;
    }
    return o;
}
/* SparseEntry::accumulate_pair_evals (sparse.rs:283-325): [t = 0, t = inf] of ra_t * val_t + wa_t * (val_t + inc_t) */
static void reg_pair_evals(const reg_entry *even, const reg_entry *odd, const fr_t inc_evals[2], fr_t acc[2]) {
    fr_t ra0, ra_m, wa0, wa_m, val0, val_m;
    if (even && odd) {
        ra0 = even->ra; ra_m = FSUB(odd->ra, even->ra); wa0 = even->wa; wa_m = FSUB(odd->wa, even->wa); val0 = even->val; val_m = FSUB(odd->val, even->val);
    } else if (even) {
        ra0 = even->ra; ra_m = FNEG(even->ra); wa0 = even->wa; wa_m = FNEG(even->wa); val0 = even->val; val_m = FSUB(fr_from_u64(even->next_u), even->val);
    } else {
        ra0 = fr_zero(); ra_m = odd->ra; wa0 = fr_zero(); wa_m = odd->wa; val0 = fr_zero(); val_m = FSUB(odd->val, fr_from_u64(odd->prev_u));
    }
    if (even) acc[0] = FADD(acc[0], FADD(FMUL(ra0, val0), FMUL(wa0, FADD(val0, inc_evals[0]))));
    acc[1] = FADD(acc[1], FADD(FMUL(ra_m, val_m), FMUL(wa_m, FADD(val_m, inc_evals[1]))));
}
/* sparse_quadratic (sparse.rs:689-823): eq head of a pair = e_out[pair >> in_bits] * e_in[pair & mask]; e_in of length <= 1 is the factor 1 */
EXPORT void orc_regrw_cycle_round(const reg_matrix *m, const fr_t *e_out, const fr_t *e_in, size_t e_in_len, const fr_t *inc, fr_t out[2]) {
    size_t in_bits = 0;
    while (((size_t)1 << in_bits) < e_in_len) in_bits++;
    const size_t mask = ((size_t)1 << in_bits) - 1;
    out[0] = fr_zero();
    out[1] = fr_zero();
    size_t i = 0;
    while (i < m->n) {
        size_t pair = m->e[i].row / 2, end = i;
        while (end < m->n && m->e[end].row / 2 == pair) end++;
        size_t odd_start = i;
        while (odd_start < end && m->e[odd_start].row % 2 == 0) odd_start++;
        const fr_t inc0 = inc[2 * pair];
        const fr_t inc_evals[2] = {inc0, FSUB(inc[2 * pair + 1], inc0)};
        fr_t inner[2] = {fr_zero(), fr_zero()};
        size_t a = i, b = odd_start;
        while (a < odd_start && b < end) {
            if (m->e[a].col == m->e[b].col) { reg_pair_evals(&m->e[a], &m->e[b], inc_evals, inner); a++; b++; }
            else if (m->e[a].col < m->e[b].col) { reg_pair_evals(&m->e[a], NULL, inc_evals, inner); a++; }
            else { reg_pair_evals(NULL, &m->e[b], inc_evals, inner); b++; }
        }
        for (; a < odd_start; ++a) reg_pair_evals(&m->e[a], NULL, inc_evals, inner);
        for (; b < end; ++b) reg_pair_evals(NULL, &m->e[b], inc_evals, inner);
        const fr_t head = FMUL(e_out[pair >> in_bits], e_in_len <= 1 ? fr_one() : e_in[pair & mask]);
        out[0] = FADD(out[0], FMUL(head, inner[0]));
        out[1] = FADD(out[1], FMUL(head, inner[1]));
        i = end;
    }
}
/* bind_sparse_entries (sparse.rs:566-683) */
EXPORT void orc_regrw_cycle_bind(reg_matrix *m, const fr_t *r) {
    reg_entry *out = (reg_entry *)calloc(m->n ? m->n : 1, sizeof(reg_entry));
    size_t k = 0, i = 0;
    while (i < m->n) {
        size_t pair = m->e[i].row / 2, end = i;
        while (end < m->n && m->e[end].row / 2 == pair) end++;
        size_t odd_start = i;
        while (odd_start < end && m->e[odd_start].row % 2 == 0) odd_start++;
        size_t a = i, b = odd_start;
        while (a < odd_start && b < end) {
            if (m->e[a].col == m->e[b].col) { out[k++] = reg_bind(&m->e[a], &m->e[b], *r); a++; b++; }
            else if (m->e[a].col < m->e[b].col) { out[k++] = reg_bind(&m->e[a], NULL, *r); a++; }
            else { out[k++] = reg_bind(NULL, &m->e[b], *r); b++; }
        }
        for (; a < odd_start; ++a) out[k++] = reg_bind(&m->e[a], NULL, *r);
        for (; b < end; ++b) out[k++] = reg_bind(NULL, &m->e[b], *r);
        i = end;
    }
    free(m->e);
    m->e = out;
    m->n = k;
}
/* SparseEntries::into_dense (sparse.rs:532-561): the fully cycle-bound single row scattered into K-sized arrays (untouched registers: 0) */
EXPORT int orc_regrw_into_dense(const reg_matrix *m, size_t k, fr_t *ra, fr_t *wa, fr_t *val) {
    for (size_t c = 0; c < k; ++c) { ra[c] = fr_zero(); wa[c] = fr_zero(); val[c] = fr_zero(); }
    for (size_t i = 0; i < m->n; ++i) {
        if (m->e[i].row != 0 || m->e[i].col >= k) return -1;
        ra[m->e[i].col] = m->e[i].ra; wa[m->e[i].col] = m->e[i].wa; val[m->e[i].col] = m->e[i].val;
    }
    return 0;
}
/* ReadWriteKernel::address_round_message (mod.rs:217-252): s(0..3) of eq_scalar * sum_y [ wa_t (inc + val_t) + ra_t val_t ] over low-to-high pairs */
EXPORT void orc_regrw_address_round(const fr_t *ra, const fr_t *wa, const fr_t *val, size_t len, const fr_t *inc_scalar, const fr_t *eq_scalar, fr_t evals[4]) {
    for (int t = 0; t < 4; ++t) evals[t] = fr_zero();
    for (size_t y = 0; y < len / 2; ++y) {
        fr_t ra_t = ra[2 * y], ra_m = FSUB(ra[2 * y + 1], ra_t), wa_t = wa[2 * y], wa_m = FSUB(wa[2 * y + 1], wa_t), val_t = val[2 * y], val_m = FSUB(val[2 * y + 1], val_t);
        for (int t = 0; t < 4; ++t) {
            evals[t] = FADD(evals[t], FADD(FMUL(wa_t, FADD(*inc_scalar, val_t)), FMUL(ra_t, val_t)));
            ra_t = FADD(ra_t, ra_m); wa_t = FADD(wa_t, wa_m); val_t = FADD(val_t, val_m);
        }
    }
    for (int t = 0; t < 4; ++t) evals[t] = FMUL(*eq_scalar, evals[t]);
}
/* one_hot_operand_claims (mod.rs:296-372) as the definition it implements: sum_j [idx_j hot] * eq(r_address, idx_j) * eq(r_cycle, j), both points
 * big-endian (the split into sqrt-sized tables there is an evaluation strategy) */
EXPORT void orc_regrw_operand_claim(const uint8_t *idx, size_t cycles, const fr_t *eq_address /* K */, const fr_t *eq_cycle /* cycles */, fr_t *out) {
    fr_t acc = fr_zero();
    for (size_t j = 0; j < cycles; ++j)
        if (idx[j] != REG_NONE) acc = FADD(acc, FMUL(eq_address[idx[j]], eq_cycle[j]));
    *out = acc;
}
