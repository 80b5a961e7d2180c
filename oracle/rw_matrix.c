/*
 * oracle/rw_matrix.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the sparse (K x T) read-write matrix of the optimized RAM read/write-checking kernel (SURVEY.md section 8f
 * row 4), paths relative to /root/reference/crates/jolt-kernels/src/optimized/:
 *   CycleMajorEntry::{bind, quadratic_evals, merge_bind_rows, merge_quadratic_evals}   rw_matrix.rs:70-253
 *   CycleMajorMatrix::{bind, quadratic_coefficients, into_address_major}                rw_matrix.rs:268-337
 *   AddressMajorEntry::{bind, address_round_evals, merge_*}                             rw_matrix.rs:339-588
 *   AddressMajorMatrix::{bind, address_round_evals, final_values}                       rw_matrix.rs:596-690
 *   RamReadWriteKernel (phase machine, entry construction, round messages)              ram_read_write.rs:58-330
 * Summand: eq(tau_low, j) * ra(k,j) * (val(k,j) + gamma * (val(k,j) + inc(j))) over (address || cycle), bound low-to-high, cycle
 * variables first; the dense reference member is reference/ram_read_write.rs:29-69.
 *
 * PARITY UNPINNED by vectors: the reference pins this kernel by lock-step equality with its dense reference member
 * (optimized/ram_read_write.rs tests via parity.rs:79-118); tests/test_oracle_rw.py re-runs that identity here (sparse
 * restatement == the oracle's dense naive member over the materialised K x T grids, round for round).
 */
#include "fr.h"
#include <stdlib.h>

#define EXPORT __attribute__((visibility("default")))
#define RW_NO_ACCESS 0xFFFFFFFFFFFFFFFFull /* ram_trace.rs:22 */

typedef struct {
    uint64_t row, col;
    uint64_t prev_u, next_u; /* cycle phase checkpoints: raw memory values (rw_matrix.rs:29-42) */
    fr_t prev_f, next_f;     /* address phase checkpoints (rw_matrix.rs:47-54) */
    fr_t val, ra;
} rw_entry;

typedef struct {
    rw_entry *e;
    size_t n;
    int address_major;
} rw_matrix;

static fr_t slope_term(fr_t val, fr_t inc, fr_t gamma) { return FADD(val, FMUL(gamma, FADD(inc, val))); } /* rw_matrix.rs:257-260 */

/* ram_read_write.rs:291-306: one entry per RAM access, row = cycle, col = address, val = pre value, ra = 1 */
EXPORT rw_matrix *orc_rw_create(const uint64_t *addresses, const uint64_t *pre, const uint64_t *post, size_t cycles) {
    rw_matrix *m = (rw_matrix *)calloc(1, sizeof(rw_matrix));
    m->e = (rw_entry *)calloc(cycles ? cycles : 1, sizeof(rw_entry));
    for (size_t j = 0; j < cycles; ++j) {
        if (addresses[j] == RW_NO_ACCESS) continue;
        rw_entry *x = &m->e[m->n++];
        x->row = j;
        x->col = addresses[j];
        x->prev_u = pre[j];
        x->next_u = post[j];
        x->val = fr_from_u64(pre[j]);
        x->ra = fr_one();
    }
    return m;
}
EXPORT void orc_rw_destroy(rw_matrix *m) { if (m) { free(m->e); free(m); } }
EXPORT size_t orc_rw_len(const rw_matrix *m) { return m->n; }
/* rows, cols (u64), val, ra (Fr), checkpoints as Fr (raw values promoted in the cycle phase) */
EXPORT void orc_rw_export(const rw_matrix *m, uint64_t *rows, uint64_t *cols, fr_t *val, fr_t *ra, fr_t *prev, fr_t *next) {
    for (size_t i = 0; i < m->n; ++i) {
        rows[i] = m->e[i].row;
        cols[i] = m->e[i].col;
        val[i] = m->e[i].val;
        ra[i] = m->e[i].ra;
        prev[i] = m->address_major ? m->e[i].prev_f : fr_from_u64(m->e[i].prev_u);
        next[i] = m->address_major ? m->e[i].next_f : fr_from_u64(m->e[i].next_u);
    }
}

/* ---- cycle phase --------------------------------------------------------------------------------------------------- */
static rw_entry cyc_bind(const rw_entry *even, const rw_entry *odd, fr_t r) { /* rw_matrix.rs:70-110 */
    rw_entry o;
    memset(&o, 0, sizeof o);
    if (even && odd) {
        o.row = even->row / 2; o.col = even->col;
        o.ra = FADD(even->ra, FMUL(r, FSUB(odd->ra, even->ra)));
        o.val = FADD(even->val, FMUL(r, FSUB(odd->val, even->val)));
        o.prev_u = even->prev_u; o.next_u = odd->next_u;
    } else if (even) {
        fr_t odd_val = fr_from_u64(even->next_u);
        o.row = even->row / 2; o.col = even->col;
        o.ra = FMUL(FSUB(fr_one(), r), even->ra);
        o.val = FADD(even->val, FMUL(r, FSUB(odd_val, even->val)));
        o.prev_u = even->prev_u; o.next_u = even->next_u;
    } else {
        fr_t even_val = fr_from_u64(odd->prev_u);
        o.row = odd->row / 2; o.col = odd->col;
        o.ra = FMUL(r, odd->ra);
        o.val = FADD(even_val, FMUL(r, FSUB(odd->val, even_val)));
        o.prev_u = odd->prev_u; o.next_u = odd->next_u;
    }
    return o;
}
static void cyc_quadratic(const rw_entry *even, const rw_entry *odd, const fr_t inc_evals[2], fr_t gamma, fr_t out[2]) { /* rw_matrix.rs:116-145 */
    if (even && odd) {
        out[0] = FMUL(even->ra, slope_term(even->val, inc_evals[0], gamma));
        out[1] = FMUL(FSUB(odd->ra, even->ra), slope_term(FSUB(odd->val, even->val), inc_evals[1], gamma));
    } else if (even) {
        fr_t odd_val = fr_from_u64(even->next_u);
        out[0] = FMUL(even->ra, slope_term(even->val, inc_evals[0], gamma));
        out[1] = FMUL(FNEG(even->ra), slope_term(FSUB(odd_val, even->val), inc_evals[1], gamma));
    } else {
        fr_t even_val = fr_from_u64(odd->prev_u);
        out[0] = fr_zero();
        out[1] = FMUL(odd->ra, slope_term(FSUB(odd->val, even_val), inc_evals[1], gamma));
    }
}
/* CycleMajorMatrix::quadratic_coefficients (rw_matrix.rs:287-325): eq_head(pair) = e_out[pair >> in_bits] * e_in[pair & mask]
 * (ram_read_write.rs:163-171) */
EXPORT void orc_rw_cycle_round(const rw_matrix *m, const fr_t *e_out, const fr_t *e_in, size_t in_bits, const fr_t *inc, const fr_t *gamma, fr_t out[2]) {
    out[0] = fr_zero();
    out[1] = fr_zero();
    size_t mask = ((size_t)1 << in_bits) - 1, i = 0;
    while (i < m->n) {
        size_t pair = m->e[i].row / 2, end = i;
        while (end < m->n && m->e[end].row / 2 == pair) end++;
        size_t odd_start = i;
        while (odd_start < end && m->e[odd_start].row % 2 == 0) odd_start++;
        fr_t inc0 = inc[2 * pair];
        fr_t inc_evals[2] = {inc0, FSUB(inc[2 * pair + 1], inc0)};
        fr_t inner[2] = {fr_zero(), fr_zero()}, c[2];
        size_t a = i, b = odd_start;
        while (a < odd_start && b < end) { /* merge_quadratic_evals, rw_matrix.rs:190-253 */
            if (m->e[a].col == m->e[b].col) { cyc_quadratic(&m->e[a], &m->e[b], inc_evals, *gamma, c); a++; b++; }
            else if (m->e[a].col < m->e[b].col) { cyc_quadratic(&m->e[a], NULL, inc_evals, *gamma, c); a++; }
            else { cyc_quadratic(NULL, &m->e[b], inc_evals, *gamma, c); b++; }
            inner[0] = FADD(inner[0], c[0]); inner[1] = FADD(inner[1], c[1]);
        }
        for (; a < odd_start; ++a) { cyc_quadratic(&m->e[a], NULL, inc_evals, *gamma, c); inner[0] = FADD(inner[0], c[0]); inner[1] = FADD(inner[1], c[1]); }
        for (; b < end; ++b) { cyc_quadratic(NULL, &m->e[b], inc_evals, *gamma, c); inner[0] = FADD(inner[0], c[0]); inner[1] = FADD(inner[1], c[1]); }
        fr_t head = FMUL(e_out[pair >> in_bits], e_in[pair & mask]);
        out[0] = FADD(out[0], FMUL(head, inner[0]));
        out[1] = FADD(out[1], FMUL(head, inner[1]));
        i = end;
    }
}
/* CycleMajorMatrix::bind (rw_matrix.rs:268-285) via merge_bind_rows (:150-185) */
EXPORT void orc_rw_cycle_bind(rw_matrix *m, const fr_t *r) {
    rw_entry *out = (rw_entry *)calloc(m->n ? m->n : 1, sizeof(rw_entry));
    size_t k = 0, i = 0;
    while (i < m->n) {
        size_t pair = m->e[i].row / 2, end = i;
        while (end < m->n && m->e[end].row / 2 == pair) end++;
        size_t odd_start = i;
        while (odd_start < end && m->e[odd_start].row % 2 == 0) odd_start++;
        size_t a = i, b = odd_start;
        while (a < odd_start && b < end) {
            if (m->e[a].col == m->e[b].col) { out[k++] = cyc_bind(&m->e[a], &m->e[b], *r); a++; b++; }
            else if (m->e[a].col < m->e[b].col) { out[k++] = cyc_bind(&m->e[a], NULL, *r); a++; }
            else { out[k++] = cyc_bind(NULL, &m->e[b], *r); b++; }
        }
        for (; a < odd_start; ++a) out[k++] = cyc_bind(&m->e[a], NULL, *r);
        for (; b < end; ++b) out[k++] = cyc_bind(NULL, &m->e[b], *r);
        i = end;
    }
    free(m->e);
    m->e = out;
    m->n = k;
}
/* CycleMajorMatrix::into_address_major (rw_matrix.rs:327-337): every row is 0, only the checkpoint representation changes */
EXPORT int orc_rw_into_address_major(rw_matrix *m) {
    for (size_t i = 0; i < m->n; ++i) {
        if (m->e[i].row != 0) return -1;
        m->e[i].prev_f = fr_from_u64(m->e[i].prev_u);
        m->e[i].next_f = fr_from_u64(m->e[i].next_u);
    }
    m->address_major = 1;
    return 0;
}

/* ---- address phase --------------------------------------------------------------------------------------------------- */
static rw_entry adr_bind(const rw_entry *even, const rw_entry *odd, fr_t even_cp, fr_t odd_cp, fr_t r) { /* rw_matrix.rs:342-383 */
    rw_entry o;
    memset(&o, 0, sizeof o);
    if (even && odd) {
        o.row = even->row; o.col = even->col / 2;
        o.ra = FADD(even->ra, FMUL(r, FSUB(odd->ra, even->ra)));
        o.val = FADD(even->val, FMUL(r, FSUB(odd->val, even->val)));
        o.prev_f = FADD(even->prev_f, FMUL(r, FSUB(odd->prev_f, even->prev_f)));
        o.next_f = FADD(even->next_f, FMUL(r, FSUB(odd->next_f, even->next_f)));
    } else if (even) {
        o.row = even->row; o.col = even->col / 2;
        o.ra = FMUL(FSUB(fr_one(), r), even->ra);
        o.val = FADD(even->val, FMUL(r, FSUB(odd_cp, even->val)));
        o.prev_f = FADD(even->prev_f, FMUL(r, FSUB(odd_cp, even->prev_f)));
        o.next_f = FADD(even->next_f, FMUL(r, FSUB(odd_cp, even->next_f)));
    } else {
        o.row = odd->row; o.col = odd->col / 2;
        o.ra = FMUL(r, odd->ra);
        o.val = FADD(even_cp, FMUL(r, FSUB(odd->val, even_cp)));
        o.prev_f = FADD(even_cp, FMUL(r, FSUB(odd->prev_f, even_cp)));
        o.next_f = FADD(even_cp, FMUL(r, FSUB(odd->next_f, even_cp)));
    }
    return o;
}
static void adr_evals(const rw_entry *even, const rw_entry *odd, fr_t even_cp, fr_t odd_cp, fr_t inc_eval, fr_t eq_eval, fr_t gamma, fr_t out[2]) { /* :388-419 */
    if (even && odd) {
        fr_t ra2 = FSUB(FADD(odd->ra, odd->ra), even->ra), val2 = FSUB(FADD(odd->val, odd->val), even->val);
        out[0] = FMUL(FMUL(eq_eval, even->ra), slope_term(even->val, inc_eval, gamma));
        out[1] = FMUL(FMUL(eq_eval, ra2), slope_term(val2, inc_eval, gamma));
    } else if (even) {
        fr_t val2 = FSUB(FADD(odd_cp, odd_cp), even->val);
        out[0] = FMUL(FMUL(eq_eval, even->ra), slope_term(even->val, inc_eval, gamma));
        out[1] = FMUL(FMUL(eq_eval, FNEG(even->ra)), slope_term(val2, inc_eval, gamma));
    } else {
        out[0] = fr_zero();
        out[1] = FMUL(FMUL(eq_eval, FADD(odd->ra, odd->ra)), slope_term(FSUB(FADD(odd->val, odd->val), even_cp), inc_eval, gamma));
    }
}
/* AddressMajorMatrix::address_round_evals (rw_matrix.rs:641-676): [s(0), s(2)]; inc / eq indexed by row (cycle-bound: length 1) */
EXPORT void orc_rw_address_round(const rw_matrix *m, const fr_t *val_init, const fr_t *inc, const fr_t *eq, const fr_t *gamma, fr_t out[2]) {
    out[0] = fr_zero();
    out[1] = fr_zero();
    size_t i = 0;
    while (i < m->n) {
        size_t pair = m->e[i].col / 2, end = i;
        while (end < m->n && m->e[end].col / 2 == pair) end++;
        size_t odd_start = i;
        while (odd_start < end && m->e[odd_start].col % 2 == 0) odd_start++;
        fr_t even_cp = val_init[2 * pair], odd_cp = val_init[2 * pair + 1], c[2];
        size_t a = i, b = odd_start;
        while (a < odd_start && b < end) { /* merge_address_round_evals, rw_matrix.rs:497-588 */
            if (m->e[a].row == m->e[b].row) {
                adr_evals(&m->e[a], &m->e[b], even_cp, odd_cp, inc[m->e[a].row], eq[m->e[a].row], *gamma, c);
                even_cp = m->e[a].next_f; odd_cp = m->e[b].next_f; a++; b++;
            } else if (m->e[a].row < m->e[b].row) {
                adr_evals(&m->e[a], NULL, even_cp, odd_cp, inc[m->e[a].row], eq[m->e[a].row], *gamma, c);
                even_cp = m->e[a].next_f; a++;
            } else {
                adr_evals(NULL, &m->e[b], even_cp, odd_cp, inc[m->e[b].row], eq[m->e[b].row], *gamma, c);
                odd_cp = m->e[b].next_f; b++;
            }
            out[0] = FADD(out[0], c[0]); out[1] = FADD(out[1], c[1]);
        }
        for (; a < odd_start; ++a) { adr_evals(&m->e[a], NULL, even_cp, odd_cp, inc[m->e[a].row], eq[m->e[a].row], *gamma, c); even_cp = m->e[a].next_f; out[0] = FADD(out[0], c[0]); out[1] = FADD(out[1], c[1]); }
        for (; b < end; ++b) { adr_evals(NULL, &m->e[b], even_cp, odd_cp, inc[m->e[b].row], eq[m->e[b].row], *gamma, c); odd_cp = m->e[b].next_f; out[0] = FADD(out[0], c[0]); out[1] = FADD(out[1], c[1]); }
        i = end;
    }
}
/* AddressMajorMatrix::bind (rw_matrix.rs:599-637): merge every adjacent column pair against the val_init checkpoints, then bind
 * val_init itself low-to-high (in place, len -> len / 2) */
EXPORT void orc_rw_address_bind(rw_matrix *m, const fr_t *r, fr_t *val_init, size_t val_init_len) {
    rw_entry *out = (rw_entry *)calloc(m->n ? m->n : 1, sizeof(rw_entry));
    size_t k = 0, i = 0;
    while (i < m->n) {
        size_t pair = m->e[i].col / 2, end = i;
        while (end < m->n && m->e[end].col / 2 == pair) end++;
        size_t odd_start = i;
        while (odd_start < end && m->e[odd_start].col % 2 == 0) odd_start++;
        fr_t even_cp = val_init[2 * pair], odd_cp = val_init[2 * pair + 1];
        size_t a = i, b = odd_start;
        while (a < odd_start && b < end) { /* merge_bind_cols, rw_matrix.rs:429-495 */
            if (m->e[a].row == m->e[b].row) { out[k++] = adr_bind(&m->e[a], &m->e[b], even_cp, odd_cp, *r); even_cp = m->e[a].next_f; odd_cp = m->e[b].next_f; a++; b++; }
            else if (m->e[a].row < m->e[b].row) { out[k++] = adr_bind(&m->e[a], NULL, even_cp, odd_cp, *r); even_cp = m->e[a].next_f; a++; }
            else { out[k++] = adr_bind(NULL, &m->e[b], even_cp, odd_cp, *r); odd_cp = m->e[b].next_f; b++; }
        }
        for (; a < odd_start; ++a) { out[k++] = adr_bind(&m->e[a], NULL, even_cp, odd_cp, *r); even_cp = m->e[a].next_f; }
        for (; b < end; ++b) { out[k++] = adr_bind(NULL, &m->e[b], even_cp, odd_cp, *r); odd_cp = m->e[b].next_f; }
        i = end;
    }
    free(m->e);
    m->e = out;
    m->n = k;
    for (size_t y = 0; y < val_init_len / 2; ++y) { /* dense.rs:223-263, LowToHigh */
        fr_t lo = val_init[2 * y], hi = val_init[2 * y + 1];
        val_init[y] = FADD(lo, FMUL(*r, FSUB(hi, lo)));
    }
}
/* AddressMajorMatrix::final_values (rw_matrix.rs:680-688) */
EXPORT void orc_rw_final_values(const rw_matrix *m, const fr_t *val_init, fr_t *ra_out, fr_t *val_out) {
    if (m->n) { *ra_out = m->e[0].ra; *val_out = m->e[0].val; }
    else { *ra_out = fr_zero(); *val_out = val_init[0]; }
}
