// jolt_amd/csrc/read_raf_address.hip -- the 128 address rounds of instruction read+RAF checking (stage 5), host side of the device scans.
//
// Replaces, together with read_raf.hip, the address half of OptimizedInstructionReadRafKernel (crates/jolt-kernels/src/optimized/instruction_read_raf.rs):
//   init_phase (:747-900)        the T-scale sums come from jolt_read_raf_phase_scan (device); here the 256-entry prefix polynomials are built from the
//                                checkpoints (lookup_tables.hpp) and the four RAF decompositions (left / right operand, identity, upper-all-ones) assembled
//   address_message (:973-1050)  s(0), s(2) over the live half of the chunk domain, s(1) = previous_claim - s(0)
//   bind (:1235-1282)            HighToLow bind of every 256-entry polynomial; after 8 binds the phase's eq table and the new checkpoints
//   init_cycle_rounds (:1140-1160)  table values, gamma-combined operand values for jolt_read_raf_cycle_tables
// Per proof this is 128 rounds of O(256 x present tables) field work -- no T-sized data -- which is why it stays on the host (the reference keeps
// it in one rayon task per 8 entries); the device part of a phase is one scan launch + one condensation.  Nothing here touches the device.
#include <cstring>
#include <new>
#include <vector>

#include "ctx.hpp"
#include "lookup_tables.hpp"

using jolt::Fr;
using namespace jolt_lookup;

namespace {

constexpr uint32_t kChunkLen = 8, kChunkSize = 1u << kChunkLen;

// prefix * q_shift + q_value, each a 256-entry polynomial of the chunk (instruction_read_raf.rs:387-445)
struct RafDecomposition {
    std::vector<Fr> prefix, q_shift, q_value;
    Fr checkpoint;
    void message(size_t b, size_t half, Fr& at0, Fr& at2) const {
        const Fr p2 = jolt::sub(jolt::dbl(prefix[b + half]), prefix[b]), s2 = jolt::sub(jolt::dbl(q_shift[b + half]), q_shift[b]),
                 v2 = jolt::sub(jolt::dbl(q_value[b + half]), q_value[b]);
        at0 = jolt::add(jolt::mul(prefix[b], q_shift[b]), q_value[b]);
        at2 = jolt::add(jolt::mul(p2, s2), v2);
    }
};

void bind_high_to_low(std::vector<Fr>& t, size_t half, const Fr& r) {
    for (size_t b = 0; b < half; ++b) t[b] = jolt::add(t[b], jolt::mul(r, jolt::sub(t[b + half], t[b])));
}

// EqPolynomial::evals, big-endian (crates/jolt-poly/src/eq.rs:299-315)
std::vector<Fr> eq_table(const std::vector<Fr>& point) {
    std::vector<Fr> e((size_t)1 << point.size());
    e[0] = Fr::one();
    size_t size = 1;
    for (const Fr& r : point) {
        for (size_t i = size; i-- > 0;) {
            const Fr hi = jolt::mul(e[i], r);
            e[2 * i + 1] = hi;
            e[2 * i] = jolt::sub(e[i], hi);
        }
        size *= 2;
    }
    return e;
}

}  // namespace

struct jolt_read_raf_address {
    Fr gamma;
    bool canonical;
    std::vector<uint8_t> present;          // table ids with at least one row
    std::vector<uint8_t> prefix_indices;   // prefixes those tables read
    Fr checkpoints[kNumPrefixes];
    std::vector<std::vector<Fr>> prefix_tables;               // [position in prefix_indices][256]
    std::vector<std::vector<std::vector<Fr>>> suffix_tables;  // [position in present][suffix][256]
    RafDecomposition left, right, identity, upper;
    std::vector<Fr> phase_challenges;
    std::vector<std::vector<Fr>> v_tables;  // completed phases' eq tables
    uint32_t phase = 0;
    bool phase_open = false;
    uint32_t suffix_offsets[kNumTables + 1];
};

extern "C" uint32_t jolt_lookup_table_count(void) { return kNumTables; }
extern "C" uint32_t jolt_lookup_prefix_count(void) { return kNumPrefixes; }

extern "C" int32_t jolt_lookup_table_suffixes(uint32_t kind, uint8_t* kinds_out, uint32_t* n_out) {
    if (kind >= (uint32_t)kNumTables || !kinds_out || !n_out) return JOLT_ERR_INVALID_ARG;
    const TableDesc& t = table_descs()[kind];
    std::memcpy(kinds_out, t.suffixes, t.n_suffixes);
    *n_out = t.n_suffixes;
    return JOLT_OK;
}
extern "C" int32_t jolt_lookup_table_prefixes(uint32_t kind, uint8_t* prefixes_out, uint32_t* n_out) {
    if (kind >= (uint32_t)kNumTables || !prefixes_out || !n_out) return JOLT_ERR_INVALID_ARG;
    const TableDesc& t = table_descs()[kind];
    std::memcpy(prefixes_out, t.prefixes, t.n_prefixes);
    *n_out = t.n_prefixes;
    return JOLT_OK;
}
extern "C" int32_t jolt_host_lookup_prefix_default_checkpoints(jolt_fr_t* out) {
    if (!out) return JOLT_ERR_INVALID_ARG;
    for (int p = 0; p < kNumPrefixes; ++p) fr_to_abi(&out[p], prefix_default_checkpoint(p));
    return JOLT_OK;
}
extern "C" int32_t jolt_host_lookup_prefix_evaluate(uint32_t prefix, const jolt_fr_t* checkpoints, uint32_t b, uint32_t b_len, uint32_t suffix_len, jolt_fr_t* out) {
    if (prefix >= (uint32_t)kNumPrefixes || !checkpoints || !out || b_len == 0 || b_len > 16 || (b_len & 1) || suffix_len + b_len > (uint32_t)kLogK) return JOLT_ERR_INVALID_ARG;
    Fr cp[kNumPrefixes];
    for (int p = 0; p < kNumPrefixes; ++p) cp[p] = fr_from_abi(&checkpoints[p]);
    fr_to_abi(out, prefix_evaluate(prefix, cp, b, b_len, suffix_len));
    return JOLT_OK;
}
// one prefix over a whole chunk domain: out[x] = evaluate(checkpoints, x, suffix_len) for x < 2^b_len (what init_phase materialises, :878-897)
extern "C" int32_t jolt_host_lookup_prefix_table(uint32_t prefix, const jolt_fr_t* checkpoints, uint32_t b_len, uint32_t suffix_len, jolt_fr_t* out) {
    if (prefix >= (uint32_t)kNumPrefixes || !checkpoints || !out || b_len == 0 || b_len > 16 || (b_len & 1) || suffix_len + b_len > (uint32_t)kLogK) return JOLT_ERR_INVALID_ARG;
    Fr cp[kNumPrefixes];
    for (int p = 0; p < kNumPrefixes; ++p) cp[p] = fr_from_abi(&checkpoints[p]);
    for (uint32_t x = 0; x < (1u << b_len); ++x) fr_to_abi(&out[x], prefix_evaluate(prefix, cp, x, b_len, suffix_len));
    return JOLT_OK;
}
extern "C" int32_t jolt_host_lookup_table_combine(uint32_t kind, const jolt_fr_t* prefixes, const jolt_fr_t* suffixes, jolt_fr_t* out) {
    if (kind >= (uint32_t)kNumTables || !prefixes || !suffixes || !out) return JOLT_ERR_INVALID_ARG;
    const TableDesc& t = table_descs()[kind];
    Fr p[kNumPrefixes], s[5];
    for (int i = 0; i < kNumPrefixes; ++i) p[i] = fr_from_abi(&prefixes[i]);
    for (uint32_t i = 0; i < t.n_suffixes; ++i) s[i] = fr_from_abi(&suffixes[i]);
    fr_to_abi(out, table_combine(t, p, s));
    return JOLT_OK;
}

extern "C" int32_t jolt_host_read_raf_address_create(const jolt_fr_t* gamma, const uint8_t* table_present, int32_t canonical, jolt_read_raf_address** out) {
    if (!gamma || !table_present || !out) return JOLT_ERR_INVALID_ARG;
    auto* h = new (std::nothrow) jolt_read_raf_address();
    if (!h) return JOLT_ERR_OOM;
    h->gamma = fr_from_abi(gamma);
    if (!fr_is_canonical(h->gamma)) { delete h; return JOLT_ERR_INVALID_ARG; }
    h->canonical = canonical != 0;
    bool reads[kNumPrefixes] = {};
    h->suffix_offsets[0] = 0;
    for (int t = 0; t < kNumTables; ++t) {
        const TableDesc& d = table_descs()[t];
        h->suffix_offsets[t + 1] = h->suffix_offsets[t] + d.n_suffixes;
        if (!table_present[t]) continue;
        h->present.push_back((uint8_t)t);
        for (uint32_t k = 0; k < d.n_prefixes; ++k) reads[d.prefixes[k]] = true;
    }
    for (int p = 0; p < kNumPrefixes; ++p) {
        h->checkpoints[p] = prefix_default_checkpoint(p);
        if (reads[p]) h->prefix_indices.push_back((uint8_t)p);
    }
    h->left.checkpoint = h->right.checkpoint = h->identity.checkpoint = Fr::zero();
    h->upper.checkpoint = Fr::one();  // an AND over address bits: the empty product (:417-423)
    *out = h;
    return JOLT_OK;
}
extern "C" int32_t jolt_host_read_raf_address_destroy(jolt_read_raf_address* h) {
    delete h;
    return JOLT_OK;
}
// offsets of a table's suffix accumulators in the flattened layout jolt_read_raf_phase_scan writes (n_tables = 42, LookupTableKind order)
extern "C" int32_t jolt_lookup_suffix_layout(uint32_t* offsets_out /* 43 */, uint8_t* kinds_out /* offsets[42] */) {
    if (!offsets_out) return JOLT_ERR_INVALID_ARG;
    uint32_t at = 0;
    for (int t = 0; t < kNumTables; ++t) {
        const TableDesc& d = table_descs()[t];
        offsets_out[t] = at;
        if (kinds_out) std::memcpy(kinds_out + at, d.suffixes, d.n_suffixes);
        at += d.n_suffixes;
    }
    offsets_out[kNumTables] = at;
    return JOLT_OK;
}

// init_phase: raf_sums[q * 256 + chunk], q = left, right, identity, shift_half, shift_full, upper_all_ones (raw, as jolt_read_raf_phase_scan returns them);
// suffix_sums[(offsets[t] + s) * 256 + chunk] in the layout of jolt_lookup_suffix_layout.
extern "C" int32_t jolt_host_read_raf_address_init_phase(jolt_read_raf_address* h, uint32_t phase, const jolt_fr_t* raf_sums, const jolt_fr_t* suffix_sums) {
    if (!h || !raf_sums || !suffix_sums || phase != h->phase || h->phase_open || phase >= (uint32_t)kLogK / kChunkLen) return JOLT_ERR_INVALID_ARG;
    const uint32_t suffix_len = kLogK - (phase + 1) * kChunkLen;
    auto column = [&](uint32_t q) {
        std::vector<Fr> v(kChunkSize);
        for (uint32_t x = 0; x < kChunkSize; ++x) v[x] = fr_from_abi(&raf_sums[(size_t)q * kChunkSize + x]);
        return v;
    };
    const Fr half_scale = fr_pow2(suffix_len / 2), full_scale = fr_pow2(suffix_len);
    std::vector<Fr> q_shift_half = column(3), q_shift_full = column(4);
    for (Fr& v : q_shift_half) v = jolt::mul(v, half_scale);
    for (Fr& v : q_shift_full) v = jolt::mul(v, full_scale);
    // operand prefixes: the bound part moves up by the chunk's share of bits, the chunk's own bits are added (:826-842)
    const Fr up_half = fr_pow2(kChunkLen / 2), up_full = fr_pow2(kChunkLen);
    h->left.prefix.assign(kChunkSize, Fr::zero());
    h->right.prefix.assign(kChunkSize, Fr::zero());
    h->identity.prefix.assign(kChunkSize, Fr::zero());
    for (uint32_t x = 0; x < kChunkSize; ++x) {
        const Chunk c = make_chunk(x, kChunkLen, suffix_len);
        h->left.prefix[x] = jolt::add(jolt::mul(h->left.checkpoint, up_half), jolt::fr_from_u64(c.x));
        h->right.prefix[x] = jolt::add(jolt::mul(h->right.checkpoint, up_half), jolt::fr_from_u64(c.y));
        h->identity.prefix[x] = jolt::add(jolt::mul(h->identity.checkpoint, up_full), jolt::fr_from_u64(x));
    }
    h->left.q_shift = q_shift_half;
    h->left.q_value = column(0);
    h->right.q_shift = q_shift_half;
    h->right.q_value = column(1);
    h->identity.q_shift = q_shift_full;
    h->identity.q_value = column(2);
    if (h->canonical) {  // the chunk's share of the upper word must be all ones (:857-876)
        const uint32_t done = phase * kChunkLen, upper_bits = (uint32_t)kLogK / 2 > done ? ((uint32_t)kLogK / 2 - done < kChunkLen ? (uint32_t)kLogK / 2 - done : kChunkLen) : 0;
        h->upper.prefix.assign(kChunkSize, Fr::zero());
        for (uint32_t x = 0; x < kChunkSize; ++x)
            if (upper_bits == 0 || (x >> (kChunkLen - upper_bits)) == (1u << upper_bits) - 1) h->upper.prefix[x] = h->upper.checkpoint;
        h->upper.q_shift = column(5);
        h->upper.q_value.assign(kChunkSize, Fr::zero());
    }
    h->suffix_tables.clear();
    for (uint8_t t : h->present) {
        const TableDesc& d = table_descs()[t];
        std::vector<std::vector<Fr>> polys(d.n_suffixes, std::vector<Fr>(kChunkSize));
        for (uint32_t s = 0; s < d.n_suffixes; ++s)
            for (uint32_t x = 0; x < kChunkSize; ++x) polys[s][x] = fr_from_abi(&suffix_sums[((size_t)h->suffix_offsets[t] + s) * kChunkSize + x]);
        h->suffix_tables.push_back(std::move(polys));
    }
    h->prefix_tables.clear();
    for (uint8_t p : h->prefix_indices) {
        std::vector<Fr> table(kChunkSize);
        for (uint32_t x = 0; x < kChunkSize; ++x) table[x] = prefix_evaluate(p, h->checkpoints, x, kChunkLen, suffix_len);
        h->prefix_tables.push_back(std::move(table));
    }
    h->phase_challenges.clear();
    h->phase_open = true;
    return JOLT_OK;
}

// address_message: evals_out = s(0), s(1), s(2) (UnivariatePoly::from_evals order)
extern "C" int32_t jolt_host_read_raf_address_message(const jolt_read_raf_address* h, const jolt_fr_t* previous_claim, jolt_fr_t* evals_out) {
    if (!h || !previous_claim || !evals_out || !h->phase_open) return JOLT_ERR_INVALID_ARG;
    const size_t half = (kChunkSize >> h->phase_challenges.size()) / 2;
    Fr read0 = Fr::zero(), read2 = Fr::zero(), sums[8];
    for (Fr& s : sums) s = Fr::zero();
    Fr p0[kNumPrefixes], p2[kNumPrefixes], s0[5], s2[5];
    for (int p = 0; p < kNumPrefixes; ++p) p0[p] = p2[p] = Fr::zero();
    for (size_t b = 0; b < half; ++b) {
        for (size_t i = 0; i < h->prefix_indices.size(); ++i) {
            const std::vector<Fr>& t = h->prefix_tables[i];
            p0[h->prefix_indices[i]] = t[b];
            p2[h->prefix_indices[i]] = jolt::sub(jolt::dbl(t[b + half]), t[b]);
        }
        for (size_t i = 0; i < h->present.size(); ++i) {
            const TableDesc& d = table_descs()[h->present[i]];
            for (uint32_t s = 0; s < d.n_suffixes; ++s) {
                const std::vector<Fr>& q = h->suffix_tables[i][s];
                s0[s] = q[b];
                s2[s] = jolt::sub(jolt::dbl(q[b + half]), q[b]);
            }
            read0 = jolt::add(read0, table_combine(d, p0, s0));
            read2 = jolt::add(read2, table_combine(d, p2, s2));
        }
        Fr a0, a2;
        h->left.message(b, half, a0, a2);
        sums[0] = jolt::add(sums[0], a0);
        sums[1] = jolt::add(sums[1], a2);
        h->right.message(b, half, a0, a2);
        sums[2] = jolt::add(sums[2], a0);
        sums[3] = jolt::add(sums[3], a2);
        h->identity.message(b, half, a0, a2);
        sums[4] = jolt::add(sums[4], a0);
        sums[5] = jolt::add(sums[5], a2);
        if (h->canonical) {
            h->upper.message(b, half, a0, a2);
            sums[6] = jolt::add(sums[6], a0);
            sums[7] = jolt::add(sums[7], a2);
        }
    }
    const Fr g = h->gamma, g2 = jolt::mul(g, g), g3 = jolt::mul(g2, g);
    Fr e0 = jolt::add(read0, jolt::add(jolt::mul(g, sums[0]), jolt::mul(g2, jolt::add(sums[2], sums[4]))));
    Fr e2 = jolt::add(read2, jolt::add(jolt::mul(g, sums[1]), jolt::mul(g2, jolt::add(sums[3], sums[5]))));
    if (h->canonical) {
        e0 = jolt::add(e0, jolt::mul(g3, sums[6]));
        e2 = jolt::add(e2, jolt::mul(g3, sums[7]));
    }
    fr_to_abi(&evals_out[0], e0);
    fr_to_abi(&evals_out[1], jolt::sub(fr_from_abi(previous_claim), e0));
    fr_to_abi(&evals_out[2], e2);
    return JOLT_OK;
}

// bind: phase_done = 1 when this was the phase's 8th bind (the eq table of the phase and the new checkpoints are then in place)
extern "C" int32_t jolt_host_read_raf_address_bind(jolt_read_raf_address* h, const jolt_fr_t* challenge, int32_t* phase_done) {
    if (!h || !challenge || !h->phase_open) return JOLT_ERR_INVALID_ARG;
    const Fr r = fr_from_abi(challenge);
    if (!fr_is_canonical(r)) return JOLT_ERR_INVALID_ARG;
    const size_t half = (kChunkSize >> h->phase_challenges.size()) / 2;
    for (auto& t : h->prefix_tables) bind_high_to_low(t, half, r);
    for (auto& polys : h->suffix_tables)
        for (auto& q : polys) bind_high_to_low(q, half, r);
    for (RafDecomposition* d : {&h->left, &h->right, &h->identity, &h->upper}) {
        if (d == &h->upper && !h->canonical) continue;
        bind_high_to_low(d->prefix, half, r);
        bind_high_to_low(d->q_shift, half, r);
        bind_high_to_low(d->q_value, half, r);
    }
    h->phase_challenges.push_back(r);
    const bool done = h->phase_challenges.size() == kChunkLen;
    if (done) {
        h->v_tables.push_back(eq_table(h->phase_challenges));
        for (size_t i = 0; i < h->prefix_indices.size(); ++i) h->checkpoints[h->prefix_indices[i]] = h->prefix_tables[i][0];
        h->left.checkpoint = h->left.prefix[0];
        h->right.checkpoint = h->right.prefix[0];
        h->identity.checkpoint = h->identity.prefix[0];
        if (h->canonical) h->upper.checkpoint = h->upper.prefix[0];
        h->phase += 1;
        h->phase_open = false;
    }
    if (phase_done) *phase_done = done ? 1 : 0;
    return JOLT_OK;
}
extern "C" int32_t jolt_host_read_raf_address_v_table(const jolt_read_raf_address* h, uint32_t phase, jolt_fr_t* out) {
    if (!h || !out || phase >= h->v_tables.size()) return JOLT_ERR_INVALID_ARG;
    for (uint32_t x = 0; x < kChunkSize; ++x) fr_to_abi(&out[x], h->v_tables[phase][x]);
    return JOLT_OK;
}
// init_cycle_rounds (:1140-1160): every table's value at r_address (its combine over the final checkpoints and the suffixes of the empty string), and the
// gamma-combined operand values the two RAF branches add
extern "C" int32_t jolt_host_read_raf_address_finish(const jolt_read_raf_address* h, jolt_fr_t* table_values, jolt_fr_t* raf_interleaved, jolt_fr_t* raf_identity) {
    if (!h || !table_values || !raf_interleaved || !raf_identity || h->phase != (uint32_t)kLogK / kChunkLen) return JOLT_ERR_INVALID_ARG;
    for (int t = 0; t < kNumTables; ++t) {
        const TableDesc& d = table_descs()[t];
        Fr s[5];
        for (uint32_t k = 0; k < d.n_suffixes; ++k) s[k] = jolt::fr_from_u64(jolt::suffix_mle(d.suffixes[k], 0, 0, 0));
        fr_to_abi(&table_values[t], table_combine(d, h->checkpoints, s));
    }
    const Fr g = h->gamma, g2 = jolt::mul(g, g);
    fr_to_abi(raf_interleaved, jolt::add(jolt::mul(g, h->left.checkpoint), jolt::mul(g2, h->right.checkpoint)));
    Fr id = jolt::mul(g2, h->identity.checkpoint);
    if (h->canonical) id = jolt::add(id, jolt::mul(jolt::mul(g2, g), h->upper.checkpoint));
    fr_to_abi(raf_identity, id);
    return JOLT_OK;
}
