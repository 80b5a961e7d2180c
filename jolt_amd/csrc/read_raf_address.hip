// jolt_amd/csrc/read_raf_address.hip -- the 128 address rounds of instruction read+RAF checking (stage 5), host side of the device scans.
//
// Replaces, together with read_raf.hip, the address half of OptimizedInstructionReadRafKernel (crates/jolt-kernels/src/optimized/instruction_read_raf.rs):
//   init_phase (:747-900)        the T-scale sums come from jolt_read_raf_phase_scan (device); here the 256-entry prefix polynomials are built from the
//                                checkpoints (lookup_tables.hpp) and the four RAF decompositions (left / right operand, identity, upper-all-ones) assembled
//   address_message (:973-1050)  s(0), s(2) over the live half of the chunk domain, s(1) = previous_claim - s(0)
//   bind (:1235-1282)            HighToLow bind of every 256-entry polynomial; after 8 binds the phase's eq table and the new checkpoints
//   init_cycle_rounds (:1140-1160)  table values, gamma-combined operand values for jolt_read_raf_cycle_tables
// Per proof this is 128 rounds of O(256 x present tables) field work -- no T-sized data -- which is why it stays on the host (the reference keeps
// it in rayon tasks of 8 entries); the device part of a phase is one scan launch + one condensation.  HOST ONLY: nothing here touches the device; the file keeps the
// .hip suffix because it shares field.hip.h with the kernels (one definition of the field arithmetic for both sides; hipcc emits no device code for it).
//
// Cost shape (all 42 tables present): per phase ~30 prefix + 94 suffix + 12 RAF polynomials of 256 entries; a round is ~60 dot products over the live half
// (the bilinear terms of the tables' `combine`, summed term by term instead of table by table: a table's value is never formed per entry) and ~136 binds.
// 64-bit-limb Montgomery arithmetic (x86 has the 64 x 64 -> 128 multiplier), one reduction per dot product, and SHARDS: the present tables are dealt out to a
// handful of worker threads, each owning its tables' suffix polynomials, private copies of the prefix polynomials they read and their terms, so that a round
// is ONE hand-off (bind with the previous challenge + extensions + partial sums) instead of three.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <future>
#include <mutex>
#include <new>
#include <pthread.h>
#include <thread>
#include <vector>

#include "ctx.hpp"
#include "lookup_tables.hpp"

using jolt::Fr;
using namespace jolt_lookup;


namespace {

constexpr uint32_t kChunkLen = 8, kChunkSize = 1u << kChunkLen, kPhases = (uint32_t)kLogK / kChunkLen;
typedef unsigned __int128 u128;

// ---- BN254 Fr on four 64-bit limbs (Montgomery form, canonical; the bytes of jolt_fr_t / jolt::Fr) ----
struct F { uint64_t l[4]; };
constexpr uint64_t kP[4] = {0x43E1F593F0000001ull, 0x2833E84879B97091ull, 0xB85045B68181585Dull, 0x30644E72E131A029ull};
constexpr uint64_t kNegInv = 0xC2E1F593EFFFFFFFull;  // -p^-1 mod 2^64
inline F from_fr(const Fr& v) { F r; std::memcpy(&r, &v, sizeof(F)); return r; }
inline Fr to_fr(const F& v) { Fr r; std::memcpy(&r, &v, sizeof(F)); return r; }
inline F f_zero() { return F{{0, 0, 0, 0}}; }
inline bool geq_p(const uint64_t t[4]) {
    for (int i = 3; i >= 0; --i) {
        if (t[i] != kP[i]) return t[i] > kP[i];
    }
    return true;
}
inline void sub_p_inplace(uint64_t t[4]) {
    u128 borrow = 0;
    for (int i = 0; i < 4; ++i) {
        const u128 d = (u128)t[i] - kP[i] - (uint64_t)borrow;
        t[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
}
inline F f_add(const F& a, const F& b) {
    F r;
    u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (c || geq_p(r.l)) sub_p_inplace(r.l);  // p < 2^254: the carry never survives the subtraction
    return r;
}
inline F f_sub(const F& a, const F& b) {
    F r;
    u128 borrow = 0;
    for (int i = 0; i < 4; ++i) {
        const u128 d = (u128)a.l[i] - b.l[i] - (uint64_t)borrow;
        r.l[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
    if (borrow) {
        u128 c = 0;
        for (int i = 0; i < 4; ++i) { c += (u128)r.l[i] + kP[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    }
    return r;
}
inline F f_mul(const F& a, const F& b) {  // CIOS, the product of jolt::host_mul64 without the 32-bit limb packing
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * kNegInv;
        c = ((u128)m * kP[0] + t[0]) >> 64;
        for (int j = 1; j < 4; ++j) { c += (u128)m * kP[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    F r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq_p(r.l)) sub_p_inplace(r.l);
    return r;
}
inline F f_ext2(const F& lo, const F& hi) { return f_sub(f_add(hi, hi), lo); }  // the line through (0, lo), (1, hi) at 2

// a * r for a challenge r of the reference's 125-bit shape: Montgomery limbs (0, 0, r2, r3) (from_challenge_bytes, crates/jolt-field/src/bn254/mod.rs:171-184).
// With r driving the outer loop the first two CIOS steps add nothing to a zero accumulator: half the work of f_mul.
inline F f_mul_challenge(const F& a, const F& r) {
    if (r.l[0] | r.l[1]) return f_mul(a, r);
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 2; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)a.l[j] * r.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * kNegInv;
        c = ((u128)m * kP[0] + t[0]) >> 64;
        for (int j = 1; j < 4; ++j) { c += (u128)m * kP[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    F out = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq_p(out.l)) sub_p_inplace(out.l);
    return out;
}

// sum_k a_k b_k with ONE Montgomery reduction: the 512-bit products are summed in nine limbs (at most 2^7 products of canonical operands: < 2^515), four
// reduction steps divide by 2^256, and what is left (< 33 p, five limbs) comes down by conditional subtractions of 32p, 16p, .. p.
struct DotAcc {
    uint64_t w[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    void fma(const F& a, const F& b) {
        uint64_t carry_top = 0;
        for (int i = 0; i < 4; ++i) {
            u128 c = 0;
            for (int j = 0; j < 4; ++j) { c += (u128)a.l[j] * b.l[i] + w[i + j]; w[i + j] = (uint64_t)c; c >>= 64; }
            for (int k = i + 4; k < 9 && c; ++k) { c += w[k]; w[k] = (uint64_t)c; c >>= 64; }
            carry_top |= (uint64_t)c;
        }
        (void)carry_top;  // cannot happen within the stated bound
    }
    F reduce() {
        for (int i = 0; i < 4; ++i) {
            const uint64_t m = w[i] * kNegInv;
            u128 c = 0;
            for (int j = 0; j < 4; ++j) { c += (u128)m * kP[j] + w[i + j]; w[i + j] = (uint64_t)c; c >>= 64; }
            for (int k = i + 4; k < 9 && c; ++k) { c += w[k]; w[k] = (uint64_t)c; c >>= 64; }
        }
        uint64_t v[5] = {w[4], w[5], w[6], w[7], w[8]};
        for (int shift = 5; shift >= 0; --shift) {  // v -= (p << shift) while that keeps v non-negative
            uint64_t m[5];
            m[0] = kP[0] << shift;
            for (int i = 1; i < 4; ++i) m[i] = (kP[i] << shift) | (shift ? kP[i - 1] >> (64 - shift) : 0);
            m[4] = shift ? kP[3] >> (64 - shift) : 0;
            bool ge = true;
            for (int i = 4; i >= 0; --i) {
                if (v[i] != m[i]) { ge = v[i] > m[i]; break; }
            }
            if (!ge) continue;
            u128 borrow = 0;
            for (int i = 0; i < 5; ++i) {
                const u128 d = (u128)v[i] - m[i] - (uint64_t)borrow;
                v[i] = (uint64_t)d;
                borrow = (d >> 64) & 1;
            }
        }
        return F{{v[0], v[1], v[2], v[3]}};
    }
};

// ---- worker threads (JOLT_HOST_THREADS > 1): one hand-off per round.  Workers spin for the next hand-off for a few milliseconds and block on a condition
// variable after that; the caller does the same while it waits.
class Pool {
   public:
    // Never destroyed (the workers die with the process: no join against sleeping threads at exit) and fork-aware: a child of fork() has no workers, only their
    // std::thread husks, so it runs everything on the calling thread.
    static Pool& get() {
        static Pool* p = [] {
            Pool* made = new Pool;
            pthread_atfork(nullptr, nullptr, [] { forked().store(true, std::memory_order_relaxed); });
            return made;
        }();
        return *p;
    }
    unsigned size() const { return (unsigned)workers_.size() + 1; }  // shards are dealt for this many threads whoever ends up running them
    // f(tid) for tid = 0 .. size() - 1; tid 0 runs on the caller
    void run(const std::function<void(unsigned)>& f) {
        if (workers_.empty()) { f(0); return; }
        if (forked().load(std::memory_order_relaxed)) {
            for (unsigned t = 0; t < size(); ++t) f(t);
            return;
        }
        std::lock_guard<std::mutex> one_job(run_m_);  // the pool is shared by every handle of the process: two provers on two threads take turns
        {
            std::lock_guard<std::mutex> g(m_);
            job_ = &f;
            pending_.store((unsigned)workers_.size(), std::memory_order_relaxed);
            generation_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        f(0);
        for (unsigned spins = 0; pending_.load(std::memory_order_acquire); ++spins) {
            if (spins < kSpins) { __builtin_ia32_pause(); continue; }
            std::unique_lock<std::mutex> g(m_);
            done_cv_.wait(g, [&] { return pending_.load(std::memory_order_acquire) == 0; });
        }
    }

   private:
    static std::atomic<bool>& forked() { static std::atomic<bool> f{false}; return f; }
    static constexpr unsigned kSpins = 200000;  // some milliseconds: the device scan between two phases included
    Pool() {
        // JOLT_HOST_THREADS, else 8 on a host with cores to spare (>= 32 hardware threads) and 1 otherwise.  The hand-offs are only cheap while the workers SPIN
        // between them, and spinning workers are only harmless next to idle cores.  Measured, all 42 tables present, per proof: MI355X host (2 x EPYC 9575F)
        // 17.2 ms with one thread, 8.7 / 4.4 / 3.7 ms with 4 / 8 / 16; a shared 8-CPU container 29 ms with one thread, 15 ms with 4, but 600 ms with 2 while
        // both threads sat on one core.
        const unsigned hw = std::thread::hardware_concurrency();
        long n = hw >= 32 ? 8 : 1;
        if (const char* e = std::getenv("JOLT_HOST_THREADS")) n = std::atol(e);
        const long cap = hw ? (long)hw : 64;  // hardware_concurrency() may report 0: still never more than a sane number of workers
        if (n < 1) n = 1;
        if (n > cap) n = cap;
        // thread creation can fail (std::system_error) and this constructor runs under an extern "C" entry point: keep the workers that did start
        try {
            workers_.reserve((size_t)n);
            for (long t = 1; t < n; ++t) workers_.emplace_back([this, t] { loop((unsigned)t); });
        } catch (...) {
        }
    }
    ~Pool() {
        { std::lock_guard<std::mutex> g(m_); stop_.store(true, std::memory_order_release); }
        generation_.fetch_add(1, std::memory_order_release);
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
    void loop(unsigned tid) {
        uint64_t seen = 0;
        for (;;) {
            for (unsigned spins = 0; generation_.load(std::memory_order_acquire) == seen; ++spins) {
                if (spins < kSpins) { __builtin_ia32_pause(); continue; }
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return generation_.load(std::memory_order_acquire) != seen || stop_.load(std::memory_order_acquire); });
            }
            if (stop_.load(std::memory_order_acquire)) return;
            seen = generation_.load(std::memory_order_acquire);
            (*job_)(tid);
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> g(m_);
                done_cv_.notify_one();
            }
        }
    }
    std::vector<std::thread> workers_;
    const std::function<void(unsigned)>* job_ = nullptr;
    std::atomic<uint64_t> generation_{0};
    std::atomic<unsigned> pending_{0};
    std::mutex m_, run_m_;
    std::condition_variable cv_, done_cv_;
    std::atomic<bool> stop_{false};
};

typedef F Poly[kChunkSize];  // a polynomial of the chunk; the first 256 >> (rounds bound) entries are live

// EqPolynomial::evals, big-endian (crates/jolt-poly/src/eq.rs:299-315)
void eq_table(const F* point, size_t n, F* e) {
    e[0] = from_fr(Fr::one());
    size_t size = 1;
    for (size_t k = 0; k < n; ++k) {
        for (size_t i = size; i-- > 0;) {
            const F hi = f_mul(e[i], point[k]);
            e[2 * i + 1] = hi;
            e[2 * i] = f_sub(e[i], hi);
        }
        size *= 2;
    }
}

struct TermRef { int8_t coef; int32_t prefix_poly /* shard-local, -1: none */; uint32_t suffix_poly /* shard-local */; };

// What one thread owns: some of the present tables -- their suffix polynomials, their terms, a private copy of every prefix polynomial they read -- and
// possibly some of the four RAF decompositions (prefix * q_shift + q_value, instruction_read_raf.rs:387-445).
struct Shard {
    std::vector<uint8_t> prefixes;        // prefix ids materialised here
    std::vector<uint32_t> suffix_source;  // per suffix polynomial: its row in the scan's layout
    std::vector<TermRef> terms;
    std::vector<uint8_t> raf;             // 0 = left, 1 = right, 2 = identity, 3 = upper_all_ones
    std::vector<F> polys, ext;            // [prefixes | suffixes | 3 per RAF decomposition][256]; ext = the top variable at 2 over the live half
    F partial[3];
    size_t n_polys() const { return prefixes.size() + suffix_source.size() + 3 * raf.size(); }
    F* prefix_poly(size_t i) { return polys.data() + i * kChunkSize; }
    F* suffix_poly(size_t i) { return polys.data() + (prefixes.size() + i) * kChunkSize; }
    F* raf_poly(size_t i, int which) { return polys.data() + (prefixes.size() + suffix_source.size() + 3 * i + which) * kChunkSize; }
    F* ext_of(const F* poly) { return ext.data() + (poly - polys.data()); }
};

}  // namespace

struct jolt_read_raf_address {
    F gamma, gamma2, gamma3, c_ones64, c_pow64, c_mask32;
    bool canonical;
    std::vector<uint8_t> present;         // table ids with at least one row
    std::vector<uint8_t> prefix_indices;  // prefixes those tables read, ascending
    Fr checkpoints[kNumPrefixes];
    F raf_checkpoint[4];
    std::vector<Shard> shards;
    F phase_challenges[kChunkLen];
    uint32_t bound = 0;       // binds done in the open phase
    std::vector<F> v_tables;  // completed phases' eq tables, [phase][256]
    uint32_t phase = 0;
    bool phase_open = false;
    // inputs of the open phase, read by the shards in their first step
    const jolt_fr_t *raf_sums = nullptr, *suffix_sums = nullptr;
    // The prefix polynomials of the NEXT phase depend on the checkpoints only, not on the scan: they are built on a background thread from the moment a phase
    // closes, i.e. while the caller condenses and scans on the device.
    std::future<void> prefixes_ready;
    ~jolt_read_raf_address() { if (prefixes_ready.valid()) prefixes_ready.wait(); }
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


// offsets of a table's suffix accumulators in the flattened layout jolt_read_raf_phase_scan writes for n_tables = 42 (LookupTableKind order), and the kinds
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

namespace { void build_prefixes(jolt_read_raf_address* h); void start_prefix_build(jolt_read_raf_address* h); }

extern "C" int32_t jolt_host_read_raf_address_create(const jolt_fr_t* gamma, const uint8_t* table_present, int32_t canonical, jolt_read_raf_address** out) {
    if (!gamma || !table_present || !out) return JOLT_ERR_INVALID_ARG;
    const Fr g = fr_from_abi(gamma);
    if (!fr_is_canonical(g)) return JOLT_ERR_INVALID_ARG;
    auto* h = new (std::nothrow) jolt_read_raf_address();
    if (!h) return JOLT_ERR_OOM;
    try {  // vector growth, the pool's first use (thread creation): nothing may unwind through the C ABI
    h->gamma = from_fr(g);
    h->gamma2 = f_mul(h->gamma, h->gamma);
    h->gamma3 = f_mul(h->gamma2, h->gamma);
    h->c_ones64 = from_fr(jolt::fr_from_u64(~0ull));
    h->c_pow64 = from_fr(fr_pow2(64));
    h->c_mask32 = from_fr(jolt::fr_from_u64(0xFFFFFFFFull));
    h->canonical = canonical != 0;
    bool reads[kNumPrefixes] = {};
    for (int t = 0; t < kNumTables; ++t) {
        if (!table_present[t]) continue;
        const TableDesc& d = table_descs()[t];
        h->present.push_back((uint8_t)t);
        for (uint32_t k = 0; k < d.n_prefixes; ++k) reads[d.prefixes[k]] = true;
    }
    for (int p = 0; p < kNumPrefixes; ++p) {
        h->checkpoints[p] = prefix_default_checkpoint(p);
        if (reads[p]) h->prefix_indices.push_back((uint8_t)p);
    }
    for (int q = 0; q < 3; ++q) h->raf_checkpoint[q] = f_zero();
    h->raf_checkpoint[3] = from_fr(Fr::one());  // an AND over address bits: the empty product (:417-423)
    // deal the work out: the RAF decompositions first (one shard each, from the last shard down), then the tables, heaviest first, each to the lightest shard
    const unsigned n_shards = Pool::get().size();
    h->shards.resize(n_shards);
    std::vector<uint32_t> load(n_shards, 0);
    const int n_raf = h->canonical ? 4 : 3;
    for (int q = 0; q < n_raf; ++q) {
        const unsigned s = (n_shards - 1 - (unsigned)q % n_shards);
        h->shards[s].raf.push_back((uint8_t)q);
        load[s] += 5;
    }
    uint32_t layout[kNumTables + 1];
    layout[0] = 0;
    for (int t = 0; t < kNumTables; ++t) layout[t + 1] = layout[t] + table_descs()[t].n_suffixes;
    auto cost = [](const TableDesc& d) { return 2u * d.n_terms + d.n_suffixes + d.n_prefixes; };
    std::vector<uint8_t> order = h->present;
    std::stable_sort(order.begin(), order.end(), [&](uint8_t a, uint8_t b) { return cost(table_descs()[a]) > cost(table_descs()[b]); });
    for (uint8_t t : order) {
        const TableDesc& d = table_descs()[t];
        unsigned s = 0;
        for (unsigned k = 1; k < n_shards; ++k)
            if (load[k] < load[s]) s = k;
        load[s] += cost(d);
        Shard& sh = h->shards[s];
        const uint32_t base = (uint32_t)sh.suffix_source.size();
        for (uint32_t k = 0; k < d.n_suffixes; ++k) sh.suffix_source.push_back(layout[t] + k);
        for (uint32_t k = 0; k < d.n_terms; ++k) {
            const Term& term = d.terms[k];
            if (term.suffix < 0) { delete h; return JOLT_ERR_UNSUPPORTED; }  // every term carries the rows' mass through a suffix
            int32_t slot = -1;
            if (term.prefix >= 0) {
                for (size_t i = 0; i < sh.prefixes.size(); ++i)
                    if (sh.prefixes[i] == (uint8_t)term.prefix) slot = (int32_t)i;
                if (slot < 0) { slot = (int32_t)sh.prefixes.size(); sh.prefixes.push_back((uint8_t)term.prefix); }
            }
            sh.terms.push_back(TermRef{term.coef, slot, base + (uint32_t)term.suffix});
        }
        // prefixes a table lists only because another of its prefixes reads their checkpoint (Eq under LessThan, ...) must be bound too: materialise all it lists
        for (uint32_t k = 0; k < d.n_prefixes; ++k) {
            bool have = false;
            for (uint8_t p : sh.prefixes) have |= p == d.prefixes[k];
            if (!have) sh.prefixes.push_back(d.prefixes[k]);
        }
    }
    for (Shard& sh : h->shards) {
        sh.polys.resize(sh.n_polys() * kChunkSize);
        sh.ext.resize(sh.polys.size());
    }
    start_prefix_build(h);
    } catch (...) {
        delete h;
        return JOLT_ERR_OOM;
    }
    *out = h;
    return JOLT_OK;
}
extern "C" int32_t jolt_host_read_raf_address_destroy(jolt_read_raf_address* h) {
    delete h;
    return JOLT_OK;
}

namespace {

// The first step of a phase on one shard: its polynomials from the scan's sums and the checkpoints (init_phase :814-898)
void build_prefixes(jolt_read_raf_address* h) {  // init_phase :878-897; a prefix shared by several shards is evaluated once and copied
    const uint32_t suffix_len = kLogK - (h->phase + 1) * kChunkLen;
    F* built[kNumPrefixes] = {};
    for (Shard& sh : h->shards)
        for (size_t i = 0; i < sh.prefixes.size(); ++i) {
            F* table = sh.prefix_poly(i);
            const uint8_t p = sh.prefixes[i];
            if (built[p]) { std::memcpy(table, built[p], sizeof(Poly)); continue; }
            for (uint32_t x = 0; x < kChunkSize; ++x) table[x] = from_fr(prefix_evaluate(p, h->checkpoints, x, kChunkLen, suffix_len));
            built[p] = table;
        }
}
// in the background when a thread can be had (nothing may unwind through the C ABI); init_phase builds them itself otherwise
void start_prefix_build(jolt_read_raf_address* h) {
    try {
        h->prefixes_ready = std::async(std::launch::async, [h] { build_prefixes(h); });
    } catch (...) {
        h->prefixes_ready = std::future<void>();
    }
}
void shard_init(jolt_read_raf_address* h, Shard& sh) {
    const uint32_t suffix_len = kLogK - (h->phase + 1) * kChunkLen;
    for (size_t i = 0; i < sh.suffix_source.size(); ++i) std::memcpy(sh.suffix_poly(i), &h->suffix_sums[(size_t)sh.suffix_source[i] * kChunkSize], sizeof(Poly));
    for (size_t i = 0; i < sh.raf.size(); ++i) {
        const uint8_t q = sh.raf[i];
        F *prefix = sh.raf_poly(i, 0), *q_shift = sh.raf_poly(i, 1), *q_value = sh.raf_poly(i, 2);
        auto column = [&](uint32_t c, F* out) { std::memcpy(out, &h->raf_sums[(size_t)c * kChunkSize], sizeof(Poly)); };
        if (q == 3) {  // the chunk's share of the upper word must be all ones (:857-876)
            const uint32_t done = h->phase * kChunkLen, word = (uint32_t)kLogK / 2;
            const uint32_t upper_bits = word > done ? (word - done < kChunkLen ? word - done : kChunkLen) : 0;
            for (uint32_t x = 0; x < kChunkSize; ++x) {
                prefix[x] = (upper_bits == 0 || (x >> (kChunkLen - upper_bits)) == (1u << upper_bits) - 1) ? h->raf_checkpoint[3] : f_zero();
                q_value[x] = f_zero();
            }
            column(5, q_shift);
            continue;
        }
        // operand prefixes: the bound part moves up by the chunk's share of bits, the chunk's own bits are added (:826-842); shift sums scaled (:814-823)
        const bool full = q == 2;
        const F scale = from_fr(fr_pow2(full ? suffix_len : suffix_len / 2));
        const F base = f_mul(h->raf_checkpoint[q], from_fr(fr_pow2(full ? kChunkLen : kChunkLen / 2)));
        column(full ? 4 : 3, q_shift);
        column(q, q_value);
        F small[kChunkSize];
        for (uint32_t x = 0; x < (full ? kChunkSize : 1u << (kChunkLen / 2)); ++x) small[x] = from_fr(jolt::fr_from_u64(x));
        for (uint32_t x = 0; x < kChunkSize; ++x) {
            const Chunk c = make_chunk(x, kChunkLen, suffix_len);
            prefix[x] = f_add(base, small[q == 0 ? c.x : (q == 1 ? c.y : x)]);
            q_shift[x] = f_mul(q_shift[x], scale);
        }
    }
}

void shard_bind(Shard& sh, size_t half, const F& r) {
    for (size_t i = 0; i < sh.n_polys(); ++i) {
        F* t = sh.polys.data() + i * kChunkSize;
        for (size_t b = 0; b < half; ++b) t[b] = f_add(t[b], f_mul_challenge(f_sub(t[b + half], t[b]), r));
    }
}
void shard_extend(Shard& sh, size_t half) {
    for (size_t i = 0; i < sh.n_polys(); ++i) {
        const F* t = sh.polys.data() + i * kChunkSize;
        F* e = sh.ext.data() + i * kChunkSize;
        for (size_t b = 0; b < half; ++b) e[b] = f_ext2(t[b], t[b + half]);
    }
}

// the shard's share of s(0), s(2) (and s(1) when the caller has no running claim): one dot product per bilinear term and point
void shard_message(const jolt_read_raf_address* h, Shard& sh, size_t half, bool with_one) {
    F e[3] = {f_zero(), f_zero(), f_zero()};
    for (const TermRef& term : sh.terms) {  // sum_b P(c, b) Q(c, b) of one term of one table's combine
        const F* q = sh.suffix_poly(term.suffix_poly);
        F acc[3] = {f_zero(), f_zero(), f_zero()};
        if (term.prefix_poly < 0) {
            for (size_t b = 0; b < half; ++b) {
                acc[0] = f_add(acc[0], q[b]);
                acc[1] = f_add(acc[1], q[b + half]);
            }
            acc[2] = f_sub(f_add(acc[1], acc[1]), acc[0]);  // linear in c
        } else {
            const F *p = sh.prefix_poly((size_t)term.prefix_poly), *p2 = sh.ext_of(p), *q2 = sh.ext_of(q);
            DotAcc a0, a1, a2;
            for (size_t b = 0; b < half; ++b) {
                a0.fma(p[b], q[b]);
                a2.fma(p2[b], q2[b]);
                if (with_one) a1.fma(p[b + half], q[b + half]);
            }
            acc[0] = a0.reduce();
            acc[2] = a2.reduce();
            if (with_one) acc[1] = a1.reduce();
        }
        for (int c = 0; c < 3; ++c) {
            if (c == 1 && !with_one) continue;
            switch (term.coef) {
                case kPlus: e[c] = f_add(e[c], acc[c]); break;
                case kMinus: e[c] = f_sub(e[c], acc[c]); break;
                case kOnes64: e[c] = f_add(e[c], f_mul(acc[c], h->c_ones64)); break;
                case kPow64: e[c] = f_add(e[c], f_mul(acc[c], h->c_pow64)); break;
                default: e[c] = f_add(e[c], f_mul(acc[c], h->c_mask32)); break;
            }
        }
    }
    const F weight[4] = {h->gamma, h->gamma2, h->gamma2, h->gamma3};  // gamma * left + gamma^2 (right + identity) (+ gamma^3 upper)
    for (size_t i = 0; i < sh.raf.size(); ++i) {
        const F *prefix = sh.raf_poly(i, 0), *q_shift = sh.raf_poly(i, 1), *q_value = sh.raf_poly(i, 2);
        const F *prefix2 = sh.ext_of(prefix), *q_shift2 = sh.ext_of(q_shift), *q_value2 = sh.ext_of(q_value);
        DotAcc a0, a1, a2;
        F v0 = f_zero(), v1 = f_zero(), v2 = f_zero();
        for (size_t b = 0; b < half; ++b) {
            a0.fma(prefix[b], q_shift[b]);
            a2.fma(prefix2[b], q_shift2[b]);
            v0 = f_add(v0, q_value[b]);
            v2 = f_add(v2, q_value2[b]);
            if (with_one) {
                a1.fma(prefix[b + half], q_shift[b + half]);
                v1 = f_add(v1, q_value[b + half]);
            }
        }
        const F w = weight[sh.raf[i]];
        e[0] = f_add(e[0], f_mul(f_add(a0.reduce(), v0), w));
        e[2] = f_add(e[2], f_mul(f_add(a2.reduce(), v2), w));
        if (with_one) e[1] = f_add(e[1], f_mul(f_add(a1.reduce(), v1), w));
    }
    for (int c = 0; c < 3; ++c) sh.partial[c] = e[c];
}

// One hand-off: [first step of the phase: build] [bind with r] extensions, partial sums.  Each shard touches only what it owns.
enum : unsigned { kStepInit = 1, kStepBind = 2, kStepMessage = 4, kStepWithOne = 8 };
void step_body(jolt_read_raf_address* h, unsigned what, const F* r, F out[3]) {
    const size_t live = kChunkSize >> h->bound;  // before the bind of this step
    const std::function<void(unsigned)> work = [&](unsigned tid) {
        if (tid >= h->shards.size()) return;
        Shard& sh = h->shards[tid];
        size_t now = live;
        if (what & kStepInit) shard_init(h, sh);
        if (what & kStepBind) { shard_bind(sh, live / 2, *r); now = live / 2; }
        if (what & kStepMessage) {
            shard_extend(sh, now / 2);
            shard_message(h, sh, now / 2, (what & kStepWithOne) != 0);
        }
    };
    if (live <= 8 && !(what & kStepInit)) {  // the last rounds of a phase are a handful of entries: not worth a hand-off
        for (unsigned t = 0; t < h->shards.size(); ++t) work(t);
    } else {
        Pool::get().run(work);
    }
    if (what & kStepBind) h->phase_challenges[h->bound++] = *r;
    if (what & kStepMessage)
        for (int c = 0; c < 3; ++c) {
            out[c] = f_zero();
            for (const Shard& sh : h->shards) out[c] = f_add(out[c], sh.partial[c]);
        }
}

// the std::function of a hand-off may allocate, the pool's first use creates threads: an exception becomes a status here, never unwinds through the C ABI
int32_t step(jolt_read_raf_address* h, unsigned what, const F* r, F out[3]) {
    try {
        step_body(h, what, r, out);
    } catch (...) {
        return JOLT_ERR_OOM;
    }
    return JOLT_OK;
}

// after the 8th bind: the phase's eq table and the new checkpoints (:1262-1275)
void close_phase_body(jolt_read_raf_address* h) {
    h->v_tables.resize((size_t)(h->phase + 1) * kChunkSize);
    eq_table(h->phase_challenges, kChunkLen, h->v_tables.data() + (size_t)h->phase * kChunkSize);
    for (Shard& sh : h->shards) {
        for (size_t i = 0; i < sh.prefixes.size(); ++i) h->checkpoints[sh.prefixes[i]] = to_fr(sh.prefix_poly(i)[0]);  // copies in other shards hold the same value
        for (size_t i = 0; i < sh.raf.size(); ++i) h->raf_checkpoint[sh.raf[i]] = sh.raf_poly(i, 0)[0];
    }
    h->phase += 1;
    h->phase_open = false;
    h->raf_sums = h->suffix_sums = nullptr;
    if (h->phase < kPhases) start_prefix_build(h);
}
int32_t close_phase(jolt_read_raf_address* h) {
    try {
        close_phase_body(h);
    } catch (...) {
        return JOLT_ERR_OOM;
    }
    return JOLT_OK;
}

}  // namespace

// init_phase: raf_sums[q * 256 + chunk], q = left, right, identity, shift_half, shift_full, upper_all_ones (raw, as jolt_read_raf_phase_scan returns them);
// suffix_sums[(offsets[t] + s) * 256 + chunk] in the layout of jolt_lookup_suffix_layout.  The sums are copied into the phase's polynomials here.
extern "C" int32_t jolt_host_read_raf_address_init_phase(jolt_read_raf_address* h, uint32_t phase, const jolt_fr_t* raf_sums, const jolt_fr_t* suffix_sums) {
    if (!h || !raf_sums || !suffix_sums || phase != h->phase || h->phase_open || phase >= kPhases) return JOLT_ERR_INVALID_ARG;
    try {
        if (h->prefixes_ready.valid()) h->prefixes_ready.get();
        else build_prefixes(h);
    } catch (...) {
        return JOLT_ERR_OOM;
    }
    h->raf_sums = raf_sums;
    h->suffix_sums = suffix_sums;
    h->bound = 0;
    h->phase_open = true;
    F unused[3];
    const int32_t st = step(h, kStepInit, nullptr, unused);
    h->raf_sums = h->suffix_sums = nullptr;
    return st;
}

// address_message: evals_out = s(0), s(1), s(2) (UnivariatePoly::from_evals order).  previous_claim == NULL: s(1) is summed from the tables instead of
// taken from the running claim -- in round 0 that makes s(0) + s(1) the relation's input claim.
extern "C" int32_t jolt_host_read_raf_address_message(jolt_read_raf_address* h, const jolt_fr_t* previous_claim, jolt_fr_t* evals_out) {
    if (!h || !evals_out || !h->phase_open) return JOLT_ERR_INVALID_ARG;
    F e[3];
    if (const int32_t st = step(h, kStepMessage | (previous_claim ? 0u : (unsigned)kStepWithOne), nullptr, e)) return st;
    if (previous_claim) {
        F claim;
        std::memcpy(&claim, previous_claim, sizeof(F));
        e[1] = f_sub(claim, e[0]);
    }
    std::memcpy(evals_out, e, sizeof(e));
    return JOLT_OK;
}

// bind: phase_done = 1 when this was the phase's 8th bind (the eq table of the phase and the new checkpoints are then in place)
extern "C" int32_t jolt_host_read_raf_address_bind(jolt_read_raf_address* h, const jolt_fr_t* challenge, int32_t* phase_done) {
    if (!h || !challenge || !h->phase_open) return JOLT_ERR_INVALID_ARG;
    const Fr r = fr_from_abi(challenge);
    if (!fr_is_canonical(r)) return JOLT_ERR_INVALID_ARG;
    const F rf = from_fr(r);
    F unused[3];
    if (const int32_t st = step(h, kStepBind, &rf, unused)) return st;
    const bool done = h->bound == kChunkLen;
    if (done)
        if (const int32_t st = close_phase(h)) return st;
    if (phase_done) *phase_done = done ? 1 : 0;
    return JOLT_OK;
}

// bind + the next message in ONE hand-off (the fused ProveRounds contract, prover.rs:45-51): for the binds that do not close the phase (the 1st .. 7th of a phase);
// the 8th goes through jolt_host_read_raf_address_bind, after which the next phase's scan sums are due (init_phase) before a message exists.
extern "C" int32_t jolt_host_read_raf_address_bind_message(jolt_read_raf_address* h, const jolt_fr_t* challenge, const jolt_fr_t* previous_claim, jolt_fr_t* evals_out) {
    if (!h || !challenge || !previous_claim || !evals_out || !h->phase_open || h->bound + 1 >= kChunkLen) return JOLT_ERR_INVALID_ARG;
    const Fr r = fr_from_abi(challenge);
    if (!fr_is_canonical(r)) return JOLT_ERR_INVALID_ARG;
    const F rf = from_fr(r);
    F e[3], claim;
    if (const int32_t st = step(h, kStepBind | kStepMessage, &rf, e)) return st;
    std::memcpy(&claim, previous_claim, sizeof(F));
    e[1] = f_sub(claim, e[0]);
    std::memcpy(evals_out, e, sizeof(e));
    return JOLT_OK;
}

// The 8 rounds of the open phase in one call: message -> UnivariatePoly::from_evals coefficients [c0, c1, c2] -> transcript -> bind.  The transcript is the
// caller's hook (jolt_round_transcript_fn: absorbs the coefficients, returns the challenge) or, with fn == NULL, the library's test transcript `test_transcript`
// (append the three coefficients, Transcript::challenge).  claim: in = the running claim before the phase, out = after it.  A round's bind shares its
// hand-off with the next round's sums.
extern "C" int32_t jolt_host_read_raf_address_prove_phase(jolt_read_raf_address* h, jolt_fr_t* claim, jolt_round_transcript_fn fn, void* user, jolt_host_transcript* test_transcript,
                                                          jolt_fr_t* coeffs_out /* 8 x 3 */, jolt_fr_t* challenges_out /* 8 */) {
    if (!h || !claim || (!fn && !test_transcript) || !h->phase_open || h->bound != 0) return JOLT_ERR_INVALID_ARG;
    static const F inv2 = from_fr(jolt::inv(jolt::fr_from_u64(2)));
    F running, rf = f_zero();
    std::memcpy(&running, claim, sizeof(F));
    for (uint32_t round = 0; round < kChunkLen; ++round) {
        F e[3];
        if (const int32_t st = step(h, round ? (kStepBind | kStepMessage) : (unsigned)kStepMessage, &rf, e)) return st;
        e[1] = f_sub(running, e[0]);
        // the quadratic through (0, e0), (1, e1), (2, e2): c2 = (e2 - 2 e1 + e0) / 2, c1 = e1 - e0 - c2
        F c[3];
        c[0] = e[0];
        c[2] = f_mul(f_add(f_sub(e[2], f_add(e[1], e[1])), e[0]), inv2);
        c[1] = f_sub(f_sub(e[1], e[0]), c[2]);
        jolt_fr_t coeffs[3], challenge;
        std::memcpy(coeffs, c, sizeof(c));
        if (fn) {
            const int32_t st = fn(user, coeffs, 3, &challenge);
            if (st != JOLT_OK) return st;
        } else {
            int32_t st = jolt_host_transcript_append_fr(test_transcript, coeffs, 3);
            if (st == JOLT_OK) st = jolt_host_transcript_challenge(test_transcript, 0, &challenge);
            if (st != JOLT_OK) return st;
        }
        const Fr r = fr_from_abi(&challenge);
        if (!fr_is_canonical(r)) return JOLT_ERR_INVALID_ARG;
        rf = from_fr(r);
        running = f_add(c[0], f_mul(rf, f_add(c[1], f_mul(rf, c[2]))));
        if (coeffs_out) std::memcpy(&coeffs_out[3 * round], coeffs, sizeof(coeffs));
        if (challenges_out) challenges_out[round] = challenge;
    }
    F unused[3];
    if (const int32_t st = step(h, kStepBind, &rf, unused)) return st;
    if (const int32_t st = close_phase(h)) return st;
    std::memcpy(claim, &running, sizeof(F));
    return JOLT_OK;
}

extern "C" int32_t jolt_host_read_raf_address_v_table(const jolt_read_raf_address* h, uint32_t phase, jolt_fr_t* out) {
    if (!h || !out || (size_t)(phase + 1) * kChunkSize > h->v_tables.size()) return JOLT_ERR_INVALID_ARG;
    std::memcpy(out, h->v_tables.data() + (size_t)phase * kChunkSize, sizeof(Poly));
    return JOLT_OK;
}
// init_cycle_rounds (:1140-1160): every table's value at r_address (its combine over the final checkpoints and the suffixes of the empty string), and the
// gamma-combined operand values the two RAF branches add
extern "C" int32_t jolt_host_read_raf_address_finish(const jolt_read_raf_address* h, jolt_fr_t* table_values, jolt_fr_t* raf_interleaved, jolt_fr_t* raf_identity) {
    if (!h || !table_values || !raf_interleaved || !raf_identity || h->phase != kPhases) return JOLT_ERR_INVALID_ARG;
    for (int t = 0; t < kNumTables; ++t) {
        const TableDesc& d = table_descs()[t];
        Fr s[5];
        for (uint32_t k = 0; k < d.n_suffixes; ++k) s[k] = jolt::fr_from_u64(jolt::suffix_mle(d.suffixes[k], 0, 0, 0));
        fr_to_abi(&table_values[t], table_combine(d, h->checkpoints, s));
    }
    const F interleaved = f_add(f_mul(h->gamma, h->raf_checkpoint[0]), f_mul(h->gamma2, h->raf_checkpoint[1]));
    F id = f_mul(h->gamma2, h->raf_checkpoint[2]);
    if (h->canonical) id = f_add(id, f_mul(h->gamma3, h->raf_checkpoint[3]));
    std::memcpy(raf_interleaved, &interleaved, sizeof(F));
    std::memcpy(raf_identity, &id, sizeof(F));
    return JOLT_OK;
}

