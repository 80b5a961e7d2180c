// jolt_amd/csrc/host_mirror.hip -- implementation of host_mirror.hpp and of the jolt_host_* exports (sumcheck side).
#include "host_mirror.hpp"

#include <algorithm>
#include <array>
#include <string>

#include "member.hpp"

using namespace jolt;

namespace jolt_host {

// ---- field helpers -------------------------------------------------------------------------------------------
Fr fr_mul_pow_2(Fr a, size_t k) {
    for (size_t i = 0; i < k; ++i) a = add(a, a);
    return a;
}
void fr_to_bytes_le(const Fr& a, uint8_t out[32]) {
    Fr c = from_mont(a);
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(c.l[i] >> (8 * j));
}
// crates/jolt-field/src/bn254/mod.rs:171-184,254: masked 125-bit value placed raw in the two high u64 limbs
Fr fr_from_challenge_bytes(const uint8_t* b, size_t n) {
    uint8_t buf[16] = {0};
    std::memcpy(buf, b, n < 16 ? n : 16);
    Fr r = Fr::zero();
    for (int i = 0; i < 4; ++i) {
        uint32_t w = 0;
        for (int j = 3; j >= 0; --j) w = (w << 8) | buf[4 * i + j];
        r.l[4 + i] = w;
    }
    r.l[7] &= 0x1FFFFFFFu;  // top 3 bits of the high u64 limb cleared
    return r;
}
// mod.rs:188-193: big-endian integer of the bytes, reduced mod r
Fr fr_from_scalar_challenge_bytes(const uint8_t* b, size_t n) {
    Fr acc = Fr::zero();
    Fr base = fr_from_u64(256);
    for (size_t i = 0; i < n; ++i) acc = add(mul(acc, base), fr_from_u64(b[i]));
    return acc;
}

// ---- UnivariatePoly ------------------------------------------------------------------------------------------
// 1/k for the small integers Newton interpolation divides by (computed once: a Fermat inversion costs ~400 host multiplies)
static const Fr& small_inverse(size_t k) {
    static std::vector<Fr> table = [] {
        std::vector<Fr> t(33, Fr::zero());
        for (size_t i = 1; i < t.size(); ++i) t[i] = inv(fr_from_u64(i));
        return t;
    }();
    return table[k];
}

UnivariatePoly UnivariatePoly::from_evals(const Fr* evals, size_t n) {
    // unique interpolant through (0,e0)..(n-1,e_{n-1}); the reference solves the Vandermonde system
    // (univariate.rs:198-202), here Newton's forward differences followed by basis expansion.
    std::vector<Fr> d(evals, evals + n);
    for (size_t k = 1; k < n; ++k) {
        Fr kinv = k < 33 ? small_inverse(k) : inv(fr_from_u64(k));
        for (size_t i = n - 1; i >= k; --i) d[i] = mul(sub(d[i], d[i - 1]), kinv);
    }
    std::vector<Fr> c(n, Fr::zero()), basis(1, Fr::one());
    for (size_t k = 0; k < n; ++k) {
        for (size_t i = 0; i < basis.size(); ++i) c[i] = add(c[i], mul(d[k], basis[i]));
        if (k + 1 < n) {
            Fr kf = fr_from_u64(k);
            std::vector<Fr> nb(basis.size() + 1, Fr::zero());
            for (size_t i = 0; i < basis.size(); ++i) {
                nb[i + 1] = add(nb[i + 1], basis[i]);
                nb[i] = sub(nb[i], mul(basis[i], kf));
            }
            basis.swap(nb);
        }
    }
    UnivariatePoly p;
    p.coefficients = std::move(c);
    return p;
}
Fr UnivariatePoly::evaluate(const Fr& x) const {
    Fr acc = Fr::zero();
    for (size_t i = coefficients.size(); i-- > 0;) acc = add(mul(acc, x), coefficients[i]);
    return acc;
}
size_t UnivariatePoly::degree() const { return coefficients.empty() ? 0 : coefficients.size() - 1; }

// split_eq.rs:383-417
Fr inverse_or_given(const Fr& x, const Fr* given) {
    if (given && mul(x, *given) == Fr::one()) return *given;
    return inv(x);
}

int32_t gruen_poly_deg_3(const Fr& current_scalar, const Fr& point_i, const Fr& q_constant, const Fr& q_quadratic, const Fr& s0_plus_s1,
                         UnivariatePoly* out, const Fr* inv_l1) {
    Fr eq1 = mul(current_scalar, point_i);
    Fr eq0 = sub(current_scalar, eq1);
    Fr eqm = sub(eq1, eq0);
    Fr eq2 = add(eq1, eqm);
    Fr eq3 = add(eq2, eqm);
    Fr cubic0 = mul(eq0, q_constant);
    Fr cubic1 = sub(s0_plus_s1, cubic0);
    if (eq1.is_zero()) return JOLT_ERR_NOT_INVERTIBLE;
    Fr quad1 = mul(cubic1, inverse_or_given(eq1, inv_l1));
    Fr e2 = add(q_quadratic, q_quadratic);
    Fr quad2 = add(sub(add(quad1, quad1), q_constant), e2);
    Fr quad3 = add(add(sub(add(quad2, quad1), q_constant), e2), e2);
    Fr evals[4] = {cubic0, cubic1, mul(eq2, quad2), mul(eq3, quad3)};
    *out = UnivariatePoly::from_evals(evals, 4);  // interpolate_over_integers: same unique cubic
    return JOLT_OK;
}

// s(t) = l(t) q(t) with l(0) = scalar (1 - w_i), l(1) = scalar w_i, from q(0), q(2), .., q(dq) and claim = s(0) + s(1):
// q(1) = (claim - l(0) q(0)) / l(1), interpolate q on {0..dq}, multiply by the linear factor.  Same polynomial as
// GruenSplitEqPolynomial::gruen_poly_from_evals (split_eq.rs:419-447) produces from its own evaluation set.
int32_t gruen_poly_from_q(const Fr& current_scalar, const Fr& point_i, const Fr* q_evals, size_t dq, const Fr& s0_plus_s1, UnivariatePoly* out,
                          const Fr* inv_l1) {
    Fr l1 = mul(current_scalar, point_i);
    Fr l0 = sub(current_scalar, l1);
    if (l1.is_zero()) return JOLT_ERR_NOT_INVERTIBLE;
    std::vector<Fr> q(dq + 1);
    q[0] = q_evals[0];
    q[1] = mul(sub(s0_plus_s1, mul(l0, q[0])), inverse_or_given(l1, inv_l1));
    for (size_t t = 2; t <= dq; ++t) q[t] = q_evals[t - 1];
    UnivariatePoly qp = UnivariatePoly::from_evals(q.data(), q.size());
    Fr lc1 = sub(l1, l0);
    std::vector<Fr> sc(dq + 2, Fr::zero());
    for (size_t k = 0; k <= dq; ++k) {
        sc[k] = add(sc[k], mul(qp.coefficients[k], l0));
        sc[k + 1] = add(sc[k + 1], mul(qp.coefficients[k], lc1));
    }
    out->coefficients = std::move(sc);
    return JOLT_OK;
}

// ---- transcript ----------------------------------------------------------------------------------------------
Fr Transcript::challenge() {
    uint8_t b[16];
    draw16(b);
    return fr_from_challenge_bytes(b, 16);
}
Fr Transcript::challenge_scalar() {
    uint8_t b[16];
    draw16(b);
    return fr_from_scalar_challenge_bytes(b, 16);
}
void Transcript::append_fr(const Fr& v) {  // legacy.rs:116-123: to_bytes_le, reversed
    uint8_t le[32], be[32];
    fr_to_bytes_le(v, le);
    std::reverse_copy(le, le + 32, be);
    append_bytes(be, 32);
}
namespace {
struct Word32 {
    uint8_t b[32] = {};
    void text(const char* label, size_t cap) { std::memcpy(b, label, std::min(std::strlen(label), cap)); }
    void tail_u64(uint64_t v) {
        for (int i = 0; i < 8; ++i) b[31 - i] = (uint8_t)(v >> (8 * i));
    }
};
}  // namespace
void Transcript::append_label(const char* label) {
    Word32 w;
    w.text(label, 32);
    append_bytes(w.b, 32);
}
void Transcript::append_label_with_count(const char* label, uint64_t count) {
    Word32 w;
    w.text(label, 24);
    w.tail_u64(count);
    append_bytes(w.b, 32);
}
void Transcript::append_u64_word(uint64_t v) {
    Word32 w;
    w.tail_u64(v);
    append_bytes(w.b, 32);
}
void Transcript::append_round_poly(const char* label, const Fr* coefficients, size_t n) {
    if (n == 0) return;  // round_proof.rs:135-137
    append_label_with_count(label, (uint64_t)(n - 1));
    append_fr(coefficients[0]);
    for (size_t k = 2; k < n; ++k) append_fr(coefficients[k]);
}

static inline uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
MockTranscript::MockTranscript(uint64_t label) {
    s[0] = 0x6a09e667f3bcc908ull ^ label;
    s[1] = 0xbb67ae8584caa73bull;
    s[2] = 0x3c6ef372fe94f82bull;
    s[3] = 0xa54ff53a5f1d36f1ull;
}
void MockTranscript::absorb_word(uint64_t w) {
    s[0] = mix64(s[0] ^ w);
    s[1] = mix64(s[1] + s[0]);
    s[2] ^= (s[1] << 23) | (s[1] >> 41);
    s[3] = mix64(s[3] ^ s[2] ^ w);
}
void MockTranscript::append_bytes(const uint8_t* b, size_t n) {
    absorb_word((uint64_t)n);
    for (size_t i = 0; i < n; i += 8) {
        uint64_t w = 0;
        for (size_t j = 0; j < 8 && i + j < n; ++j) w |= (uint64_t)b[i + j] << (8 * j);
        absorb_word(w);
    }
}
void MockTranscript::draw16(uint8_t out[16]) {
    absorb_word(0xC4A11E46Eull);
    uint64_t lo = mix64(s[0] ^ s[2]);
    uint64_t hi = mix64(s[1] ^ s[3]);
    absorb_word(lo ^ hi);
    for (int i = 0; i < 8; ++i) { out[i] = (uint8_t)(lo >> (8 * i)); out[8 + i] = (uint8_t)(hi >> (8 * i)); }
}
void MockTranscript::state(uint8_t out[32]) const { std::memcpy(out, s, 32); }

// ---- BLAKE2b (RFC 7693): eight-word chain value, 128-byte blocks, twelve rounds of the G mixing over the message schedule ----
namespace {
class Blake2b {
   public:
    explicit Blake2b(size_t digest_bytes) : out_(digest_bytes) {
        h_ = iv();
        h_[0] ^= 0x01010000ull | (uint64_t)digest_bytes;
    }
    void update(const uint8_t* p, size_t n) {
        for (size_t i = 0; i < n; ++i) {
            if (fill_ == block_.size()) flush(false);  // only once more input is known to follow
            block_[fill_++] = p[i];
        }
    }
    void finish(uint8_t* out) {
        std::fill(block_.begin() + fill_, block_.end(), (uint8_t)0);
        flush(true);
        for (size_t i = 0; i < out_; ++i) out[i] = (uint8_t)(h_[i >> 3] >> (8 * (i & 7)));
    }

   private:
    static std::array<uint64_t, 8> iv() {
        return {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    }
    void flush(bool last) {
        counter_ += fill_;  // messages here are far below 2^64 bytes: the high counter word stays zero
        std::array<uint64_t, 16> m{}, v{};
        for (size_t w = 0; w < 16; ++w)
            for (size_t k = 0; k < 8; ++k) m[w] |= (uint64_t)block_[8 * w + k] << (8 * k);
        const std::array<uint64_t, 8> c = iv();
        for (size_t i = 0; i < 8; ++i) { v[i] = h_[i]; v[8 + i] = c[i]; }
        v[12] ^= counter_;
        if (last) v[14] = ~v[14];
        auto rr = [](uint64_t x, unsigned k) { return (x >> k) | (x << (64 - k)); };
        auto g = [&](int a, int b, int cc, int d, uint64_t x, uint64_t y) {
            v[a] += v[b] + x; v[d] = rr(v[d] ^ v[a], 32); v[cc] += v[d]; v[b] = rr(v[b] ^ v[cc], 24);
            v[a] += v[b] + y; v[d] = rr(v[d] ^ v[a], 16); v[cc] += v[d]; v[b] = rr(v[b] ^ v[cc], 63);
        };
        static const uint8_t schedule[10][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
        for (int round = 0; round < 12; ++round) {
            const uint8_t* sg = schedule[round % 10];
            for (int col = 0; col < 4; ++col) g(col, 4 + col, 8 + col, 12 + col, m[sg[2 * col]], m[sg[2 * col + 1]]);
            for (int diag = 0; diag < 4; ++diag)
                g(diag, 4 + (diag + 1) % 4, 8 + (diag + 2) % 4, 12 + (diag + 3) % 4, m[sg[8 + 2 * diag]], m[sg[9 + 2 * diag]]);
        }
        for (size_t i = 0; i < 8; ++i) h_[i] ^= v[i] ^ v[8 + i];
        fill_ = 0;
    }
    std::array<uint64_t, 8> h_{};
    std::array<uint8_t, 128> block_{};
    size_t fill_ = 0, out_;
    uint64_t counter_ = 0;
};
}  // namespace
void blake2b_digest(const uint8_t* in, size_t n, size_t outlen, uint8_t* out) {
    Blake2b h(outlen);
    h.update(in, n);
    h.finish(out);
}

LegacyBlake2bTranscript::LegacyBlake2bTranscript(const uint8_t* label, size_t n) {  // digest.rs:153-174
    uint8_t padded[32] = {};
    std::memcpy(padded, label, std::min<size_t>(n, 32));
    blake2b_digest(padded, 32, 32, chain);
}
void LegacyBlake2bTranscript::step(const uint8_t* payload, size_t n) {  // hasher() + update_state (digest.rs:100-104,129-131)
    uint8_t round_word[32] = {};
    for (int i = 0; i < 4; ++i) round_word[31 - i] = (uint8_t)(n_rounds >> (8 * i));
    Blake2b h(32);
    h.update(chain, 32);
    h.update(round_word, 32);
    h.update(payload, n);
    h.finish(chain);
    ++n_rounds;
}
void LegacyBlake2bTranscript::append_bytes(const uint8_t* b, size_t n) { step(b, n); }
void LegacyBlake2bTranscript::draw16(uint8_t out[16]) {  // challenge_bytes of 16 <= 32 bytes: one digest, its first half (digest.rs:106-127)
    step(nullptr, 0);
    std::memcpy(out, chain, 16);
}
void LegacyBlake2bTranscript::state(uint8_t out[32]) const { std::memcpy(out, chain, 32); }

// ---- Keccak-f[1600] (FIPS 202): theta, rho + pi along the 24-step lane walk, chi, iota ----
void keccak_f1600(uint8_t bytes[200]) {
    static const uint64_t iota[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
                                      0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
                                      0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
                                      0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
                                      0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    static const int walk_lane[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    static const int walk_rot[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    uint64_t a[25];
    for (int i = 0; i < 25; ++i) {
        a[i] = 0;
        for (int k = 7; k >= 0; --k) a[i] = (a[i] << 8) | bytes[8 * i + k];
    }
    auto rl = [](uint64_t x, int k) { return (x << k) | (x >> (64 - k)); };
    for (int round = 0; round < 24; ++round) {
        uint64_t parity[5];
        for (int x = 0; x < 5; ++x) parity[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; ++x) {
            const uint64_t d = parity[(x + 4) % 5] ^ rl(parity[(x + 1) % 5], 1);
            for (int y = 0; y < 25; y += 5) a[y + x] ^= d;
        }
        uint64_t carry = a[1];
        for (int i = 0; i < 24; ++i) {
            const uint64_t next = a[walk_lane[i]];
            a[walk_lane[i]] = rl(carry, walk_rot[i]);
            carry = next;
        }
        for (int y = 0; y < 25; y += 5) {
            uint64_t row[5];
            for (int x = 0; x < 5; ++x) row[x] = a[y + x];
            for (int x = 0; x < 5; ++x) a[y + x] = row[x] ^ (~row[(x + 1) % 5] & row[(x + 2) % 5]);
        }
        a[0] ^= iota[round];
    }
    for (int i = 0; i < 25; ++i)
        for (int k = 0; k < 8; ++k) bytes[8 * i + k] = (uint8_t)(a[i] >> (8 * k));
}
// the duplex (spongefish DuplexSponge, overwrite mode): input REPLACES rate bytes; a full rate is permuted before more input; the first squeeze after an absorb permutes
void KeccakSpongeTranscript::absorb(const uint8_t* b, size_t n) {
    squeeze_pos = 136;
    size_t done = 0;
    while (done < n) {
        if (absorb_pos == 136) {
            keccak_f1600(lanes);
            absorb_pos = 0;
            continue;
        }
        const size_t take = std::min<size_t>(n - done, 136 - absorb_pos);
        std::memcpy(lanes + absorb_pos, b + done, take);
        absorb_pos += (unsigned)take;
        done += take;
    }
}
void KeccakSpongeTranscript::squeeze(uint8_t* out, size_t n) {
    size_t done = 0;
    while (done < n) {
        if (squeeze_pos == 136) {
            squeeze_pos = 0;
            absorb_pos = 0;
            keccak_f1600(lanes);
        }
        const size_t take = std::min<size_t>(n - done, 136 - squeeze_pos);
        std::memcpy(out + done, lanes + squeeze_pos, take);
        squeeze_pos += (unsigned)take;
        done += take;
    }
}
KeccakSpongeTranscript::KeccakSpongeTranscript(const uint8_t* label, size_t n) {  // SpongeTranscript::new (legacy.rs:254-268)
    std::memset(lanes, 0, sizeof lanes);
    uint8_t protocol_id[64] = {};  // PROTOCOL_ID (setup.rs:38-54)
    static const char id[] = "a16z/jolt-transcript/v1";
    std::memcpy(protocol_id, id, sizeof id - 1);
    absorb(protocol_id, 64);
    std::vector<uint8_t> session(8 + n);  // BytesMsg(label): u64 LE length, then the bytes (codec.rs:36-43)
    for (int i = 0; i < 8; ++i) session[i] = (uint8_t)((uint64_t)n >> (8 * i));
    std::memcpy(session.data() + 8, label, n);
    absorb(session.data(), session.size());
    absorb(nullptr, 0);  // EmptyInstance (setup.rs:59-66)
}
void KeccakSpongeTranscript::append_bytes(const uint8_t* b, size_t n) {  // legacy.rs:270-288: marker, u64 LE length, body -- one absorb
    std::vector<uint8_t> framed(9 + n);
    framed[0] = 0x9B;
    for (int i = 0; i < 8; ++i) framed[1 + i] = (uint8_t)((uint64_t)n >> (8 * i));
    if (n) std::memcpy(framed.data() + 9, b, n);
    absorb(framed.data(), framed.size());
}
void KeccakSpongeTranscript::draw16(uint8_t out[16]) { squeeze(out, 16); }
void KeccakSpongeTranscript::state(uint8_t out[32]) const {  // peek_state (legacy.rs:243-248)
    KeccakSpongeTranscript copy = *this;
    copy.squeeze(out, 32);
}

// ---- spongefish DigestBridge<Blake2b512>: one running hash while absorbing (opened by a zero mask block and the chaining value), a ratchet H(H(.)) at the first squeeze,
// ---- output blocks H(..01 mask || chaining value || block index) with the unused tail of a block kept for the next squeeze
struct Blake2bSpongeTranscript::Bridge {
    enum class Mode { Start, Absorb, Squeeze } mode = Mode::Start;
    Blake2b running{64};
    std::array<uint8_t, 64> chaining{}, block{};
    size_t block_left = 0;
    uint64_t blocks_out = 0;
    static void mask(Blake2b& h, uint8_t tag) {
        std::array<uint8_t, 128> m{};
        m.back() = tag;
        h.update(m.data(), m.size());
    }
    static void be64(Blake2b& h, uint64_t v) {
        uint8_t b[8];
        for (int i = 0; i < 8; ++i) b[7 - i] = (uint8_t)(v >> (8 * i));
        h.update(b, 8);
    }
    void close_squeeze() {
        if (mode != Mode::Squeeze) return;
        Blake2b h(64);
        mask(h, 0x02);
        h.update(chaining.data(), 64);
        be64(h, 64 * blocks_out - block_left);
        h.finish(chaining.data());
        running = Blake2b(64);
        mode = Mode::Start;
        block_left = 0;
    }
    void absorb(const uint8_t* p, size_t n) {
        close_squeeze();
        if (mode == Mode::Start) {
            mode = Mode::Absorb;
            mask(running, 0x00);
            running.update(chaining.data(), 64);
        }
        running.update(p, n);
    }
    void squeeze(uint8_t* out, size_t n) {
        if (mode == Mode::Absorb) {
            std::array<uint8_t, 64> once;
            running.finish(once.data());
            blake2b_digest(once.data(), 64, 64, chaining.data());
            running = Blake2b(64);
            mode = Mode::Start;
        }
        if (mode == Mode::Start) {
            mode = Mode::Squeeze;
            blocks_out = 0;
            block_left = 0;
            mask(running, 0x01);
            running.update(chaining.data(), 64);
        }
        for (size_t done = 0; done < n;) {
            if (block_left == 0) {
                Blake2b h = running;
                be64(h, blocks_out++);
                h.finish(block.data());
                block_left = 64;
            }
            const size_t take = std::min(n - done, block_left);
            std::memcpy(out + done, block.data() + (64 - block_left), take);
            block_left -= take;
            done += take;
        }
    }
};
Blake2bSpongeTranscript::Blake2bSpongeTranscript(const uint8_t* label, size_t n) : bridge(std::make_shared<Bridge>()) {
    uint8_t protocol_id[64] = {};
    static const char id[] = "a16z/jolt-transcript/v1";
    std::memcpy(protocol_id, id, sizeof id - 1);
    bridge->absorb(protocol_id, 64);
    std::vector<uint8_t> session(8 + n);
    for (int i = 0; i < 8; ++i) session[i] = (uint8_t)((uint64_t)n >> (8 * i));
    std::memcpy(session.data() + 8, label, n);
    bridge->absorb(session.data(), session.size());
    bridge->absorb(nullptr, 0);
}
void Blake2bSpongeTranscript::append_bytes(const uint8_t* b, size_t n) {
    std::vector<uint8_t> framed(9 + n);
    framed[0] = 0x9B;
    for (int i = 0; i < 8; ++i) framed[1 + i] = (uint8_t)((uint64_t)n >> (8 * i));
    if (n) std::memcpy(framed.data() + 9, b, n);
    bridge->absorb(framed.data(), framed.size());
}
void Blake2bSpongeTranscript::draw16(uint8_t out[16]) { bridge->squeeze(out, 16); }
void Blake2bSpongeTranscript::state(uint8_t out[32]) const {
    Bridge copy = *bridge;
    copy.squeeze(out, 32);
}

LabelledTranscript::LabelledTranscript(int kind, const uint8_t* label, size_t n) {
    if (n > 32) return;
    if (kind == 1) inner.reset(new LegacyBlake2bTranscript(label, n));
    else if (kind == 2) inner.reset(new KeccakSpongeTranscript(label, n));
    else if (kind == 3) inner.reset(new Blake2bSpongeTranscript(label, n));
}
LabelledTranscript::LabelledTranscript(uint64_t label) {
    const int kind = (int)(label >> 62);
    if (kind == 0) {
        inner.reset(new MockTranscript(label));
        return;
    }
    const std::string text = "jolt-amd/" + std::to_string(label & ((1ull << 62) - 1));
    if (kind == 1) inner.reset(new LegacyBlake2bTranscript(reinterpret_cast<const uint8_t*>(text.data()), text.size()));
    else if (kind == 2) inner.reset(new KeccakSpongeTranscript(reinterpret_cast<const uint8_t*>(text.data()), text.size()));
    else inner.reset(new Blake2bSpongeTranscript(reinterpret_cast<const uint8_t*>(text.data()), text.size()));
}

// ---- DeviceMember --------------------------------------------------------------------------------------------
size_t DeviceMember::num_rounds() const { return m->rounds; }
size_t DeviceMember::n_evals() const { return jolt_internal_member_n_evals(m); }

bool DeviceMember::next_l1(bool has_bind, const Fr& c, Fr* l1) const {
    if (!m->has_split_eq()) return false;
    size_t bound = m->bound;
    Fr scalar = m->current_scalar;
    if (has_bind) {  // split_eq.rs:334-337, as member_note_bind will apply it
        if (bound >= m->rounds) return false;
        Fr p = m->w[m->rounds - bound - 1];
        Fr prod = mul(p, c);
        scalar = mul(scalar, add(add(sub(sub(Fr::one(), p), c), prod), prod));
        bound += 1;
    }
    if (bound >= m->rounds) return false;
    *l1 = mul(scalar, m->w[m->rounds - bound - 1]);
    return true;
}

int32_t DeviceMember::assemble(const Fr* evals, const Fr& previous_claim, UnivariatePoly* out, const Fr* inv_l1) const {
    if (m->kind == jolt_member::kSplitEqProduct || m->kind == jolt_member::kSplitEqBooleanity) {
        // ram_hamming_booleanity.rs:128-135 / booleanity.rs:628-632: message = gruen_poly_deg_3(q(0), q(inf), previous_claim)
        size_t current_index = m->rounds - m->bound;
        return gruen_poly_deg_3(m->current_scalar, m->w[current_index - 1], evals[0], evals[1], previous_claim, out, inv_l1);
    }
    if (m->kind == jolt_member::kSplitEqUniform) {
        size_t current_index = m->rounds - m->bound;
        return gruen_poly_from_q(m->current_scalar, m->w[current_index - 1], evals, m->uni_F, previous_claim, out, inv_l1);
    }
    if (m->eq_weighted) {  // eq * q members: q(0), q(2), .., q(dq) -> s = l * q (GruenRoundMessage::checked_round_poly, support.rs:340-412)
        size_t current_index = m->rounds - m->bound;
        return gruen_poly_from_q(m->current_scalar, m->w[current_index - 1], evals, m->degree, previous_claim, out, inv_l1);
    }
    std::vector<Fr> full;
    if (m->skip_one) {
        // support.rs:450-459 round_poly_from_skipped_evals
        full.push_back(evals[0]);
        full.push_back(sub(previous_claim, evals[0]));
        for (uint32_t t = 1; t < m->degree; ++t) full.push_back(evals[t]);
    } else {
        full.assign(evals, evals + m->degree + 1);
        // naive.rs:298-306 round check
        if (add(full[0], full[1]) != previous_claim) return JOLT_ERR_ROUND_CHECK;
    }
    *out = UnivariatePoly::from_evals(full.data(), full.size());
    return JOLT_OK;
}

int32_t DeviceMember::prove_round(const Fr* bind, size_t /*round*/, const Fr& previous_claim, UnivariatePoly* out) {
    jolt_fr_t evals[JOLT_MAX_DEGREE + 1];
    jolt_fr_t b;
    if (bind) fr_to_abi(&b, *bind);
    JOLT_TRY(jolt_member_prove_round(m, bind ? &b : nullptr, evals, n_evals(), nullptr));
    Fr ev[JOLT_MAX_DEGREE + 1];
    for (size_t i = 0; i < n_evals(); ++i) ev[i] = fr_from_abi(&evals[i]);
    return assemble(ev, previous_claim, out);
}
int32_t DeviceMember::finish_rounds(const Fr& bind) {
    jolt_fr_t b;
    fr_to_abi(&b, bind);
    return jolt_member_finish(m, &b);
}

// ---- schedulers ----------------------------------------------------------------------------------------------
int32_t SequentialRounds::batch_prove_round(std::vector<MemberRound>& work) {
    for (MemberRound& item : work) {
        JOLT_TRY(item.member->prove_round(item.has_bind ? &item.bind : nullptr, item.local_round, item.claim, &item.message));
        item.has_message = true;
    }
    return JOLT_OK;
}
int32_t SequentialRounds::batch_finish_rounds(std::vector<MemberFinish>& finishes) {
    for (MemberFinish& f : finishes) JOLT_TRY(f.member->finish_rounds(f.bind));
    return JOLT_OK;
}

int32_t DeviceGroupedRounds::batch_prove_round(std::vector<MemberRound>& work) {
    std::vector<jolt_member*> ms;
    std::vector<jolt_fr_t> bind_store(work.size());
    std::vector<const jolt_fr_t*> binds;
    size_t total = 0;
    for (size_t i = 0; i < work.size(); ++i) {
        DeviceMember* dm = dynamic_cast<DeviceMember*>(work[i].member);
        if (!dm) return JOLT_ERR_UNSUPPORTED;  // mixed host/device batches go through SequentialRounds
        ms.push_back(dm->m);
        if (work[i].has_bind) { fr_to_abi(&bind_store[i], work[i].bind); binds.push_back(&bind_store[i]); }
        else binds.push_back(nullptr);
        total += dm->n_evals();
    }
    std::vector<jolt_fr_t> evals(total ? total : 1);
    // the one inversion of each split-eq message depends only on the challenge: compute it while the device runs the round
    std::vector<Fr> l1(work.size()), inv_l1(work.size());
    std::vector<char> has_l1(work.size(), 0);
    for (size_t i = 0; i < work.size(); ++i)
        has_l1[i] = static_cast<DeviceMember*>(work[i].member)->next_l1(work[i].has_bind, work[i].bind, &l1[i]) && !l1[i].is_zero() ? 1 : 0;
    const std::function<void()> overlap = [&]() {
        for (size_t i = 0; i < work.size(); ++i)
            if (has_l1[i]) inv_l1[i] = inv(l1[i]);
    };
    JOLT_TRY(jolt_internal_round_group_prove(ctx, ms.data(), ms.size(), binds.data(), evals.data(), evals.size(), &overlap));
    size_t off = 0;
    for (size_t i = 0; i < work.size(); ++i) {
        DeviceMember* dm = static_cast<DeviceMember*>(work[i].member);
        Fr ev[JOLT_MAX_DEGREE + 1];
        for (size_t k = 0; k < dm->n_evals(); ++k) ev[k] = fr_from_abi(&evals[off + k]);
        off += dm->n_evals();
        JOLT_TRY(dm->assemble(ev, work[i].claim, &work[i].message, has_l1[i] ? &inv_l1[i] : nullptr));
        work[i].has_message = true;
    }
    return JOLT_OK;
}
int32_t DeviceGroupedRounds::batch_finish_rounds(std::vector<MemberFinish>& finishes) {
    std::vector<jolt_member*> ms;
    std::vector<jolt_fr_t> store(finishes.size());
    std::vector<const jolt_fr_t*> binds;
    for (size_t i = 0; i < finishes.size(); ++i) {
        DeviceMember* dm = dynamic_cast<DeviceMember*>(finishes[i].member);
        if (!dm) return JOLT_ERR_UNSUPPORTED;
        ms.push_back(dm->m);
        fr_to_abi(&store[i], finishes[i].bind);
        binds.push_back(&store[i]);
    }
    return jolt_round_group_finish(ctx, ms.data(), ms.size(), binds.data());
}

// ---- prove_batch ---------------------------------------------------------------------------------------------
BatchPrelude BatchPrelude::make(std::vector<BatchMember> members, size_t max_num_vars, size_t max_degree) {
    BatchPrelude p;
    Fr sum = Fr::zero();
    for (const BatchMember& m : members) sum = add(sum, mul(m.coefficient, fr_mul_pow_2(m.input_claim, max_num_vars - m.rounds)));  // batch.rs:57-64
    p.members = std::move(members);
    p.claimed_sum = sum;
    p.max_num_vars = max_num_vars;
    p.max_degree = max_degree;
    return p;
}

int32_t prove_batch(const BatchPrelude& prelude, std::vector<ProveRounds*>& members, RoundScheduler& scheduler, Transcript& transcript,
                    bool full_width_challenges, ProvedBatch* out, SumcheckError* err) {
    auto fail = [&](int32_t s, size_t round) { if (err) { err->status = s; err->round = round; } return s; };
    if (members.size() != prelude.members.size()) return fail(JOLT_ERR_SIZE_MISMATCH, 0);  // BatchMemberCountMismatch
    for (size_t i = 0; i < members.size(); ++i) {
        if (members[i]->num_rounds() != prelude.members[i].rounds) return fail(JOLT_ERR_SIZE_MISMATCH, 0);            // BatchMemberRoundsMismatch
        if (prelude.members[i].offset + prelude.members[i].rounds > prelude.max_num_vars) return fail(JOLT_ERR_INVALID_ARG, 0);  // WindowOutOfRange
    }
    const size_t max_num_vars = prelude.max_num_vars;
    if (max_num_vars > 0 && prelude.max_degree < 1) return fail(JOLT_ERR_INVALID_ARG, 0);  // ZeroBatchDegree
    const Fr two_inv = small_inverse(2);
    std::vector<Fr> member_claims;
    for (const BatchMember& m : prelude.members) member_claims.push_back(fr_mul_pow_2(m.input_claim, max_num_vars - m.rounds));  // prover.rs:244-248
    Fr running_claim = prelude.claimed_sum;
    std::vector<Fr> challenges;
    std::vector<bool> has_pending(members.size(), false);
    std::vector<Fr> pending(members.size(), Fr::zero());
    out->round_polys.clear();

    for (size_t round = 0; round < max_num_vars; ++round) {
        std::vector<Fr> batched(prelude.max_degree + 1, Fr::zero());
        std::vector<MemberRound> work;
        for (size_t index = 0; index < members.size(); ++index) {
            const BatchMember& described = prelude.members[index];
            bool active = round >= described.offset && round < described.offset + described.rounds;
            if (!active) {  // prover.rs:273-282
                member_claims[index] = mul(member_claims[index], two_inv);
                batched[0] = add(batched[0], mul(described.coefficient, member_claims[index]));
                continue;
            }
            MemberRound mr;
            mr.index = index;
            mr.local_round = round - described.offset;
            mr.has_bind = has_pending[index];
            mr.bind = pending[index];
            has_pending[index] = false;
            mr.claim = member_claims[index];
            mr.member = members[index];
            mr.has_message = false;
            work.push_back(std::move(mr));
        }
        int32_t s = scheduler.batch_prove_round(work);
        if (s != JOLT_OK) return fail(s, round);
        for (const MemberRound& item : work) {
            if (!item.has_message) return fail(JOLT_ERR_INVALID_ARG, round);                                // MissingRoundMessage
            if (item.message.degree() > prelude.max_degree) return fail(JOLT_ERR_UNSUPPORTED, round);       // DegreeBoundExceeded
            const BatchMember& described = prelude.members[item.index];
            for (size_t k = 0; k < item.message.coefficients.size(); ++k)
                batched[k] = add(batched[k], mul(described.coefficient, item.message.coefficients[k]));
        }
        // trim_round_polynomial (prover.rs:168-177)
        while (batched.size() > 2 && batched.back().is_zero()) batched.pop_back();
        UnivariatePoly batched_poly;
        batched_poly.coefficients = batched;
        Fr round_sum = add(batched_poly.evaluate(Fr::zero()), batched_poly.evaluate(Fr::one()));
        if (round_sum != running_claim) return fail(JOLT_ERR_ROUND_CHECK, round);  // prover.rs:316-324
        // ClearSumcheckRecorder::absorb_round (recorder.rs:118-130): compressed poly (linear term omitted), then challenge
        transcript.append_round_poly(kSumcheckRoundLabel, batched_poly.coefficients.data(), batched_poly.coefficients.size());
        Fr challenge = full_width_challenges ? transcript.challenge_scalar() : transcript.challenge();
        running_claim = batched_poly.evaluate(challenge);
        challenges.push_back(challenge);
        out->round_polys.push_back(batched_poly);
        for (const MemberRound& item : work) {
            member_claims[item.index] = item.message.evaluate(challenge);
            pending[item.index] = challenge;
            has_pending[item.index] = true;
        }
    }
    std::vector<MemberFinish> finishes;  // prover.rs:343-355
    for (size_t i = 0; i < members.size(); ++i)
        if (has_pending[i]) finishes.push_back(MemberFinish{pending[i], members[i]});
    int32_t s = scheduler.batch_finish_rounds(finishes);
    if (s != JOLT_OK) return fail(s, max_num_vars);
    out->challenges = std::move(challenges);
    out->final_claim = running_claim;
    out->member_claims = std::move(member_claims);
    return JOLT_OK;
}

}  // namespace jolt_host

// ------------------------------------------------------------------------------------------------------------------
// C exports
// ------------------------------------------------------------------------------------------------------------------
using namespace jolt_host;

extern "C" int32_t jolt_host_fr_mul(const jolt_fr_t* a, const jolt_fr_t* b, jolt_fr_t* out) {
    if (!a || !b || !out) return JOLT_ERR_INVALID_ARG;
    fr_to_abi(out, mul(fr_from_abi(a), fr_from_abi(b)));
    return JOLT_OK;
}
// The kernels' multiplication algorithm (field.hip.h mul_limbs29: product scanning over nine 29-bit limbs), compiled for the host
// so that the CPU suite pins it against the oracle; field 0 = Fr, 1 = Fq.  Operands must be canonical.
extern "C" int32_t jolt_host_mul_limbs29(int32_t field, const jolt_fr_t* a, const jolt_fr_t* b, jolt_fr_t* out) {
    if (!a || !b || !out) return JOLT_ERR_INVALID_ARG;
    if (field == 0) {
        fr_to_abi(out, jolt::mul_limbs29(fr_from_abi(a), fr_from_abi(b)));
    } else if (field == 1) {
        jolt::Fq x, y;
        std::memcpy(&x, a, sizeof(x));
        std::memcpy(&y, b, sizeof(y));
        jolt::Fq r = jolt::mul_limbs29(x, y);
        std::memcpy(out, &r, sizeof(r));
    } else {
        return JOLT_ERR_INVALID_ARG;
    }
    return JOLT_OK;
}
// The limb-form Fq arithmetic of the bucket sums (fq_limb.hip.h), compiled for the host so that the CPU suite pins it against the oracle.
// Operands are canonical Fq in STANDARD Montgomery form; they are taken to L-form (x * 2^261: a product with the Montgomery form of 32),
// combined there, and the result is brought back.  op 0: a * b   1: a^2   2: a * b + c * d (one reduction)   3: (a - b) * c through the lazily
// reduced difference a + 8p - b   4: (a - b - 2c) * d through a + 6p - b - 2c.
#include "g1.hip.h"
#include "fq_limb.hip.h"
extern "C" int32_t jolt_host_fq_limb_op(int32_t op, const jolt_fr_t* a, const jolt_fr_t* b, const jolt_fr_t* c, const jolt_fr_t* d, jolt_fr_t* out) {
    if (!a || !out || (op != 1 && !b) || (op >= 2 && !c) || ((op == 2 || op == 4) && !d)) return JOLT_ERR_INVALID_ARG;
    using namespace jolt;
    Fq thirty_two = Fq::zero();
    thirty_two.l[0] = 32;
    const Fq m32 = to_mont(thirty_two);
    auto load = [&](const jolt_fr_t* p) {
        Fq x;
        std::memcpy(&x, p, sizeof(x));
        return fql_from_words(mul(x, m32));
    };
    const FqL r256 = fql_from_words(Fq::one());
    FqL r;
    switch (op) {
        case 0: r = fql_mul(load(a), load(b)); break;
        case 1: r = fql_sqr(load(a)); break;
        case 2: r = fql_mul2(load(a), load(b), load(c), load(d)); break;
        case 3: r = fql_mul(fql_diff<8, false>(load(a), load(b), load(a)), load(c)); break;
        case 4: r = fql_mul(fql_diff<6, true>(load(a), load(b), load(c)), load(d)); break;
        default: return JOLT_ERR_INVALID_ARG;
    }
    const Fq res = fql_to_std(r, r256);
    std::memcpy(out, &res, sizeof(res));
    return JOLT_OK;
}
// sum of `count` affine points (standard Montgomery coordinates, (0, 0) = infinity; negate[i] != 0 adds -P_i) through the limb-form XYZZ
// accumulator of the bucket kernels: g1xl_add_mixed with its identity / doubling / P + (-P) branches, then the conversion back
extern "C" int32_t jolt_host_g1_sum_limb_form(const uint64_t* points /* count x 8 u64: x, y */, const uint8_t* negate, size_t count, jolt_g1_t* out) {
    if ((!points && count) || !out) return JOLT_ERR_INVALID_ARG;
    using namespace jolt;
    Fq thirty_two = Fq::zero();
    thirty_two.l[0] = 32;
    const Fq m32 = to_mont(thirty_two);
    const FqL one = fql_from_words(m32), r256 = fql_from_words(Fq::one());
    G1XyzzL acc = g1xl_identity();
    for (size_t i = 0; i < count; ++i) {
        G1Affine p;
        std::memcpy(&p, points + 8 * i, sizeof(p));
        if (g1_aff_is_inf(p)) continue;
        if (negate && negate[i]) p.y = neg(p.y);
        acc = g1xl_add_mixed(acc, fql_from_words(mul(p.x, m32)), fql_from_words(mul(p.y, m32)), one);
    }
    G1Jac r = g1_identity();
    if (!g1xl_is_identity(acc)) {
        r.x = fql_to_std(fql_mul(acc.x, fql_sqr(acc.zz)), r256);
        r.y = fql_to_std(fql_mul(acc.y, fql_sqr(acc.zzz)), r256);
        r.z = fql_to_std(acc.zzz, r256);
    }
    std::memcpy(out, &r, sizeof(r));
    return JOLT_OK;
}
extern "C" int32_t jolt_host_fr_add(const jolt_fr_t* a, const jolt_fr_t* b, jolt_fr_t* out) {
    if (!a || !b || !out) return JOLT_ERR_INVALID_ARG;
    fr_to_abi(out, add(fr_from_abi(a), fr_from_abi(b)));
    return JOLT_OK;
}
extern "C" int32_t jolt_host_fr_sub(const jolt_fr_t* a, const jolt_fr_t* b, jolt_fr_t* out) {
    if (!a || !b || !out) return JOLT_ERR_INVALID_ARG;
    fr_to_abi(out, sub(fr_from_abi(a), fr_from_abi(b)));
    return JOLT_OK;
}
extern "C" int32_t jolt_host_fr_inv(const jolt_fr_t* a, jolt_fr_t* out) {
    if (!a || !out) return JOLT_ERR_INVALID_ARG;
    Fr v = fr_from_abi(a);
    if (v.is_zero()) return JOLT_ERR_NOT_INVERTIBLE;
    fr_to_abi(out, inv(v));
    return JOLT_OK;
}
extern "C" int32_t jolt_host_fr_from_u64(uint64_t v, jolt_fr_t* out) {
    if (!out) return JOLT_ERR_INVALID_ARG;
    fr_to_abi(out, fr_from_u64(v));
    return JOLT_OK;
}
// EqPolynomial::evals_serial (crates/jolt-poly/src/eq.rs:299-315) on the host: the K-entry address tables (K = 16 / 256) a
// one-hot member pre-scales are too small to be worth a device round trip.  Big-endian index, optional scale.
extern "C" int32_t jolt_host_eq_evals(const jolt_fr_t* r, size_t n, const jolt_fr_t* scale, jolt_fr_t* out) {
    if ((!r && n) || !out || n > 20) return JOLT_ERR_INVALID_ARG;
    std::vector<Fr> e((size_t)1 << n);
    e[0] = scale ? fr_from_abi(scale) : Fr::one();
    size_t size = 1;
    for (size_t j = 0; j < n; ++j) {
        const Fr rj = fr_from_abi(&r[j]);
        if (!fr_is_canonical(rj)) return JOLT_ERR_INVALID_ARG;
        for (size_t i = size; i-- > 0;) {  // eq.rs:306-313: evals[2i+1] = s * r_j, evals[2i] = s - evals[2i+1]
            const Fr s = e[i];
            const Fr hi = mul(s, rj);
            e[2 * i + 1] = hi;
            e[2 * i] = sub(s, hi);
        }
        size *= 2;
    }
    for (size_t i = 0; i < size; ++i) fr_to_abi(&out[i], e[i]);
    return JOLT_OK;
}

extern "C" int32_t jolt_host_fr_mul_shifted(const jolt_fr_t* a, const jolt_fr_t* c, jolt_fr_t* out) {
    if (!a || !c || !out) return JOLT_ERR_INVALID_ARG;
    Fr cc = fr_from_abi(c);
    if (!fr_low_limbs_zero(cc)) return JOLT_ERR_INVALID_ARG;
    uint32_t chi[4] = {cc.l[4], cc.l[5], cc.l[6], cc.l[7]};
    fr_to_abi(out, mul_shifted(fr_from_abi(a), chi));
    return JOLT_OK;
}
extern "C" int32_t jolt_host_univariate_from_evals(const jolt_fr_t* evals, size_t n, jolt_fr_t* coeffs_out) {
    if (!evals || !coeffs_out || n == 0 || n > 16) return JOLT_ERR_INVALID_ARG;
    std::vector<Fr> e(n);
    for (size_t i = 0; i < n; ++i) e[i] = fr_from_abi(&evals[i]);
    UnivariatePoly p = UnivariatePoly::from_evals(e.data(), n);
    for (size_t i = 0; i < n; ++i) fr_to_abi(&coeffs_out[i], p.coefficients[i]);
    return JOLT_OK;
}
extern "C" int32_t jolt_host_univariate_evaluate(const jolt_fr_t* coeffs, size_t n, const jolt_fr_t* x, jolt_fr_t* out) {
    if (!coeffs || !x || !out) return JOLT_ERR_INVALID_ARG;
    UnivariatePoly p;
    for (size_t i = 0; i < n; ++i) p.coefficients.push_back(fr_from_abi(&coeffs[i]));
    fr_to_abi(out, p.evaluate(fr_from_abi(x)));
    return JOLT_OK;
}
extern "C" int32_t jolt_host_gruen_poly_deg_3(const jolt_fr_t* current_scalar, const jolt_fr_t* point_i, const jolt_fr_t* q_constant,
                                              const jolt_fr_t* q_quadratic, const jolt_fr_t* s0_plus_s1, jolt_fr_t* coeffs_out) {
    if (!current_scalar || !point_i || !q_constant || !q_quadratic || !s0_plus_s1 || !coeffs_out) return JOLT_ERR_INVALID_ARG;
    UnivariatePoly p;
    JOLT_TRY(gruen_poly_deg_3(fr_from_abi(current_scalar), fr_from_abi(point_i), fr_from_abi(q_constant), fr_from_abi(q_quadratic),
                              fr_from_abi(s0_plus_s1), &p));
    for (size_t i = 0; i < 4; ++i) fr_to_abi(&coeffs_out[i], p.coefficients[i]);
    return JOLT_OK;
}

// ---- booleanity address phase (stage 6a), host half: OptimizedBooleanityAddressKernel::{prove_round, bind} (crates/jolt-kernels/src/optimized/
// booleanity.rs:320-398) over the K-entry pushforward masses jolt_onehot_pushforward produced.  16 .. 256 points per table: the reference keeps this
// loop on the host ("negligible next to the T-scale table construction", :46-49) and so does a Rust caller; these two entry points let the
// Python stage driver (jolt_amd/stages.py) run it without per-element FFI calls.  Tables: n_polys rows of `stride` entries, the first `len` live.
extern "C" int32_t jolt_host_booleanity_address_round(const jolt_fr_t* linear, const jolt_fr_t* squared, size_t n_polys, size_t stride, size_t len, const jolt_fr_t* weights,
                                                      const jolt_fr_t* eq_address, jolt_fr_t* evals_out /* 4 */) {
    if (!linear || !squared || !weights || !eq_address || !evals_out || len < 2 || (len & (len - 1)) || len > stride) return JOLT_ERR_INVALID_ARG;
    const size_t half = len / 2;
    for (uint64_t c = 0; c < 4; ++c) {
        const Fr point = fr_from_u64(c), point_sqr = mul(point, point), om = sub(Fr::one(), point), one_minus_sqr = mul(om, om);
        Fr sum = Fr::zero();
        for (size_t y = 0; y < half; ++y) {
            Fr inner = Fr::zero();
            for (size_t i = 0; i < n_polys; ++i) {
                const jolt_fr_t *sq = squared + i * stride, *lin = linear + i * stride;
                const Fr s0 = fr_from_abi(&sq[2 * y]), s1 = fr_from_abi(&sq[2 * y + 1]), l0 = fr_from_abi(&lin[2 * y]), l1 = fr_from_abi(&lin[2 * y + 1]);
                const Fr squared_ext = add(mul(one_minus_sqr, s0), mul(point_sqr, s1));
                const Fr linear_ext = add(l0, mul(point, sub(l1, l0)));
                inner = add(inner, mul(fr_from_abi(&weights[i]), sub(squared_ext, linear_ext)));
            }
            const Fr e0 = fr_from_abi(&eq_address[2 * y]), e1 = fr_from_abi(&eq_address[2 * y + 1]);
            sum = add(sum, mul(add(e0, mul(point, sub(e1, e0))), inner));
        }
        fr_to_abi(&evals_out[c], sum);
    }
    return JOLT_OK;
}
extern "C" int32_t jolt_host_booleanity_address_bind(jolt_fr_t* linear, jolt_fr_t* squared, size_t n_polys, size_t stride, size_t len, jolt_fr_t* eq_address, const jolt_fr_t* challenge) {
    if (!linear || !squared || !eq_address || !challenge || len < 2 || (len & (len - 1)) || len > stride) return JOLT_ERR_INVALID_ARG;
    const Fr r = fr_from_abi(challenge);
    if (!fr_is_canonical(r)) return JOLT_ERR_INVALID_ARG;
    const size_t half = len / 2;
    const Fr om = sub(Fr::one(), r), one_minus_sqr = mul(om, om), challenge_sqr = mul(r, r);
    for (size_t i = 0; i < n_polys; ++i) {
        jolt_fr_t *lin = linear + i * stride, *sq = squared + i * stride;
        for (size_t k = 0; k < half; ++k) {
            const Fr l0 = fr_from_abi(&lin[2 * k]), l1 = fr_from_abi(&lin[2 * k + 1]);
            fr_to_abi(&lin[k], add(l0, mul(r, sub(l1, l0))));
            fr_to_abi(&sq[k], add(mul(one_minus_sqr, fr_from_abi(&sq[2 * k])), mul(challenge_sqr, fr_from_abi(&sq[2 * k + 1]))));  // squared weights: one-hot columns
        }
    }
    for (size_t k = 0; k < half; ++k) {
        const Fr e0 = fr_from_abi(&eq_address[2 * k]), e1 = fr_from_abi(&eq_address[2 * k + 1]);
        fr_to_abi(&eq_address[k], add(e0, mul(r, sub(e1, e0))));
    }
    return JOLT_OK;
}

// ---- Hamming-weight claim reduction (stage 7), host half: HammingWeightKernel (crates/jolt-kernels/src/optimized/hamming_weight_claim_reduction.rs:150-300)
// over the K_chunk-entry pushforward masses G_i of ALL RA columns (jolt_onehot_pushforward against the shared eq(r_cycle, .): the one T-scale pass of the
// relation, :83-117).  weights: W_i(k) = g^(3i) + g^(3i+1) eq(r_address, k) + g^(3i+2) eq(virt_i, k) (:187-207); round: the summand sum_i G_i W_i at
// t = 0 and t = 2 summed over the pair groups (group_evals :255-266) plus the plain sum (the input claim before the first round); bind: every table as
// a multilinear (:243-252).  Tables are n_polys rows of `stride` entries, the first `len` live.
extern "C" int32_t jolt_host_hamming_weights(const jolt_fr_t* gamma, const jolt_fr_t* r_address, const jolt_fr_t* virtualization_points, size_t n_polys, size_t log_k,
                                             jolt_fr_t* out) {
    if (!gamma || !out || (log_k && (!r_address || !virtualization_points)) || log_k > 16) return JOLT_ERR_INVALID_ARG;
    const size_t K = (size_t)1 << log_k;
    auto eq_table = [&](const jolt_fr_t* point, std::vector<Fr>& t) {  // EqPolynomial::evals, big-endian point
        t.assign(1, Fr::one());
        for (size_t v = 0; v < log_k; ++v) {
            const Fr r = fr_from_abi(&point[v]);
            std::vector<Fr> next(t.size() * 2);
            for (size_t i = 0; i < t.size(); ++i) {
                next[2 * i + 1] = mul(t[i], r);
                next[2 * i] = sub(t[i], next[2 * i + 1]);
            }
            t.swap(next);
        }
    };
    std::vector<Fr> eq_bool, eq_virt;
    eq_table(r_address, eq_bool);
    const Fr g = fr_from_abi(gamma);
    Fr power = Fr::one();
    for (size_t i = 0; i < n_polys; ++i) {
        const Fr g0 = power, g1 = mul(g0, g), g2 = mul(g1, g);
        power = mul(g2, g);
        eq_table(virtualization_points + i * log_k, eq_virt);
        for (size_t k = 0; k < K; ++k) fr_to_abi(&out[i * K + k], add(g0, add(mul(g1, eq_bool[k]), mul(g2, eq_virt[k]))));
    }
    return JOLT_OK;
}
extern "C" int32_t jolt_host_pair_tables_round(const jolt_fr_t* g, const jolt_fr_t* w, size_t n_polys, size_t stride, size_t len, jolt_fr_t* evals_out /* 3 */) {
    if (!g || !w || !evals_out || len < 1 || (len & (len - 1)) || len > stride) return JOLT_ERR_INVALID_ARG;
    Fr s0 = Fr::zero(), s2 = Fr::zero(), total = Fr::zero();
    for (size_t i = 0; i < n_polys; ++i) {
        const jolt_fr_t *gi = g + i * stride, *wi = w + i * stride;
        for (size_t y = 0; y < len / 2; ++y) {
            const Fr g_lo = fr_from_abi(&gi[2 * y]), g_hi = fr_from_abi(&gi[2 * y + 1]), w_lo = fr_from_abi(&wi[2 * y]), w_hi = fr_from_abi(&wi[2 * y + 1]);
            s0 = add(s0, mul(g_lo, w_lo));
            s2 = add(s2, mul(sub(add(g_hi, g_hi), g_lo), sub(add(w_hi, w_hi), w_lo)));
            total = add(total, add(mul(g_lo, w_lo), mul(g_hi, w_hi)));
        }
        if (len == 1) total = add(total, mul(fr_from_abi(&gi[0]), fr_from_abi(&wi[0])));
    }
    fr_to_abi(&evals_out[0], s0);
    fr_to_abi(&evals_out[1], s2);
    fr_to_abi(&evals_out[2], total);
    return JOLT_OK;
}
extern "C" int32_t jolt_host_pair_tables_bind(jolt_fr_t* g, jolt_fr_t* w, size_t n_polys, size_t stride, size_t len, const jolt_fr_t* challenge) {
    if (!g || !w || !challenge || len < 2 || (len & (len - 1)) || len > stride) return JOLT_ERR_INVALID_ARG;
    const Fr r = fr_from_abi(challenge);
    if (!fr_is_canonical(r)) return JOLT_ERR_INVALID_ARG;
    for (size_t i = 0; i < n_polys; ++i)
        for (jolt_fr_t* t : {g + i * stride, w + i * stride})
            for (size_t k = 0; k < len / 2; ++k) {
                const Fr lo = fr_from_abi(&t[2 * k]), hi = fr_from_abi(&t[2 * k + 1]);
                fr_to_abi(&t[k], add(lo, mul(r, sub(hi, lo))));
            }
    return JOLT_OK;
}

// ---- the caller-side Fiat-Shamir of members that are driven round by round outside prove_batch (sparse read-write matrix, read-RAF
// phases): the deterministic test transcript behind four entry points; a Rust caller uses its own Transcript instead.
struct jolt_host_transcript {
    LabelledTranscript t;
    explicit jolt_host_transcript(uint64_t label) : t(label) {}
    jolt_host_transcript(int kind, const uint8_t* label, size_t n) : t(kind, label, n) {}
};
extern "C" int32_t jolt_host_transcript_create(uint64_t label, jolt_host_transcript** out) {
    if (!out) return JOLT_ERR_INVALID_ARG;
    *out = new (std::nothrow) jolt_host_transcript(label);
    return *out ? JOLT_OK : JOLT_ERR_OOM;
}
extern "C" int32_t jolt_host_transcript_create_labelled(int32_t kind, const uint8_t* label, size_t label_len, jolt_host_transcript** out) {
    if (!out || (!label && label_len) || label_len > 32 || kind < JOLT_TRANSCRIPT_BLAKE2B_LEGACY || kind > JOLT_TRANSCRIPT_BLAKE2B_SPONGE) return JOLT_ERR_INVALID_ARG;
    *out = new (std::nothrow) jolt_host_transcript((int)kind, label, label_len);
    return *out ? JOLT_OK : JOLT_ERR_OOM;
}
extern "C" int32_t jolt_host_transcript_append_label(jolt_host_transcript* t, const char* label, int32_t with_count, uint64_t count) {
    if (!t || !label || std::strlen(label) > (with_count ? 24u : 32u)) return JOLT_ERR_INVALID_ARG;
    if (with_count) t->t.append_label_with_count(label, count);
    else t->t.append_label(label);
    return JOLT_OK;
}
extern "C" int32_t jolt_host_transcript_append_u64_word(jolt_host_transcript* t, uint64_t value) {
    if (!t) return JOLT_ERR_INVALID_ARG;
    t->t.append_u64_word(value);
    return JOLT_OK;
}
extern "C" int32_t jolt_host_transcript_append_round_poly(jolt_host_transcript* t, const char* label, const jolt_fr_t* coefficients, size_t count) {
    if (!t || !label || std::strlen(label) > 24 || (!coefficients && count)) return JOLT_ERR_INVALID_ARG;
    std::vector<Fr> c(count);
    for (size_t i = 0; i < count; ++i) {
        c[i] = fr_from_abi(&coefficients[i]);
        if (!fr_is_canonical(c[i])) return JOLT_ERR_INVALID_ARG;
    }
    t->t.append_round_poly(label, c.data(), count);
    return JOLT_OK;
}
extern "C" int32_t jolt_host_transcript_state(const jolt_host_transcript* t, uint8_t* out32) {
    if (!t || !out32) return JOLT_ERR_INVALID_ARG;
    t->t.state(out32);
    return JOLT_OK;
}
extern "C" int32_t jolt_host_blake2b(const uint8_t* in, size_t n, size_t outlen, uint8_t* out) {
    if ((!in && n) || !out || outlen < 1 || outlen > 64) return JOLT_ERR_INVALID_ARG;
    blake2b_digest(in, n, outlen, out);
    return JOLT_OK;
}
extern "C" int32_t jolt_host_keccak_f1600(uint8_t* state200) {
    if (!state200) return JOLT_ERR_INVALID_ARG;
    keccak_f1600(state200);
    return JOLT_OK;
}
extern "C" int32_t jolt_host_transcript_append_fr(jolt_host_transcript* t, const jolt_fr_t* values, size_t count) {
    if (!t || (!values && count)) return JOLT_ERR_INVALID_ARG;
    for (size_t i = 0; i < count; ++i) {
        const Fr v = fr_from_abi(&values[i]);
        if (!fr_is_canonical(v)) return JOLT_ERR_INVALID_ARG;
        t->t.append_fr(v);
    }
    return JOLT_OK;
}
extern "C" int32_t jolt_host_transcript_append_bytes(jolt_host_transcript* t, const uint8_t* bytes, size_t count) {
    if (!t || (!bytes && count)) return JOLT_ERR_INVALID_ARG;
    t->t.append_bytes(bytes, count);
    return JOLT_OK;
}
extern "C" int32_t jolt_host_transcript_challenge(jolt_host_transcript* t, int32_t full_width, jolt_fr_t* out) {
    if (!t || !out) return JOLT_ERR_INVALID_ARG;
    fr_to_abi(out, full_width ? t->t.challenge_scalar() : t->t.challenge());
    return JOLT_OK;
}
extern "C" int32_t jolt_host_transcript_destroy(jolt_host_transcript* t) {
    delete t;
    return JOLT_OK;
}

extern "C" int32_t jolt_host_gruen_poly_from_q(const jolt_fr_t* current_scalar, const jolt_fr_t* point_i, const jolt_fr_t* q_evals, size_t dq,
                                               const jolt_fr_t* s0_plus_s1, jolt_fr_t* coeffs_out) {
    if (!current_scalar || !point_i || !q_evals || !s0_plus_s1 || !coeffs_out || dq < 1 || dq > 8) return JOLT_ERR_INVALID_ARG;
    std::vector<Fr> q(dq);
    for (size_t i = 0; i < dq; ++i) q[i] = fr_from_abi(&q_evals[i]);
    UnivariatePoly p;
    JOLT_TRY(gruen_poly_from_q(fr_from_abi(current_scalar), fr_from_abi(point_i), q.data(), dq, fr_from_abi(s0_plus_s1), &p));
    for (size_t i = 0; i < dq + 2; ++i) fr_to_abi(&coeffs_out[i], p.coefficients[i]);
    return JOLT_OK;
}

extern "C" int32_t jolt_host_prove_batch(jolt_ctx* ctx, jolt_member* const* members, size_t n_members, const jolt_fr_t* input_claims,
                                         const jolt_fr_t* coefficients, const size_t* offsets, size_t max_num_vars, size_t max_degree,
                                         uint64_t transcript_label, int32_t challenge_mode, int32_t use_round_group, jolt_fr_t* out_polys,
                                         jolt_fr_t* out_challenges, jolt_fr_t* out_member_claims, jolt_fr_t* out_final_claim) {
    if (!ctx || (!members && n_members) || !input_claims || !coefficients || !offsets || !out_polys || !out_challenges || !out_member_claims ||
        !out_final_claim)
        return JOLT_ERR_INVALID_ARG;
    std::vector<std::unique_ptr<DeviceMember>> owned;
    std::vector<ProveRounds*> ms;
    std::vector<BatchMember> described;
    for (size_t i = 0; i < n_members; ++i) {
        if (!members[i]) return JOLT_ERR_INVALID_ARG;
        if (offsets[i] > max_num_vars || members[i]->rounds > max_num_vars - offsets[i]) return JOLT_ERR_INVALID_ARG;  // WindowOutOfRange (before the prelude scales claims by 2^(max - rounds))
        owned.emplace_back(new DeviceMember(members[i]));
        ms.push_back(owned.back().get());
        described.push_back(BatchMember{fr_from_abi(&input_claims[i]), fr_from_abi(&coefficients[i]), members[i]->rounds, offsets[i]});
    }
    BatchPrelude prelude = BatchPrelude::make(std::move(described), max_num_vars, max_degree);
    LabelledTranscript tr(transcript_label);
    SequentialRounds seq;
    DeviceGroupedRounds grouped(ctx);
    RoundScheduler& sched = use_round_group ? static_cast<RoundScheduler&>(grouped) : static_cast<RoundScheduler&>(seq);
    ProvedBatch proved;
    SumcheckError err;
    JOLT_TRY(prove_batch(prelude, ms, sched, tr, challenge_mode != 0, &proved, &err));
    const size_t stride = max_degree + 1;
    Fr zero = Fr::zero();
    for (size_t r = 0; r < max_num_vars; ++r)
        for (size_t k = 0; k < stride; ++k)
            fr_to_abi(&out_polys[r * stride + k], k < proved.round_polys[r].coefficients.size() ? proved.round_polys[r].coefficients[k] : zero);
    for (size_t r = 0; r < max_num_vars; ++r) fr_to_abi(&out_challenges[r], proved.challenges[r]);
    for (size_t i = 0; i < n_members; ++i) fr_to_abi(&out_member_claims[i], proved.member_claims[i]);
    fr_to_abi(out_final_claim, proved.final_claim);
    return JOLT_OK;
}

// sum_k a[k]*b[k] through the deferred-reduction accumulator of field.hip.h (wide_fmadd / wide_reduce), flushed every
// kWideMaxProducts products: the host build of the same code the kernels use (algebra.rs:362-433 Accumulator contract).
extern "C" int32_t jolt_host_fr_wide_dot(const jolt_fr_t* a, const jolt_fr_t* b, size_t n, jolt_fr_t* out) {
    if ((!a || !b) && n) return JOLT_ERR_INVALID_ARG;
    if (!out) return JOLT_ERR_INVALID_ARG;
    Fr total = Fr::zero();
    jolt::WideAcc<jolt::FrParams> acc = jolt::wide_zero<jolt::FrParams>();
    int pending = 0;
    for (size_t k = 0; k < n; ++k) {
        Fr x = fr_from_abi(&a[k]), y = fr_from_abi(&b[k]);
        if (!fr_is_canonical(x) || !fr_is_canonical(y)) return JOLT_ERR_INVALID_ARG;
        jolt::wide_fmadd(acc, x, y);
        if (++pending == jolt::kWideMaxProducts) {
            total = add(total, jolt::wide_reduce(acc));
            acc = jolt::wide_zero<jolt::FrParams>();
            pending = 0;
        }
    }
    if (pending) total = add(total, jolt::wide_reduce(acc));
    fr_to_abi(out, total);
    return JOLT_OK;
}
