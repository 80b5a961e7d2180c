/*
 * oracle/mock_transcript.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The reference's Fiat-Shamir transcript (crates/jolt-transcript, Blake2b/Keccak via spongefish) is
 * host-side and OUT OF SCOPE.  The hot path only needs "absorb bytes, squeeze a
 * challenge" (jolt-sumcheck/src/recorder.rs:118-130, jolt-hyperkzg/src/scheme.rs:148-152,
 * kzg.rs:87-96,118-124), so the oracle and the product-side test harness share this tiny deterministic
 * stand-in.  SPEC (the product side re-implements it from this text, not from this code):
 *
 *   state s[4] (u64) = {0x6a09e667f3bcc908 ^ label, 0xbb67ae8584caa73b, 0x3c6ef372fe94f82b, 0xa54ff53a5f1d36f1}
 *   mix(x): x += 0x9E3779B97F4A7C15; x = (x ^ (x>>30)) * 0xBF58476D1CE4E5B9;
 *           x = (x ^ (x>>27)) * 0x94D049BB133111EB; return x ^ (x>>31)                (splitmix64)
 *   absorb_word(w): s0 = mix(s0 ^ w); s1 = mix(s1 + s0); s2 ^= rotl64(s1, 23); s3 = mix(s3 ^ s2 ^ w)
 *   append_bytes(b, n): absorb_word(n); then absorb_word of every 8-byte little-endian chunk (zero padded)
 *   draw16(): absorb_word(0xC4A11E46E); lo = mix(s0 ^ s2); hi = mix(s1 ^ s3); absorb_word(lo ^ hi); bytes = LE(lo)||LE(hi)
 *   challenge()        = Fr::from_challenge_bytes(draw16())            -- 125-bit "optimized" challenge shape
 *   challenge_scalar() = Fr::from_scalar_challenge_bytes(draw16())     -- non-optimized decoding, full-width limbs
 *   (both squeeze 16 bytes, as crates/jolt-transcript/src/digest.rs:178-188 and legacy.rs:280-300 do)
 */
#pragma once
#include "fr.h"

typedef struct { uint64_t s[4]; } mock_transcript;

static inline uint64_t mt_mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static inline uint64_t mt_rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline void mt_init(mock_transcript *t, uint64_t label) {
    t->s[0] = 0x6a09e667f3bcc908ull ^ label;
    t->s[1] = 0xbb67ae8584caa73bull;
    t->s[2] = 0x3c6ef372fe94f82bull;
    t->s[3] = 0xa54ff53a5f1d36f1ull;
}
static inline void mt_absorb_word(mock_transcript *t, uint64_t w) {
    t->s[0] = mt_mix(t->s[0] ^ w);
    t->s[1] = mt_mix(t->s[1] + t->s[0]);
    t->s[2] ^= mt_rotl(t->s[1], 23);
    t->s[3] = mt_mix(t->s[3] ^ t->s[2] ^ w);
}
static inline void mt_append_bytes(mock_transcript *t, const uint8_t *b, size_t n) {
    mt_absorb_word(t, (uint64_t)n);
    for (size_t i = 0; i < n; i += 8) {
        uint64_t w = 0;
        for (size_t j = 0; j < 8 && i + j < n; ++j) w |= (uint64_t)b[i + j] << (8 * j);
        mt_absorb_word(t, w);
    }
}
static inline void mt_append_fr(mock_transcript *t, const fr_t *a) {
    uint8_t bytes[32];
    fr_to_bytes_le(bytes, *a);
    mt_append_bytes(t, bytes, 32);
}
static inline void mt_draw16(mock_transcript *t, uint8_t out[16]) {
    mt_absorb_word(t, 0xC4A11E46Eull);
    uint64_t lo = mt_mix(t->s[0] ^ t->s[2]);
    uint64_t hi = mt_mix(t->s[1] ^ t->s[3]);
    mt_absorb_word(t, lo ^ hi);
    for (int i = 0; i < 8; ++i) { out[i] = (uint8_t)(lo >> (8 * i)); out[8 + i] = (uint8_t)(hi >> (8 * i)); }
}
static inline fr_t mt_challenge(mock_transcript *t) {
    uint8_t b[16];
    mt_draw16(t, b);
    return fr_from_challenge_bytes(b, 16);
}
static inline fr_t mt_challenge_scalar(mock_transcript *t) {
    uint8_t b[16];
    mt_draw16(t, b);
    return fr_from_scalar_challenge_bytes(b, 16);
}
