/*
 * oracle/fs_hash.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The two hash primitives behind the reference's Fiat-Shamir transcripts, restated from their public specifications
 * (neither lives under /root/reference: `blake2` 0.10 and `spongefish` rev d2d190b1 are Cargo dependencies, Cargo.toml:279):
 *   * BLAKE2b with a 32-byte digest (RFC 7693, unkeyed) -- the `blake2::Blake2b<U32>` of
 *     jolt_transcript::LegacyBlake2bTranscript (crates/jolt-transcript/src/lib.rs:70-75), the transcript the reference's own
 *     benchmark profile proves with (crates/jolt-prover/src/profile.rs:69);
 *   * Keccak-f[1600] (FIPS 202) and spongefish's duplex sponge over it in OVERWRITE mode, rate 136 / capacity 64, all-zero
 *     initial state -- `spongefish::instantiations::Keccak`, behind jolt_transcript::KeccakTranscript (lib.rs:77-80).
 * Pinned by tests/test_transcript_cpu.py: RFC 7693 appendix A + hashlib for BLAKE2b, the reference's own known-answer vector
 * (crates/jolt-transcript/tests/keccak_tests.rs:13-29) for the sponge.
 */
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

/* ---- BLAKE2b (RFC 7693 section 3) ---- */
typedef struct {
    uint64_t h[8];
    uint64_t t[2];
    uint8_t buf[128];
    size_t buflen, outlen;
} orc_blake2b;

static const uint64_t ORC_B2B_IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                                       0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
static const uint8_t ORC_B2B_SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};

static inline uint64_t orc_rotr64(uint64_t x, int k) { return (x >> k) | (x << (64 - k)); }
static inline uint64_t orc_load64_le(const uint8_t *p) {
    uint64_t v = 0;
    for (int i = 0; i < 8; ++i) v |= (uint64_t)p[i] << (8 * i);
    return v;
}
static inline void orc_blake2b_compress(orc_blake2b *s, const uint8_t block[128], int last) {
    uint64_t m[16], v[16];
    for (int i = 0; i < 16; ++i) m[i] = orc_load64_le(block + 8 * i);
    for (int i = 0; i < 8; ++i) { v[i] = s->h[i]; v[i + 8] = ORC_B2B_IV[i]; }
    v[12] ^= s->t[0];
    v[13] ^= s->t[1];
    if (last) v[14] = ~v[14];
#define ORC_B2B_G(a, b, c, d, x, y)                                   \
    do {                                                              \
        v[a] = v[a] + v[b] + (x); v[d] = orc_rotr64(v[d] ^ v[a], 32); \
        v[c] = v[c] + v[d];       v[b] = orc_rotr64(v[b] ^ v[c], 24); \
        v[a] = v[a] + v[b] + (y); v[d] = orc_rotr64(v[d] ^ v[a], 16); \
        v[c] = v[c] + v[d];       v[b] = orc_rotr64(v[b] ^ v[c], 63); \
    } while (0)
    for (int r = 0; r < 12; ++r) {
        const uint8_t *sg = ORC_B2B_SIGMA[r];
        ORC_B2B_G(0, 4, 8, 12, m[sg[0]], m[sg[1]]);
        ORC_B2B_G(1, 5, 9, 13, m[sg[2]], m[sg[3]]);
        ORC_B2B_G(2, 6, 10, 14, m[sg[4]], m[sg[5]]);
        ORC_B2B_G(3, 7, 11, 15, m[sg[6]], m[sg[7]]);
        ORC_B2B_G(0, 5, 10, 15, m[sg[8]], m[sg[9]]);
        ORC_B2B_G(1, 6, 11, 12, m[sg[10]], m[sg[11]]);
        ORC_B2B_G(2, 7, 8, 13, m[sg[12]], m[sg[13]]);
        ORC_B2B_G(3, 4, 9, 14, m[sg[14]], m[sg[15]]);
    }
#undef ORC_B2B_G
    for (int i = 0; i < 8; ++i) s->h[i] ^= v[i] ^ v[i + 8];
}
static inline void orc_blake2b_init(orc_blake2b *s, size_t outlen) {
    for (int i = 0; i < 8; ++i) s->h[i] = ORC_B2B_IV[i];
    s->h[0] ^= 0x01010000ull ^ (uint64_t)outlen; /* parameter block: digest length, no key, fanout = depth = 1 */
    s->t[0] = s->t[1] = 0;
    s->buflen = 0;
    s->outlen = outlen;
}
static inline void orc_blake2b_update(orc_blake2b *s, const uint8_t *in, size_t n) {
    while (n > 0) {
        if (s->buflen == 128) { /* a full buffer is only compressed when more input follows: the last block carries the final flag */
            s->t[0] += 128;
            if (s->t[0] < 128) s->t[1]++;
            orc_blake2b_compress(s, s->buf, 0);
            s->buflen = 0;
        }
        size_t take = 128 - s->buflen;
        if (take > n) take = n;
        memcpy(s->buf + s->buflen, in, take);
        s->buflen += take;
        in += take;
        n -= take;
    }
}
static inline void orc_blake2b_final(orc_blake2b *s, uint8_t *out) {
    s->t[0] += s->buflen;
    if (s->t[0] < s->buflen) s->t[1]++;
    memset(s->buf + s->buflen, 0, 128 - s->buflen);
    orc_blake2b_compress(s, s->buf, 1);
    for (size_t i = 0; i < s->outlen; ++i) out[i] = (uint8_t)(s->h[i / 8] >> (8 * (i % 8)));
}

/* ---- Keccak-f[1600] (FIPS 202 section 3) on the 200-byte state, lanes little-endian ---- */
static const uint64_t ORC_KECCAK_RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull, 0x0000000080000001ull,
    0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
    0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull,
    0x000000000000800aull, 0x800000008000000aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
static inline uint64_t orc_rotl64(uint64_t x, int k) { return k ? (x << k) | (x >> (64 - k)) : x; }
static inline void orc_keccak_f1600(uint8_t st[200]) {
    static const int rot[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14}; /* index x + 5 y */
    uint64_t a[25], b[25], c[5], d[5];
    for (int i = 0; i < 25; ++i) a[i] = orc_load64_le(st + 8 * i);
    for (int round = 0; round < 24; ++round) {
        for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ orc_rotl64(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; ++i) a[i] ^= d[i % 5];
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y) b[y + 5 * ((2 * x + 3 * y) % 5)] = orc_rotl64(a[x + 5 * y], rot[x + 5 * y]); /* rho + pi */
        for (int y = 0; y < 5; ++y)
            for (int x = 0; x < 5; ++x) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= ORC_KECCAK_RC[round];
    }
    for (int i = 0; i < 25; ++i)
        for (int k = 0; k < 8; ++k) st[8 * i + k] = (uint8_t)(a[i] >> (8 * k));
}

/* spongefish `DuplexSponge<KeccakF1600>` (its published construction, overwrite mode): absorbing WRITES the input over the rate
 * part, permuting when the rate is full before more input; a squeeze after an absorb permutes first; an absorb after a squeeze
 * discards what is left of the squeezed block. */
enum { ORC_KECCAK_RATE = 136 };
typedef struct {
    uint8_t st[200];
    uint32_t absorb_pos, squeeze_pos;
} orc_keccak_duplex;
static inline void orc_duplex_init(orc_keccak_duplex *d) {
    memset(d->st, 0, 200);
    d->absorb_pos = 0;
    d->squeeze_pos = ORC_KECCAK_RATE;
}
static inline void orc_duplex_absorb(orc_keccak_duplex *d, const uint8_t *in, size_t n) {
    d->squeeze_pos = ORC_KECCAK_RATE;
    while (n > 0) {
        if (d->absorb_pos == ORC_KECCAK_RATE) {
            orc_keccak_f1600(d->st);
            d->absorb_pos = 0;
        } else {
            size_t take = ORC_KECCAK_RATE - d->absorb_pos;
            if (take > n) take = n;
            memcpy(d->st + d->absorb_pos, in, take);
            d->absorb_pos += (uint32_t)take;
            in += take;
            n -= take;
        }
    }
}
static inline void orc_duplex_squeeze(orc_keccak_duplex *d, uint8_t *out, size_t n) {
    while (n > 0) {
        if (d->squeeze_pos == ORC_KECCAK_RATE) {
            d->squeeze_pos = 0;
            d->absorb_pos = 0;
            orc_keccak_f1600(d->st);
        }
        size_t take = ORC_KECCAK_RATE - d->squeeze_pos;
        if (take > n) take = n;
        memcpy(out, d->st + d->squeeze_pos, take);
        d->squeeze_pos += (uint32_t)take;
        out += take;
        n -= take;
    }
}
