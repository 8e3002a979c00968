// sha256.cpp -- the compression function behind sha256.hpp: x86 SHA extensions when the CPU has them, a portable loop otherwise
#include "sha256.hpp"
#include <cstdlib>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace kzg {

static const uint32_t SHA256_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74,
    0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d,
    0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e,
    0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5,
    0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static inline uint32_t sha_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static inline void sha256_blocks_portable(uint32_t st[8], const uint8_t *p, size_t blocks) {
    while (blocks--) {
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            const uint32_t s0 = sha_rotr(w[i - 15], 7) ^ sha_rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            const uint32_t s1 = sha_rotr(w[i - 2], 17) ^ sha_rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
        for (int i = 0; i < 64; i++) {
            const uint32_t t1 = h + (sha_rotr(e, 6) ^ sha_rotr(e, 11) ^ sha_rotr(e, 25)) + ((e & f) ^ (~e & g)) + SHA256_K[i] + w[i];
            const uint32_t t2 = (sha_rotr(a, 2) ^ sha_rotr(a, 13) ^ sha_rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
        p += 64;
    }
}

#if defined(__x86_64__)
// two rounds per sha256rnds2, four message words per register; state kept as (ABEF, CDGH) as the instruction wants it
__attribute__((target("sha,sse4.1,ssse3"))) static inline void sha256_blocks_shani(uint32_t st[8], const uint8_t *p, size_t blocks) {
    const __m128i bswap = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
    __m128i tmp = _mm_loadu_si128((const __m128i *)&st[0]);        // DCBA
    __m128i s1 = _mm_loadu_si128((const __m128i *)&st[4]);         // HGFE
    tmp = _mm_shuffle_epi32(tmp, 0xB1);                            // CDAB
    s1 = _mm_shuffle_epi32(s1, 0x1B);                              // EFGH
    __m128i s0 = _mm_alignr_epi8(tmp, s1, 8);                      // ABEF
    s1 = _mm_blend_epi16(s1, tmp, 0xF0);                           // CDGH
    while (blocks--) {
        const __m128i save0 = s0, save1 = s1;
        __m128i m[4];
        for (int i = 0; i < 4; i++) m[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(p + 16 * i)), bswap);
        for (int r = 0; r < 16; r++) {                               // four rounds per iteration
            __m128i w = m[r & 3];
            __m128i wk = _mm_add_epi32(w, _mm_loadu_si128((const __m128i *)&SHA256_K[4 * r]));
            s1 = _mm_sha256rnds2_epu32(s1, s0, wk);
            s0 = _mm_sha256rnds2_epu32(s0, s1, _mm_shuffle_epi32(wk, 0x0E));
            if (r < 12) {                                            // schedule words 16 + 4r .. 19 + 4r into the slot just consumed
                __m128i x = _mm_sha256msg1_epu32(m[r & 3], m[(r + 1) & 3]);
                x = _mm_add_epi32(x, _mm_alignr_epi8(m[(r + 3) & 3], m[(r + 2) & 3], 4));
                m[r & 3] = _mm_sha256msg2_epu32(x, m[(r + 3) & 3]);
            }
        }
        s0 = _mm_add_epi32(s0, save0);
        s1 = _mm_add_epi32(s1, save1);
        p += 64;
    }
    tmp = _mm_shuffle_epi32(s0, 0x1B);                             // FEBA
    s1 = _mm_shuffle_epi32(s1, 0xB1);                              // DCHG
    s0 = _mm_blend_epi16(tmp, s1, 0xF0);                           // DCBA
    s1 = _mm_alignr_epi8(s1, tmp, 8);                              // HGFE
    _mm_storeu_si128((__m128i *)&st[0], s0);
    _mm_storeu_si128((__m128i *)&st[4], s1);
}
#endif

static inline bool sha256_use_shani() {
#if defined(__x86_64__)
    static const bool use = [] {
        const char *e = getenv("KZG_HIP_SHA256");
        if (e && !strcmp(e, "portable")) return false;
        __builtin_cpu_init();
        return (bool)(__builtin_cpu_supports("sha") && __builtin_cpu_supports("sse4.1") && __builtin_cpu_supports("ssse3"));
    }();
    return use;
#else
    return false;
#endif
}

void sha256_blocks(uint32_t st[8], const uint8_t *p, size_t blocks) {
#if defined(__x86_64__)
    if (sha256_use_shani()) { sha256_blocks_shani(st, p, blocks); return; }
#endif
    sha256_blocks_portable(st, p, blocks);
}

}   // namespace kzg
