// k2r_sha256_host.h -- SHA-256 on the host, for the CIDs of objects whose bytes are in host memory anyway.
//
// A stored object's CID is the SHA-256 of its bytes (testing.rs:172-183).  The hash is one serial chain per object: on the GPU one
// lane per object runs it at one dependent instruction per ~9 cycles (139 ms for the 256 sub-chunk objects, 1.4 MB each, of a
// [32, 4096, 4096] slice: k2r_cid.hip), the x86 SHA extensions do 2.4 GB/s per core.  When `dcdf_superchunk_build`
// returns its objects in host memory, the download's worker threads hash every object as it lands (k2r_superchunk.hip); objects that
// stay on the device are hashed there (dcdf_encoder_object_sha256).  Portable fallback for hosts without the extension.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace k2r {

namespace sha256_detail {

static const uint32_t K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

inline void blocks_portable(uint32_t st[8], const uint8_t* p, size_t nblocks) {
    while (nblocks--) {
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
        for (int i = 0; i < 64; i++) {
            const uint32_t t1 = h + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
        p += 64;
    }
}

#if defined(__x86_64__)
// The x86 SHA extensions: sha256rnds2 does two rounds on the state kept as (ABEF, CDGH); sha256msg1 / msg2 the message schedule.
__attribute__((target("sha,sse4.1,ssse3"))) inline void blocks_shani(uint32_t st[8], const uint8_t* p, size_t nblocks) {
    const __m128i shuf = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);  // big-endian words
    __m128i tmp = _mm_loadu_si128((const __m128i*)&st[0]);     // DCBA
    __m128i s1 = _mm_loadu_si128((const __m128i*)&st[4]);      // HGFE
    tmp = _mm_shuffle_epi32(tmp, 0xB1);                        // CDAB
    s1 = _mm_shuffle_epi32(s1, 0x1B);                          // EFGH
    __m128i s0 = _mm_alignr_epi8(tmp, s1, 8);                  // ABEF
    s1 = _mm_blend_epi16(s1, tmp, 0xF0);                       // CDGH
    while (nblocks--) {
        const __m128i a0 = s0, a1 = s1;
        __m128i m[4], msg;
        for (int i = 0; i < 4; i++) m[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 16 * i)), shuf);
        // sixteen groups of four rounds; group g consumes schedule words 4g .. 4g+3 (held in m[g & 3])
        for (int g = 0; g < 16; g++) {
            __m128i& cur = m[g & 3];
            msg = _mm_add_epi32(cur, _mm_loadu_si128((const __m128i*)&K[4 * g]));
            s1 = _mm_sha256rnds2_epu32(s1, s0, msg);
            msg = _mm_shuffle_epi32(msg, 0x0E);
            s0 = _mm_sha256rnds2_epu32(s0, s1, msg);
            if (g < 12) {  // words 4(g+4) .. 4(g+4)+3 replace this group's words: W[t] = s1(W[t-2]) + W[t-7] + s0(W[t-15]) + W[t-16]
                __m128i& n1 = m[(g + 1) & 3];   // W[t-12 .. t-9]
                __m128i& n2 = m[(g + 2) & 3];   // W[t-8 .. t-5]
                __m128i& n3 = m[(g + 3) & 3];   // W[t-4 .. t-1]
                __m128i x = _mm_sha256msg1_epu32(cur, n1);            // W[t-16] + s0(W[t-15])
                x = _mm_add_epi32(x, _mm_alignr_epi8(n3, n2, 4));     // + W[t-7]
                cur = _mm_sha256msg2_epu32(x, n3);                    // + s1(W[t-2])
            }
        }
        s0 = _mm_add_epi32(s0, a0);
        s1 = _mm_add_epi32(s1, a1);
        p += 64;
    }
    tmp = _mm_shuffle_epi32(s0, 0x1B);                 // FEBA
    s1 = _mm_shuffle_epi32(s1, 0xB1);                  // DCHG
    s0 = _mm_blend_epi16(tmp, s1, 0xF0);               // DCBA
    s1 = _mm_alignr_epi8(s1, tmp, 8);                  // HGFE
    _mm_storeu_si128((__m128i*)&st[0], s0);
    _mm_storeu_si128((__m128i*)&st[4], s1);
}
inline bool have_shani() {
    static const bool ok = __builtin_cpu_supports("sha") && __builtin_cpu_supports("sse4.1") && __builtin_cpu_supports("ssse3");
    return ok;
}
#endif

inline void blocks(uint32_t st[8], const uint8_t* p, size_t nblocks) {
#if defined(__x86_64__)
    if (have_shani()) {
        blocks_shani(st, p, nblocks);
        return;
    }
#endif
    blocks_portable(st, p, nblocks);
}

}  // namespace sha256_detail

// SHA-256 of data[0..len) -> out[32]
inline void sha256_host(const uint8_t* data, size_t len, uint8_t out[32]) {
    uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    const size_t full = len / 64;
    sha256_detail::blocks(st, data, full);
    uint8_t tail[128];
    const size_t rem = len - 64 * full;
    memset(tail, 0, sizeof(tail));
    if (rem) memcpy(tail, data + 64 * full, rem);
    tail[rem] = 0x80;
    const size_t tl = rem + 1 + 8 <= 64 ? 64 : 128;
    const uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    sha256_detail::blocks(st, tail, tl / 64);
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)(st[i] >> 24);
        out[4 * i + 1] = (uint8_t)(st[i] >> 16);
        out[4 * i + 2] = (uint8_t)(st[i] >> 8);
        out[4 * i + 3] = (uint8_t)st[i];
    }
}

}  // namespace k2r
