// k2r_cid.hip -- content addressing of the stored chunk objects on the device.
//
// The reference stores a chunk as the object  u16 MAGIC (0xDCDF + 1) | u32 FORMAT_VERSION (1) | NODE_MMSTRUCT3 (2) |
// NODE_SUBCHUNK (4) | Chunk::write_to bytes  (resolver.rs:17-18,126-138; mmstruct.rs:215-218; node.rs:11,13; integers
// big-endian, extio.rs) and names it by  CIDv1(codec 0x12, multihash sha2-256(object))  (testing.rs:172-183).
// Hashing 1.4 GB of encoded chunks is what remains of the per-chunk host work after the encode moved to the GPU
// (SURVEY 8(f) rank 3): here every encoded chunk of a session is hashed where it lies in HBM, one thread per chunk
// (SHA-256 is sequential per message; 64 chunks advance in lockstep in a wave), and only the 32-byte digests travel.
#include <hip/hip_runtime.h>

#include <vector>

#include "k2r_runtime.h"

namespace k2r {

__device__ __constant__ uint32_t kSha256K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

__device__ __forceinline__ void sha256_block(uint32_t (&h)[8], uint32_t (&w)[16]) {
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            const uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
            const uint32_t s0 = rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3), s1 = rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
        }
        const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g);
        const uint32_t t1 = hh + S1 + ch + kSha256K[i] + w[i & 15];
        const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c);
        const uint32_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

// message = the 8-byte object header followed by the chunk bytes
__device__ __forceinline__ uint32_t msg_byte(const uint8_t* data, uint64_t len, uint64_t i) {
    const uint8_t hdr[8] = {0xDC, 0xE0, 0, 0, 0, 1, 2, 4};
    if (i < 8) return hdr[i];
    i -= 8;
    return i < len ? data[i] : (i == len ? 0x80u : 0u);
}

__global__ void __launch_bounds__(64) k_object_sha256(const TileArgs* __restrict__ tiles, const TileResult* __restrict__ results,
                                                      uint32_t n, uint8_t* __restrict__ digests) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if (t >= n) return;
    uint8_t* out = digests + 32ull * t;
    const uint64_t len = results[t].status == ST_OK ? results[t].len : 0;
    if (len == 0) {
        for (int i = 0; i < 32; i++) out[i] = 0;
        return;
    }
    const uint8_t* data = tiles[t].out;
    const uint64_t total = len + 8;                       // message bytes
    const uint64_t nblk = (total + 1 + 8 + 63) / 64;      // + 0x80 + 64-bit length
    const bool aligned = ((uintptr_t)data & 7) == 0;
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    // The 64 lanes of a wave read 64 different chunks, so a block's eight loads are eight latencies of scattered lines; they
    // are issued one block ahead (nx) and land while the ~1 800 instructions of the current block run.
    auto full_block = [&](uint64_t b) { return aligned && b > 0 && 64 * b + 64 <= total; };  // chunk bytes only, aligned words
    uint2 nx[8];
    for (uint64_t b = 0; b < nblk; b++) {
        uint32_t w[16];
        const uint64_t o = 64 * b;
        if (full_block(b)) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                w[2 * j] = __builtin_bswap32(nx[j].x);
                w[2 * j + 1] = __builtin_bswap32(nx[j].y);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint64_t q = o + 4 * j;
                w[j] = (msg_byte(data, len, q) << 24) | (msg_byte(data, len, q + 1) << 16) | (msg_byte(data, len, q + 2) << 8) |
                       msg_byte(data, len, q + 3);
            }
            if (b == nblk - 1) {  // message length in bits, big-endian, in the last 8 bytes
                const uint64_t bits = total * 8;
                w[14] = (uint32_t)(bits >> 32);
                w[15] = (uint32_t)bits;
            }
        }
        if (b + 1 < nblk && full_block(b + 1)) {  // 8-byte aligned: the header shifts the stream by 8
            const uint2* p = (const uint2*)(data + (o + 64 - 8));
#pragma unroll
            for (int j = 0; j < 8; j++) nx[j] = p[j];
        }
        sha256_block(h, w);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)(h[i] >> 24);
        out[4 * i + 1] = (uint8_t)(h[i] >> 16);
        out[4 * i + 2] = (uint8_t)(h[i] >> 8);
        out[4 * i + 3] = (uint8_t)h[i];
    }
}

// SHA-256 of whole buffers (objects that already carry their header: Links and Superchunk nodes, framed chunk objects):
// buffer i = data[offs[i] .. offs[i] + lens[i])
__global__ void __launch_bounds__(64) k_sha256_buffers(const uint8_t* __restrict__ data, const uint64_t* __restrict__ offs,
                                                       const uint64_t* __restrict__ lens, uint32_t n, uint8_t* __restrict__ digests) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if (t >= n) return;
    const uint8_t* m = data + offs[t];
    const uint64_t total = lens[t];
    const uint64_t nblk = (total + 1 + 8 + 63) / 64;
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    for (uint64_t b = 0; b < nblk; b++) {
        uint32_t w[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            uint32_t x = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint64_t i = 64 * b + 4 * j + q;
                x = (x << 8) | (i < total ? (uint32_t)m[i] : (i == total ? 0x80u : 0u));
            }
            w[j] = x;
        }
        if (b == nblk - 1) {
            const uint64_t bits = total * 8;
            w[14] = (uint32_t)(bits >> 32);
            w[15] = (uint32_t)bits;
        }
        sha256_block(h, w);
    }
    uint8_t* out = digests + 32ull * t;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)(h[i] >> 24);
        out[4 * i + 1] = (uint8_t)(h[i] >> 16);
        out[4 * i + 2] = (uint8_t)(h[i] >> 8);
        out[4 * i + 3] = (uint8_t)h[i];
    }
}
hipError_t launch_sha256_buffers(const uint8_t* data, const uint64_t* offs, const uint64_t* lens, uint32_t n, uint8_t* digests, hipStream_t stream) {
    hipLaunchKernelGGL(k_sha256_buffers, dim3((n + 63) / 64), dim3(64), 0, stream, data, offs, lens, n, digests);
    return hipGetLastError();
}

hipError_t launch_object_sha256(const TileArgs* tiles, const TileResult* results, uint32_t n, uint8_t* digests, hipStream_t stream) {
    hipLaunchKernelGGL(k_object_sha256, dim3((n + 63) / 64), dim3(64), 0, stream, tiles, results, n, digests);
    return hipGetLastError();
}

}  // namespace k2r
