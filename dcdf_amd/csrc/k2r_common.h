// k2r_common.h -- shared definitions for the MI355X K^2-raster engine.
//
// The kernel bodies in k2r_encode.h / k2r_decode.h are written against a tiny
// "execution context" (k2r_exec.h).  The shipped library instantiates them
// with the gfx950 context only (HIP kernels in k2r_kernels.hip).  The very same
// source is also compiled by g++ with a sequential context in tests/sim/ so the
// kernel logic can be checked (ASan/UBSan) on a machine without a GPU before it
// ever touches the card.  That simulator is test infrastructure: it is not part
// of libdcdf_k2r.so and nothing in the product path can reach it.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define K2R_HD __host__ __device__ __forceinline__
#define K2R_D __device__ __forceinline__
#else
#define K2R_HD inline __attribute__((always_inline))
#define K2R_D inline __attribute__((always_inline))
#endif

namespace k2r {

// status codes stored per tile (mirror include/dcdf_k2r.h)
enum : int32_t {
    ST_OK = 0,
    ST_BAD_ARG = -1,
    ST_NONFINITE = -2,
    ST_PRECISION = -3,
    ST_OVERFLOW = -4,
    ST_UNSUPPORTED = -8,
    ST_OUT_CAPACITY = -100,  // internal: output slot too small, host retries with a bigger slot
    ST_INTERNAL = -101,      // an internal consistency guard tripped (bug); TileResult.dbg has the record
    ST_RESPLIT = -102,       // internal: a chunk encoded in speculative parts whose assumption failed; the host re-runs it whole
};

enum : int32_t { ENC_I32 = 4, ENC_I64 = 8, ENC_F32 = 32, ENC_F64 = 64 };

// Fast path value-range contract: every stored value v of the tile satisfies |v| < 2^30, so all
// node differences are exact in int32 and every zig-zag code fits 4 bytes.
constexpr int32_t VALUE_LIMIT = 1 << 30;

struct TileArgs {  // one Chunk::build input (device-visible copy of dcdf_tile_desc + output slot)
    const void* base;
    int64_t st, sr, sc;  // strides in elements
    uint32_t instants, rows, cols;
    int32_t dtype;
    uint32_t fbits;
    uint32_t round;
    uint8_t* out;      // output slot
    uint64_t out_cap;  // bytes
    int64_t* minmax;   // [instants][2] or null
    uint32_t stash_words;  // 0 = default; else caps the LDS words the log stash may use (k2r_encode.h; tests, A/B runs)
    // Part of a chunk (k2r_encode.h "speculative parts"): instants [inst_begin, inst_end) only; inst_end == 0 means all.
    // inst_begin > 0 = a continuation: it assumes the chunk's first block is still open at inst_begin, with instant 0 as its
    // snapshot and inst_begin instants in it, and writes its Logs / Blocks from byte 0 of `out` (no chunk header).
    uint32_t inst_begin;
    uint32_t inst_end;
    uint32_t _reserved;
    // A chunk encoded in parts shares instant 0's compact snapshot copy: the part that starts at instant 0 writes it to
    // shared_cmp and sets shared_flag[0] (1 = copy valid, 2 = no compact copy: range beyond 16 bits, 3 = gave up), with the
    // copy's base value in shared_flag[1]; continuations wait for the flag.  Both null for a chunk encoded whole.
    uint32_t* shared_flag;
    uint32_t* shared_cmp;
};
constexpr uint32_t PART_CMP = 1, PART_NOCMP = 2, PART_FAILED = 3;

constexpr int NPROF = 20;
struct TileResult {
    int32_t status;
    uint32_t snapshots;
    uint32_t logs;
    uint32_t stash_logs;  // diagnostic: logs emitted from the LDS stash (no re-read of the input)
    uint64_t len;
    uint32_t dbg[6];  // guard record when status == ST_INTERNAL: count, code, instant, tid, value, limit
    uint32_t carry_count;  // continuation: instants of the inherited block (when it closed, or at the end if it never did)
    uint32_t _pad2;
    uint64_t prof[NPROF];  // shader-clock cycles per phase (only filled by -DK2R_PROFILE diagnostic builds)
#ifdef K2R_PROFILE
    uint64_t pw[16][NPROF][2];  // per wave and phase: [0] cycles of its own work, [1] cycles parked at barriers (k2r_exec.h)
#endif
};

K2R_HD uint32_t popc32(uint32_t x) { return (uint32_t)__builtin_popcount(x); }
K2R_HD uint32_t popc64(uint64_t x) { return (uint32_t)__builtin_popcountll(x); }

// zig-zag of a 32-bit difference (dac.rs:134-137 restricted to |n| < 2^31)
K2R_HD uint32_t zz32(int32_t n) { return ((uint32_t)n << 1) ^ (uint32_t)(n >> 31); }

// bitmap.rs:169-171
K2R_HD uint32_t bitmap_size(uint32_t nbits) { return 8u + 4u * (nbits / 128u) + 4u * ((nbits + 31u) / 32u); }

// compact the even bits of x (bit0,2,4,..) into the low half
K2R_HD uint32_t compact_even(uint32_t x) {
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0f0f0f0fu;
    x = (x | (x >> 4)) & 0x00ff00ffu;
    x = (x | (x >> 8)) & 0x0000ffffu;
    return x;
}
// Morton index (row bit above col bit at every level; snapshot.rs:468-474 child order i*k+j)
K2R_HD void morton_decode(uint32_t m, uint32_t& row, uint32_t& col) {
    col = compact_even(m);
    row = compact_even(m >> 1);
}

// big-endian u32 store at an arbitrary byte address (extio.rs:228-233)
K2R_HD void store_be32(uint8_t* p, uint32_t w) {
    p[0] = (uint8_t)(w >> 24);
    p[1] = (uint8_t)(w >> 16);
    p[2] = (uint8_t)(w >> 8);
    p[3] = (uint8_t)w;
}
K2R_HD uint32_t load_be32(const uint8_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    // one (possibly unaligned) 4-byte load + byte swap instead of four byte loads: the serialized stream has no alignment
    typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
    return __builtin_bswap32(*(const u32_unaligned*)p);
#else
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
#endif
}

}  // namespace k2r
