// k2r_encode.h -- fused Heuristic K^2-Raster chunk encoder (k = 2; built and dispatched for sidelen 16..256, sidelen 8 goes to
// the universal kernel k2r_generic.hip; the simulator suite still instantiates sidelen 8).
//
// Replaces, for one tile, the whole of `Chunk::build` + `Chunk::write_to`
// (reference chunk.rs:42-96,235-243; snapshot.rs:108-156; log.rs:112-165; dac.rs:96-132;
// bitmap.rs:44-112) with a data-parallel formulation.  One workgroup marches over the
// instants of one chunk (the heuristic is sequential, chunk.rs:55-74); inside an instant
// everything is parallel over the tile.
//
// Parallel formulation (not how the reference does it; results are bit-identical):
//  * thread `tid` owns the 8x8 cell block whose Morton index is `tid` (row bit above col
//    bit == the reference's child order i*k+j, snapshot.rs:468-474).  The block is streamed as four 4x4
//    sub-blocks, so heights 0..3 of the quadtree are thread-local and level order == (tid, local Morton)
//    order at every level.
//  * heights 4..H live in LDS ("top" arrays), built bottom-up.
//  * "internal" is a LOCAL predicate.  Snapshot: P(n) = min(n) != max(n) (snapshot.rs:133).
//    Log: P(n) = min_t != max_t and not equal(n) (log.rs:137-152).  Both are monotone
//    (P(child) => P(parent)), so a node is visited iff P(parent) and is internal iff P(n):
//    no top-down pass is needed.
//  * stream positions come from ONE workgroup exclusive scan of per-thread internal counts
//    per height: first child of internal node #r at height h+1 sits at
//    offV[h] + 4*r (cf. the decoder's 1 + rank(T, i)*k^2, snapshot.rs:177).
//  * sizes are functions of counts only, so both candidates (Snapshot
//    and Log, chunk.rs:57-60) are only COUNTED; the winner of chunk.rs:62 alone is emitted.
//  * a Log is emitted from an LDS "stash" of what phase 1 saw below height 2 (no second read of the
//    input); when all its values fit two bytes, phase 1 also counts the second bytes per level, so
//    that every byte of both Dacs is placed in a single visit of each stash record (EM_ONE).
//  * Snapshots (1 instant in ~32) and logs with wider values take the general path: the tile is
//    re-read by work-list passes, values longer than a byte go to an overflow list and are placed
//    by rank over the previous plane's continuation bitmap (the decoder's hop, dac.rs:83-90).
//  * the block's snapshot instant, compared with every later instant of the block, is kept as a
//    compact uint16 copy in global scratch when its range allows.
//
// Value-range contract of this fast path: |stored value| < 2^30 (int32 arithmetic is then
// exact and every zig-zag code fits 4 bytes).  Tiles outside it report ST_UNSUPPORTED.
#pragma once
#include "k2r_exec.h"

// -DK2R_REGCOPY: the open block's compact snapshot copy lives in registers (EncRegs::cw) instead of global scratch -- no re-read
// of 128 KB per instant, but 32 of the 128 VGPRs gone for every phase: built, bit-exact, and 27 % SLOWER (95 spilled VGPRs; 14.2
// against 11.2 ms), so it is off.

// -DK2R_DIAG_SKIP=<bits>: timing-only builds that leave a phase of the stash-log path out (the output is WRONG; every store
// position stays clamped, so it is memory-safe): what a phase costs on the critical path is the time its removal saves.
//   1 pass A   2 I records   4 Q records   8 bitmaps   16 second bytes of the top nodes   32 zero bitmaps   64 pre-pass
#ifndef K2R_DIAG_SKIP
#define K2R_DIAG_SKIP 0
#endif

namespace k2r {

template <int LOG2S>
struct EncCfg {
    static_assert(LOG2S >= 3 && LOG2S <= 8, "kernel body written for sidelen 8..256 (the library dispatches 16..256)");
    static constexpr int H = LOG2S;  // tree height; cells are height 0
    static constexpr int S = 1 << LOG2S;
    static constexpr int NBLK = 1 << (2 * (LOG2S - 3));  // 8x8 blocks in the tile
    static constexpr int NT = NBLK;                      // one thread per block
    static constexpr int NW = NT < 64 ? 1 : NT / 64;
    static constexpr int NTOP = ((1 << (2 * (H - 2))) - 1) / 3;  // nodes at heights 3..H
    static constexpr int NTOPX = NTOP - (1 << (2 * (H - 3)));     // nodes at heights 4..H (each gets its own worker thread)
    static constexpr int TBW = (NTOPX + 31) / 32 + 1;             // words of a bitset over them
    static constexpr int MAXV = ((1 << (2 * (H + 1))) - 1) / 3;  // all nodes        (|Lmax|)
    static constexpr int MAXT = ((1 << (2 * H)) - 1) / 3;        // nodes w/ children (|T|, max |Lmin|)
    static constexpr int WV = (MAXV + 31) / 32;
    static constexpr int WT = (MAXT + 31) / 32;
    // offset of height h (3..H) inside the top arrays; height 3 first
    static constexpr int top_off(int h) { return ((1 << (2 * (H - 2))) - (1 << (2 * (H - h + 1)))) / 3; }
};

struct EncRegs {
    // What emission needs to know about the four height-2 nodes of this thread's 8x8 block, carried from analysis:
    // the log candidate's values as int16 pairs (Lmax | Lmin << 16; meaningful when the log is "narrow") and flag
    // bits.  The snapshot candidate's values (and those of wide logs) are recomputed from the tile by the passes
    // that emit them -- 1 instant in 32 -- because 16 more live registers across phase 1 meant spills, and a spill
    // reload behind global stores waits for those stores (s_waitcnt vmcnt counts both on gfx9).
    uint32_t d2[4];
    uint32_t flags;  // bits 0-3: eq2[j] (all cells of node j differ from the snapshot by one constant); 4-27: internal
                     // quads per node (snapshot, log); 28: wide; bits 29-31 unused
    uint32_t u2;     // bit j: node j is uniform (or entirely outside the tile)
    uint32_t sc[MAX_SCAN_FIELDS];  // words handed to ex.scan<>/ex.reduce<> (layout: see phase 3)
    uint64_t pf_lo;                // saved exclusive prefix of the chosen candidate's lo pack
    uint32_t pf_l2;                // second bytes before this block's height-2 group (Lmax | Lmin << 16)
    uint32_t pa[5];                // what pass A emitted for nodes of heights >= 3 (positions, second bytes), replayed later
#ifdef K2R_REGCOPY
    // The compact copy of the open block's snapshot instant (unpadded tiles): 64 cells of this thread as packed uint16 offsets
    // from s_base, in the layout of store_compact -- 128 KB per workgroup that are compared with every later instant of the block
    // and would otherwise be fetched again for each (a third of phase 1's bytes, a quarter of the kernel's HBM traffic).
    // GPU: words [8p, 8p+8) = the sub-block this lane analyses in pass p of phase 1; sequential context: sub-block j = p of its block.
    uint32_t cw[32];
#endif
};

// ---- packed scan fields ------------------------------------------------------------------
// lo : I1 @0 (16b) | I2 @16 (14b) | I3 @30 (12b)        internal counts at heights 1..3
// top: I4 @0 (10b) | I5 @10 (8b) | I6 @18 (6b) | I7 @24 (4b) | I8 @28 (2b)
// mx : c1 @0 | c2 @18 | c3 @36   (# Lmax values needing > 1, > 2, > 3 bytes)
// mn : c1 @0 | c2 @16 | c3 @32   (# Lmin values ...)
K2R_HD uint32_t unpackI(int h, uint64_t lo, uint64_t top) {
    switch (h) {
        case 1: return (uint32_t)(lo & 0xffffu);
        case 2: return (uint32_t)((lo >> 16) & 0x3fffu);
        case 3: return (uint32_t)((lo >> 30) & 0xfffu);
        case 4: return (uint32_t)(top & 0x3ffu);
        case 5: return (uint32_t)((top >> 10) & 0xffu);
        case 6: return (uint32_t)((top >> 18) & 0x3fu);
        case 7: return (uint32_t)((top >> 24) & 0xfu);
        case 8: return (uint32_t)((top >> 28) & 0x3u);
    }
    return 0;
}
K2R_HD uint64_t packTop(int h) {  // contribution of one internal node at height h (4..8)
    switch (h) {
        case 4: return 1ull;
        case 5: return 1ull << 10;
        case 6: return 1ull << 18;
        case 7: return 1ull << 24;
        case 8: return 1ull << 28;
    }
    return 0;
}

struct Cls {  // counts of values needing > 1, > 2, > 3 bytes
    uint32_t c1 = 0, c2 = 0, c3 = 0;
    K2R_HD void add(uint32_t zz, bool on) {
        c1 += (on && zz > 0xffu) ? 1u : 0u;
        c2 += (on && zz > 0xffffu) ? 1u : 0u;
        c3 += (on && zz > 0xffffffu) ? 1u : 0u;
    }
    // four values at once; the common case (all one byte) costs three ORs and a compare
    K2R_HD void add4(uint32_t z0, uint32_t z1, uint32_t z2, uint32_t z3, bool on) {
        if (on && ((z0 | z1 | z2 | z3) > 0xffu)) {
            add(z0, true);
            add(z1, true);
            add(z2, true);
            add(z3, true);
        }
    }
    // only the "> 1 byte" class, straight from the signed value: zig-zag(v) > 0xff  <=>  v < -128 || v > 127
    K2R_HD void add1(int32_t v, bool on) { c1 += (on && ((uint32_t)v + 128u) > 255u) ? 1u : 0u; }
    K2R_HD void add(const Cls& o, bool on) {
        c1 += on ? o.c1 : 0u;
        c2 += on ? o.c2 : 0u;
        c3 += on ? o.c3 : 0u;
    }
};

// ---- geometry helpers ---------------------------------------------------------------------
// cell (dr,dc) of an 8x8 block -> Morton register index
constexpr int cell_m(int dr, int dc) {
    return 16 * ((((dr >> 2) & 1) << 1) | ((dc >> 2) & 1)) + 4 * ((((dr >> 1) & 1) << 1) | ((dc >> 1) & 1)) +
           (((dr & 1) << 1) | (dc & 1));
}
// Morton index m (0..63) -> (dr,dc)
constexpr int m_dr(int m) { return (((m >> 5) & 1) << 2) | (((m >> 3) & 1) << 1) | ((m >> 1) & 1); }
constexpr int m_dc(int m) { return (((m >> 4) & 1) << 2) | (((m >> 2) & 1) << 1) | (m & 1); }

K2R_HD int32_t min4(int32_t a, int32_t b, int32_t c, int32_t d) {
    int32_t x = a < b ? a : b, y = c < d ? c : d;
    return x < y ? x : y;
}
K2R_HD int32_t max4(int32_t a, int32_t b, int32_t c, int32_t d) {
    int32_t x = a > b ? a : b, y = c > d ? c : d;
    return x > y ? x : y;
}

// in-kernel marker for "value outside the fast path's range"; ranks below every conversion error in the
// workgroup-wide lds_min so that a genuine reference panic (fixed.rs:39-70) is what gets reported
constexpr int32_t ERR_RANGE = -1;

// ---- fixed point (fixed.rs:31-71), arithmetic in the input float type ------------------------
template <class F>
K2R_HD int64_t to_fixed_dev(F n, uint32_t bits, bool round, int32_t& err) {
    if (n != n) return 0;
    if (!(n - n == (F)0)) {  // +-inf
        err = ST_NONFINITE;
        return 0;
    }
    F shifted = n * (F)((int64_t)1 << bits);
    const F tr = (F)__builtin_trunc((double)shifted);  // exact for float and double
    const F fr = shifted - tr;
    if (fr > (F)0) {
        if (round) {
            // round half away from zero; shifted > 0 here because fract() > 0 only for positives
            F fl = tr;
            shifted = (shifted - fl >= (F)0.5) ? fl + (F)1 : fl;
        } else {
            if (err == 0) err = ST_PRECISION;
            return 0;
        }
    }
    shifted = shifted * (F)2;
    if (!(shifted >= (F)-9223372036854775808.0 && shifted < (F)9223372036854775808.0)) {
        if (err == 0) err = ST_OVERFLOW;
        return 0;
    }
    return (int64_t)shifted + 1;
}

// one cell -> stored int64 (MMBuffer3::get, mmbuffer.rs:301-308,565-571,627-633)
K2R_HD int64_t load_stored(const TileArgs& ta, int64_t off, int32_t& err) {
    switch (ta.dtype) {
        case ENC_I32: return (int64_t)((const int32_t*)ta.base)[off];
        case ENC_I64: return ((const int64_t*)ta.base)[off];
        case ENC_F32: return to_fixed_dev<float>(((const float*)ta.base)[off], ta.fbits, ta.round != 0, err);
        default: return to_fixed_dev<double>(((const double*)ta.base)[off], ta.fbits, ta.round != 0, err);
    }
}
K2R_HD int32_t narrow(int64_t v, int32_t& err) {
    if (v < -(int64_t)VALUE_LIMIT || v >= (int64_t)VALUE_LIMIT) {
        if (err == 0) err = ERR_RANGE;
        return 0;
    }
    return (int32_t)v;
}

// Four float32 cells -> stored values (VEC == 2).  Fast path: a cell that is NaN, or whose scaled value is an
// integer below 2^29 in magnitude, converts with a handful of VALU ops; anything else (fractions, huge values,
// infinities, rounding requested) goes through to_fixed_dev, which reproduces fixed.rs:31-71 and its error kinds.
// (the exact conversion, out of line on the GPU: inlined at every use -- 16 per sub-block, 64 per thread and phase -- its int64
// arithmetic made the float kernels four times the size of the integer ones, 53 KB of code for phase 1 alone against a 64 KB
// instruction cache, for a path the benchmark's exact values never take; returns value | error << 32)
#if defined(__HIP_DEVICE_COMPILE__)
#define K2R_OUTLINE __attribute__((noinline))
#else
#define K2R_OUTLINE
#endif
template <class F>
K2R_OUTLINE
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint64_t to_fixed_slow(F f, uint32_t fbits, uint32_t round) {
    int32_t err = 0;
    const int32_t v = narrow(to_fixed_dev<F>(f, fbits, round != 0, err), err);
    return (uint64_t)(uint32_t)v | ((uint64_t)(uint32_t)err << 32);
}
K2R_HD void fixed4_f32(const float (&f)[4], const TileArgs& ta, float scale, int32_t (&v)[4], int32_t& err) {
    bool all = ta.round == 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float w = f[i] * scale;  // exact (power-of-two scale) unless it overflows to infinity
        const bool isn = f[i] != f[i];
        const bool ok = isn || (w == __builtin_truncf(w) && __builtin_fabsf(w) < 536870912.0f);
        all = all && ok;
        v[i] = (ok && !isn) ? (int32_t)w * 2 + 1 : 0;  // fixed.rs:70: (shifted << 1) | 1; NaN is stored as 0
    }
    if (!all) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint64_t r = to_fixed_slow<float>(f[i], ta.fbits, ta.round);
            v[i] = (int32_t)(uint32_t)r;
            const int32_t re = (int32_t)(uint32_t)(r >> 32);
            if (re == ST_NONFINITE || (re != 0 && err == 0)) err = re;  // (to_fixed_dev: the first error stays, non-finite overrides)
        }
    }
}
// the same for float64 cells (VEC == 4)
K2R_HD void fixed4_f64(const double (&f)[4], const TileArgs& ta, double scale, int32_t (&v)[4], int32_t& err) {
    bool all = ta.round == 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double w = f[i] * scale;
        const bool isn = f[i] != f[i];
        const bool ok = isn || (w == __builtin_trunc(w) && __builtin_fabs(w) < 536870912.0);
        all = all && ok;
        v[i] = (ok && !isn) ? (int32_t)w * 2 + 1 : 0;
    }
    if (!all) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint64_t r = to_fixed_slow<double>(f[i], ta.fbits, ta.round);
            v[i] = (int32_t)(uint32_t)r;
            const int32_t re = (int32_t)(uint32_t)(r >> 32);
            if (re == ST_NONFINITE || (re != 0 && err == 0)) err = re;  // (to_fixed_dev: the first error stays, non-finite overrides)
        }
    }
}
K2R_HD double as_f64(int64_t x) {
    double f;
    __builtin_memcpy(&f, &x, 8);
    return f;
}
K2R_HD float as_f32(int32_t x) {
    float f;
    __builtin_memcpy(&f, &x, 4);
    return f;
}

// 16 cells of height-2 node j (rows 4*(j>>1).., cols 4*(j&1)..) in local Morton order
// VEC: 0 = generic (any dtype / strides / padding), 1 = int32 rows loaded 16 bytes at a time, 2 = the same for float32
// rows, converted to fixed point on the fly, 3 = int64 rows (two 16-byte loads per four cells), narrowed with the range
// check of the fast path's contract, 4 = float64 rows, converted like float32 ones.
template <bool PADDED, int VEC>
K2R_HD void load_sub16(const TileArgs& ta, uint32_t inst, uint32_t r0, uint32_t c0, int j, int32_t (&dst)[16],
                       int32_t& err) {
    const uint32_t rj = r0 + 4 * (j >> 1), cj = c0 + 4 * (j & 1);
    if (VEC == 3 || VEC == 4) {
        const int64_t* ib = (const int64_t*)ta.base + (int64_t)inst * ta.st;
        const uint32_t o0 = rj * (uint32_t)ta.sr + cj;
        uint32_t bad = 0;
#pragma unroll
        for (int dr = 0; dr < 4; dr++) {
            const uint32_t o = o0 + (uint32_t)dr * (uint32_t)ta.sr;
            int64_t x[4];
#if defined(__HIP_DEVICE_COMPILE__)
            typedef __attribute__((address_space(1))) const char* gptr;
            typedef long long ll2 __attribute__((ext_vector_type(2)));
            const uint32_t ob = o << 3;  // 32-bit BYTE offset (< 2^31 by the host's test)
            const ll2 a = *(__attribute__((address_space(1))) const ll2*)((gptr)ib + ob);
            const ll2 b = *(__attribute__((address_space(1))) const ll2*)((gptr)ib + ob + 16);
            x[0] = a.x; x[1] = a.y; x[2] = b.x; x[3] = b.y;
#else
            for (int i = 0; i < 4; i++) x[i] = ib[o + i];
#endif
            if (VEC == 4) {
                const double f[4] = {as_f64(x[0]), as_f64(x[1]), as_f64(x[2]), as_f64(x[3])};
                int32_t v[4];
                fixed4_f64(f, ta, (double)((int64_t)1 << ta.fbits), v, err);
#pragma unroll
                for (int dc = 0; dc < 4; dc++) dst[cell_m(dr, dc)] = v[dc];
            } else {
#pragma unroll
                for (int dc = 0; dc < 4; dc++) {
                    bad |= (uint32_t)(((uint64_t)x[dc] + (uint64_t)VALUE_LIMIT) >> 31 != 0);
                    dst[cell_m(dr, dc)] = (int32_t)x[dc];
                }
            }
        }
        if (bad && err == 0) err = ERR_RANGE;
    } else if (VEC) {
        // wave-uniform base of the instant (SGPR pair) + a 32-bit element offset per thread: the host only selects
        // this path when (rows-1)*stride_r + cols < 2^31, so the offset arithmetic stays in 32 bits
        const int32_t* ib = (const int32_t*)ta.base + (int64_t)inst * ta.st;
        const uint32_t o0 = rj * (uint32_t)ta.sr + cj;
#pragma unroll
        for (int dr = 0; dr < 4; dr++) {
            const uint32_t o = o0 + (uint32_t)dr * (uint32_t)ta.sr;
            int32_t v[4];
#if defined(__HIP_DEVICE_COMPILE__)
            typedef __attribute__((address_space(1))) const char* gptr;
            const uint32_t ob = o << 2;  // 32-bit BYTE offset (< 2^31 by the host's VEC test): "saddr + voffset" form
#ifdef K2R_NT_LOADS
            // the instant's cells are read once: streamed past the caches, so that what IS re-read (the snapshot copy) stays in L2
            typedef int i4v __attribute__((ext_vector_type(4)));
            const i4v av = __builtin_nontemporal_load((__attribute__((address_space(1))) const i4v*)((gptr)ib + ob));
            const int4 a = {av.x, av.y, av.z, av.w};
#else
            const int4 a = *(__attribute__((address_space(1))) const int4*)((gptr)ib + ob);
#endif
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
#else
            for (int i = 0; i < 4; i++) v[i] = ib[o + i];
#endif
            if (VEC == 2) {
                const float f[4] = {as_f32(v[0]), as_f32(v[1]), as_f32(v[2]), as_f32(v[3])};
                fixed4_f32(f, ta, (float)((int64_t)1 << ta.fbits), v, err);
            }
#pragma unroll
            for (int dc = 0; dc < 4; dc++) dst[cell_m(dr, dc)] = v[dc];
        }
    } else {
#pragma unroll
        for (int m = 0; m < 16; m++) {
            uint32_t r = rj + m_dr(m), c = cj + m_dc(m);
            if (PADDED) {
                r = r < ta.rows ? r : ta.rows - 1;
                c = c < ta.cols ? c : ta.cols - 1;
            }
            int64_t off = (int64_t)inst * ta.st + (int64_t)r * ta.sr + (int64_t)c * ta.sc;
            dst[m] = narrow(load_stored(ta, off, err), err);
        }
    }
}

// The compact copy of a block's snapshot instant (see encode_chunk): the 16 cells of sub-block j of thread tid as uint16
// offsets from `base`, 8 words at [(4 * tid + j) * 8, +8): height-2 nodes in Morton order.  Word 4 * p + c holds
// cell c of quad 2p (low half) and cell c of quad 2p + 1 (high half): the extremes of two quads then cost three packed
// 16-bit min / max each, and a difference against an int32 cell reads its half through an operand selector.
template <class C>
K2R_HD uint32_t compact_slot(int tid, int j) {
    return 4u * (uint32_t)tid + (uint32_t)j;  // the height-2 node's Morton index (thread = 8x8 block)
}
template <class C>
K2R_HD void load_compact_raw(const uint32_t* scmp, int tid, int j, uint32_t (&w)[8]) {
    const uint32_t* p = scmp + (size_t)compact_slot<C>(tid, j) * 8;
#if defined(__HIP_DEVICE_COMPILE__)
    // wave-uniform base (SGPR pair) + 32-bit byte offset per thread: no 64-bit address arithmetic on the VALU
    typedef __attribute__((address_space(1))) const char* gptr;
    const uint32_t ob = compact_slot<C>(tid, j) * 32u;
    const uint4 a = *(__attribute__((address_space(1))) const uint4*)((gptr)scmp + ob);
    const uint4 b = *(__attribute__((address_space(1))) const uint4*)((gptr)scmp + ob + 16);
    (void)p;
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
#else
    for (int i = 0; i < 8; i++) w[i] = p[i];
#endif
}
template <class C>
K2R_HD void load_compact(const uint32_t* scmp, int tid, int j, int32_t base, int32_t (&dst)[16]) {
    uint32_t w[8];
    load_compact_raw<C>(scmp, tid, j, w);
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            dst[8 * p + c] = base + (int32_t)(w[4 * p + c] & 0xffffu);
            dst[8 * p + 4 + c] = base + (int32_t)(w[4 * p + c] >> 16);
        }
}
template <class C>
K2R_HD void store_compact(uint32_t* scmp, int tid, int j, int32_t base, const int32_t (&src)[16]) {
    uint32_t* p = scmp + (size_t)compact_slot<C>(tid, j) * 8;
    uint32_t w[8];
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int c = 0; c < 4; c++) w[4 * q + c] = (uint32_t)(src[8 * q + c] - base) | ((uint32_t)(src[8 * q + 4 + c] - base) << 16);
#if defined(__HIP_DEVICE_COMPILE__)
    *(__attribute__((address_space(1))) uint4*)p = uint4{w[0], w[1], w[2], w[3]};
    *(__attribute__((address_space(1))) uint4*)(p + 4) = uint4{w[4], w[5], w[6], w[7]};
#else
    for (int i = 0; i < 8; i++) p[i] = w[i];
#endif
}
// two uint16 lanes per word
K2R_HD uint32_t pk_min_u16(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
#else
    const uint32_t lo = (a & 0xffffu) < (b & 0xffffu) ? (a & 0xffffu) : (b & 0xffffu), hi = (a >> 16) < (b >> 16) ? (a >> 16) : (b >> 16);
    return lo | (hi << 16);
#endif
}
K2R_HD uint32_t pk_max_u16(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
#else
    const uint32_t lo = (a & 0xffffu) > (b & 0xffffu) ? (a & 0xffffu) : (b & 0xffffu), hi = (a >> 16) > (b >> 16) ? (a >> 16) : (b >> 16);
    return lo | (hi << 16);
#endif
}
// v != 0 as 0 / 1 without a compare (the optimizer turns min(v, 1) back into compare + select, hence the asm)
K2R_HD uint32_t nz(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_min_u32 %0, 1, %1" : "=v"(r) : "v"(v));
    return r;
#else
    return v != 0 ? 1u : 0u;
#endif
}
// a * b for operands below 2^24 (a full-rate multiply; the 32-bit v_mul_lo_u32 issues at a quarter of the rate)
K2R_HD uint32_t mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return a * b;
#endif
}
// (lo & 0xffff) | (hi << 16)
K2R_HD uint32_t pk_lo16(uint32_t lo, uint32_t hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
#else
    return (lo & 0xffffu) | (hi << 16);
#endif
}

// Returns v, but opaque to the optimizer.  Everything derived from the thread index alone (block origin, the top
// node a thread looks after, bit positions...) is invariant across instants AND chunks; LICM hoists it to the
// kernel prologue, where under the 128-VGPR cap it is spilled, to be reloaded from scratch in every instant -- behind
// global stores the reload has to wait for.  Recomputing a few integer ops per phase is far cheaper.
K2R_HD uint32_t opaque(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
    return v;
}

// Keeps the instruction scheduler from hoisting the next sub-block's loads above the current sub-block's
// arithmetic (which would keep 64+ extra cell registers live and spill under the 128-VGPR cap).
K2R_HD void sched_fence() {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(K2R_NO_SCHED_FENCE)
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// the four cells of the 2x2 quad whose top-left cell is (rq,cq), row-major (== Morton) order
template <bool PADDED, int VEC>
K2R_HD void load_quad(const TileArgs& ta, uint32_t inst, uint32_t rq, uint32_t cq, int32_t (&dst)[4], int32_t& err) {
    if (VEC == 3 || VEC == 4) {
        const int64_t* ib = (const int64_t*)ta.base + (int64_t)inst * ta.st;
        const uint32_t o0 = rq * (uint32_t)ta.sr + cq;
        uint32_t bad = 0;
        int64_t raw[4];
#pragma unroll
        for (int dr = 0; dr < 2; dr++) {
            const uint32_t o = o0 + (uint32_t)dr * (uint32_t)ta.sr;
            int64_t x0, x1;
#if defined(__HIP_DEVICE_COMPILE__)
            typedef __attribute__((address_space(1))) const char* gptr;
            typedef long long ll2 __attribute__((ext_vector_type(2)));
            const ll2 a = *(__attribute__((address_space(1))) const ll2*)((gptr)ib + (o << 3));
            x0 = a.x; x1 = a.y;
#else
            x0 = ib[o]; x1 = ib[o + 1];
#endif
            raw[2 * dr] = x0;
            raw[2 * dr + 1] = x1;
        }
        if (VEC == 4) {
            const double f[4] = {as_f64(raw[0]), as_f64(raw[1]), as_f64(raw[2]), as_f64(raw[3])};
            fixed4_f64(f, ta, (double)((int64_t)1 << ta.fbits), dst, err);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                bad |= (uint32_t)(((uint64_t)raw[i] + (uint64_t)VALUE_LIMIT) >> 31 != 0);
                dst[i] = (int32_t)raw[i];
            }
        }
        if (bad && err == 0) err = ERR_RANGE;
    } else if (VEC) {
        const int32_t* ib = (const int32_t*)ta.base + (int64_t)inst * ta.st;
        const uint32_t o0 = rq * (uint32_t)ta.sr + cq;
#pragma unroll
        for (int dr = 0; dr < 2; dr++) {
            const uint32_t o = o0 + (uint32_t)dr * (uint32_t)ta.sr;
#if defined(__HIP_DEVICE_COMPILE__)
            typedef __attribute__((address_space(1))) const char* gptr;
            const uint32_t ob = o << 2;
            const int2 a = *(__attribute__((address_space(1))) const int2*)((gptr)ib + ob);
            dst[2 * dr] = a.x;
            dst[2 * dr + 1] = a.y;
#else
            dst[2 * dr] = ib[o];
            dst[2 * dr + 1] = ib[o + 1];
#endif
        }
        if (VEC == 2) {
            const float f[4] = {as_f32(dst[0]), as_f32(dst[1]), as_f32(dst[2]), as_f32(dst[3])};
            fixed4_f32(f, ta, (float)((int64_t)1 << ta.fbits), dst, err);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t r = rq + (i >> 1), c = cq + (i & 1);
            if (PADDED) {
                r = r < ta.rows ? r : ta.rows - 1;
                c = c < ta.cols ? c : ta.cols - 1;
            }
            const int64_t off = (int64_t)inst * ta.st + (int64_t)r * ta.sr + (int64_t)c * ta.sc;
            dst[i] = narrow(load_stored(ta, off, err), err);
        }
    }
}

// ---- sizes --------------------------------------------------------------
struct DacLayout {
    uint32_t n[5];       // n[j] = # values with more than j bytes (n[0] = all); n[4] = 0
    uint32_t nlev;       // dac.rs:124-128
    uint32_t bm_off[4];  // byte offset of level j's BitMap, relative to the instant's first byte
    uint32_t by_off[4];  // byte offset of level j's bytes
    uint32_t end;        // first byte after the Dac
};
K2R_HD DacLayout dac_layout(uint32_t base, uint32_t n0, uint32_t n1, uint32_t n2, uint32_t n3) {
    DacLayout L;
    L.n[0] = n0; L.n[1] = n1; L.n[2] = n2; L.n[3] = n3; L.n[4] = 0;
    L.nlev = 0;
    uint32_t off = base + 1;  // n_levels byte (dac.rs:38)
#pragma unroll
    for (int j = 0; j < 4; j++) {
        L.bm_off[j] = off;
        L.by_off[j] = off;
        if (L.n[j] > 0 && L.nlev == (uint32_t)j) {
            L.nlev = j + 1;
            off += bitmap_size(L.n[j]);
            L.by_off[j] = off;
            off += L.n[j];
        }
    }
    L.end = off;
    return L;
}

K2R_HD bool bit_test(const uint32_t* w, uint32_t b) { return (w[b >> 5] >> (b & 31)) & 1u; }

template <class C>
struct Totals {
    uint32_t Ni[C::H + 2];    // internal nodes per height (index 1..H)
    uint32_t offV[C::H + 2];  // Lmax/T index of the first node of height h
    uint32_t offI[C::H + 2];  // Lmin index of the first internal node of height h
    uint32_t offZ[C::H + 2];  // eqB index of the first T=0 node of height h
    uint32_t LT, N0, M0;
    K2R_HD void from_counts(const uint32_t* ni) {  // ni[1..H] = internal nodes per height
        constexpr int H = C::H;
#pragma unroll
        for (int h = 1; h <= H; h++) Ni[h] = ni[h];
        finish();
    }
    // internal counts of heights 1..3 come from the reduced lo pack, those of heights 4..H from the bitset
    K2R_HD void from(uint64_t lo, uint64_t top) {
        constexpr int H = C::H;
#pragma unroll
        for (int h = 1; h <= H; h++) Ni[h] = unpackI(h, lo, top);
        finish();
    }
    K2R_HD void finish() {
        constexpr int H = C::H;
        Ni[H + 1] = 0;
        uint32_t v = 0, i = 0, z = 0;
#pragma unroll
        for (int h = H; h >= 0; h--) {
            uint32_t nv = (h == H) ? 1u : 4u * Ni[h + 1];
            offV[h] = v;
            offI[h] = i;
            offZ[h] = z;
            v += nv;
            if (h >= 1) {
                i += Ni[h];
                z += nv - Ni[h];
            }
        }
        N0 = v;
        LT = offV[0];
        M0 = i;
    }
};

// What one lane decides per instant (phase 4) and every thread then reads from LDS: both candidates' level
// offsets and Dac layouts, the winner, its size.  Keeping this in LDS rather than in (replicated, wave-uniform)
// scalar registers is what keeps the kernel's register pressure in check.
template <class C>
struct InstPlan {
    Totals<C> T[2];       // [0] snapshot candidate, [1] log candidate
    DacLayout V[2], M[2]; // Lmax / Lmin Dac layouts of the candidates
    uint32_t need;        // bit 0: exact byte classes of the log wanted, bit 1: of the snapshot (staged planning)
    uint32_t narrow;      // every log value fits 16 bits
    uint32_t log_size, snap_lb, eq_off;
    uint32_t as_snapshot, use_stash, isize;
    uint32_t lngV[3], lngM[3];  // log: index (among second bytes) of the first one of heights 0..2
};

template <class C>
struct EncPool {
    // One pool of LDS words with three lives per instant (general path):
    //  (1) phase 1 .. plane-0 emission of a Log: the STASH.  While the cells of a block are in registers, phase 1
    //      records everything the emission of the log candidate will need below height 2 -- one 5-word I record per
    //      internal height-2 node (from word 0) and one 3-word Q record per internal quad (from word QBASE) -- so that
    //      emitting a Log touches no input memory at all.
    //  (2) plane-0 emission of a Snapshot, or of a Log whose stash overflowed / whose values do not fit 16 bits:
    //      the work lists L2 (internal height-2 nodes) and L1 (internal quads) of the re-reading passes.
    //  (3) Dac finishing in list mode: the second continuation bitmap and the rank prefixes of the Lmax Dac.
    static constexpr int POOL_L1 = 4 * C::NBLK;                 // word offset of L1 (16*NBLK u16 = 8*NBLK words)
    static constexpr int POOL_BMV1 = 12 * C::NBLK;              // word offset of bmV1 (WV+1 words)
    static constexpr int POOL_PREFV = POOL_BMV1 + C::WV + 1;    // word offset of prefV (WV+2 words)
#ifdef K2R_PROFILE
    static constexpr int POOLFILL = 1828 - 16 * NPROF * 4;  // (the per-wave cycle sums of the diagnostic build live in LDS too)
#else
    static constexpr int POOLFILL = 1828;
#endif
    static constexpr int POOLW = POOL_PREFV + C::WV + 2 + (C::H == 8 ? POOLFILL : 0);  // sidelen 256: LDS filled to 160 KB
    // The stash's two record kinds have fixed shares of the pool (I records from word 0, Q records from word QBASE);
    // records beyond a share go to a per-workgroup overflow area in global scratch (L2), so an instant with unusually many
    // records -- a snapshot with forced-constant 64x64 blocks makes every quad under them internal -- still takes the stash
    // path (the re-reading fallback is 2.2 x slower per instant, and the 3 % of chunks that hit it set the tail of a launch).
    static constexpr int CAPI_REC = (POOLW * 53 / 100) / 5;
    static constexpr int CAPQ_REC = (POOLW - 5 * CAPI_REC) / 3;
    static constexpr int QBASE = 5 * CAPI_REC;  // word of Q record 0 (both kinds grow upwards: an address is one multiply-add)
    static constexpr int OVI_WORDS = 5 * 4 * C::NBLK;    // every height-2 node internal
    static constexpr int OVQ_WORDS = 3 * 16 * C::NBLK;   // every quad internal
};

template <class C>
struct EncShared : EncPool<C> {
    using EncPool<C>::POOL_L1;
    using EncPool<C>::POOL_BMV1;
    using EncPool<C>::POOL_PREFV;
    using EncPool<C>::POOLW;
    int32_t tmin[C::NTOP], tmax[C::NTOP], smin[C::NTOP], smax[C::NTOP], diff[C::NTOP];
    uint32_t eq[C::NTOP];
    uint32_t pool[EncPool<C>::POOLW];
    uint32_t bmT[C::WT + 1], bmE[C::WT + 1];
    uint32_t bmV0[C::WV + 1];
    uint32_t bmM[2][C::WT + 1];
    uint32_t prefM[C::WT + 2];
    uint32_t wsum[C::NW][MAX_SCAN_FIELDS];
    uint32_t tot[MAX_SCAN_FIELDS];
    K2R_HD uint32_t* L2() { return pool; }                                  // key = blk<<2|j | (quads before) << 12
    K2R_HD uint16_t* L1() { return (uint16_t*)(pool + POOL_L1); }           // key = blk<<4|j<<2|qq
    K2R_HD uint32_t* bmV1() { return pool + POOL_BMV1; }
    K2R_HD uint32_t* prefV() { return pool + POOL_PREFV; }  // list mode: per-word rank prefixes of the Lmax Dac's plane 0
    static constexpr int PREFTOP = 48;  // words of a continuation bitmap that can hold values of heights >= 3 (<= 1365 of them)
    uint32_t prefTopV[PREFTOP + 1];     // EM_ONE mode: rank prefixes of those words only
    // per-thread exclusive prefixes for stash emission (records find their owner's): [0] I1 | I2 << 16 of the winner,
    // [1] second bytes of cell values, [2] second bytes of height-1 values: Lmax | Lmin << 16
    uint32_t pfx[3][C::NBLK];
    // Which values of a block's stash records need a second byte, left by the pre-pass over the records (sparse work) so that
    // the dense phase 1 does no per-value byte bookkeeping: lq = 4 bits per Q record of the block, in ordinal order (<= 16
    // records); li = 8 bits per I record (<= 4): Lmax flags of its four quads | Lmin flags of the internal ones << 4.  The
    // owner's counts (phase 3) and a record's "second bytes before me" (emission) are popcounts over these words.
    uint32_t lq[C::NBLK][2];
    uint32_t li[C::NBLK];
    uint32_t tbS[C::TBW], tbL[C::TBW];  // "internal" flags of the nodes at heights 4..H, bit = top_off(h) - NBLK + j
    uint32_t tbP[C::TBW + 1];           // per-word exclusive popcount prefix of the winner's flags
    uint32_t ttR[C::H + 2];             // rank (over those flags) of the first node of height h
    uint32_t nlistV, nlistM;
    uint32_t stI, stQ;  // stash record counters
    InstPlan<C> pl;
    int32_t err;
    uint32_t work;
    uint32_t fault[6];  // [0] = count, [1..5] = first record (code, instant, tid, value, limit)
    uint64_t prof[NPROF], prof_last;  // -DK2R_PROFILE only
#ifdef K2R_PROFILE
    uint64_t pw[16][NPROF][2];
#endif
};

// ---- internal consistency guard --------------------------------------------------------------------
// Every computed stream position is checked against the count it must be below before it is used as an
// address.  A violation can only come from a bug; it is recorded (first one wins) and the access skipped,
// so a logic error surfaces as ST_INTERNAL with a diagnostic instead of an out-of-bounds access on the card.
template <class EX>
K2R_HD bool guard_ok(EX& ex, bool ok, uint32_t code, uint32_t inst, uint32_t tid, uint32_t value, uint32_t limit) {
    // branch-free on the hot path: the failure bit is OR-ed into a per-thread register (published to LDS once per
    // phase by guard_flush) and the boolean predicates the guarded access
    ex.gfail |= ok ? 0u : (1u << (code & 31u));
    (void)inst; (void)tid; (void)value; (void)limit;
    return ok;
}
// position guard: returns pos when pos + span <= limit, else 0 (a safe in-range position: the tile is reported as
// failed and its output discarded) -- no branch, no divergence
// UNIFORM_SPAN = false where `span` differs from lane to lane (the eqB runs of 0..4 bits): the bound is then a vector operand
// (an "s" constraint would only be right as long as the compiler happened to keep the per-lane value in a VGPR anyway)
template <bool UNIFORM_SPAN = true, class EX>
K2R_HD uint32_t guard_pos(EX& ex, uint32_t pos, uint32_t span, uint32_t limit, uint32_t code) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(K2R_GUARD_FULL)
    // Release build of the GPU kernel: the position is clamped into [0, limit - span] with ONE instruction (median of
    // pos, 0 and the bound), which is all the memory safety needs; the recording form below cost five
    // per site (~7 % of the kernel's instructions).  A logic error then shows up as wrong bytes (parity tests) instead
    // of ST_INTERNAL; the recording form still runs in the simulator suite and in -DK2R_GUARD_FULL builds.
    int32_t r;
    if (UNIFORM_SPAN) asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"((int32_t)pos), "s"((int32_t)limit - (int32_t)span));
    else asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"((int32_t)pos), "v"((int32_t)limit - (int32_t)span));
    (void)ex; (void)code;
    return (uint32_t)r;
#else
    const bool ok = pos + span <= limit;
    ex.gfail |= ok ? 0u : (1u << (code & 31u));
    return ok ? pos : 0u;
#endif
}
template <class EX>
K2R_HD void guard_flush(EX& ex) {
    if (ex.gfail) {
        ex.lds_or(&ex.sh.fault[0], ex.gfail);
        ex.gfail = 0;
    }
}
// guard codes (bit numbers in TileResult.dbg[0])
enum : uint32_t {
    kGuardVPos = 0, kGuardVSlot = 1, kGuardMPos = 2, kGuardMSlot = 3, kGuardTOwn = 4, kGuardEOwn = 5, kGuardTRun1 = 6,
    kGuardTRun2 = 7, kGuardE2 = 8, kGuardE1 = 9, kGuardListPos = 10, kGuardListRank = 14, kGuardList2 = 18, kGuardList1 = 19,
};

// ---- LDS bitmap helpers -------------------------------------------------------------------------
// bit p lives in word p/32 at position 31-(p%32)  (bitmap.rs:176-183)
template <class EX>
K2R_HD void bm_set(EX& ex, uint32_t* bm, uint32_t p) {
    ex.lds_or(&bm[p >> 5], 0x80000000u >> (p & 31));
}
// OR a run of `len` (1..32) bits starting at bit position p; bits are right-aligned in `bits`,
// first bit of the run is the most significant of the len bits.
template <class EX>
K2R_HD void bm_or_run(EX& ex, uint32_t* bm, uint32_t p, uint32_t len, uint32_t bits) {
    if (len == 0 || bits == 0) return;
    const uint64_t v = (uint64_t)bits << (64 - len);  // left-aligned in 64
    const uint32_t sh = p & 31;
    const uint64_t w = v >> sh;  // occupies bits of words (p>>5) and (p>>5)+1
    const uint32_t hi = (uint32_t)(w >> 32), lo = (uint32_t)w;
    if (hi) ex.lds_or(&bm[p >> 5], hi);
    if (lo) ex.lds_or(&bm[(p >> 5) + 1], lo);
}
// rank1 over [0,p) using per-word exclusive prefix popcounts
K2R_HD uint32_t bm_rank(const uint32_t* bm, const uint32_t* pref, uint32_t p) {
    const uint32_t w = p >> 5, b = p & 31;
    uint32_t r = pref[w];
    if (b) r += popc32(bm[w] >> (32 - b));
    return r;
}

K2R_HD void gstore32u(uint8_t* p, uint32_t v);
K2R_HD void gstore8(uint8_t* p, uint8_t v);
// Computes per-word exclusive popcount prefixes of an LDS bitmap (pref[0..W], pref[W] = total) and
// writes the serialized BitMap (bitmap.rs:128-138) to `dst`.
template <class C, class EX>
K2R_HD void bitmap_finish_write(EX& ex, const uint32_t* bm, uint32_t nbits, uint32_t* pref, uint8_t* dst) {
    constexpr int NT = C::NT;
    const uint32_t W = (nbits + 31) / 32;
    const uint32_t CH = (W + NT - 1) / NT;  // words per thread
    ex.par_nosync([&](int tid, EncRegs& r) {
        uint32_t s = 0;
        const uint32_t w0 = (uint32_t)tid * CH;
        for (uint32_t w = w0; w < w0 + CH && w < W; w++) s += popc32(bm[w]);
        r.sc[0] = s;
    });
    ex.template scan<1>();
    ex.par([&](int tid, EncRegs& r) {
        uint32_t run = (uint32_t)r.sc[0];
        const uint32_t w0 = (uint32_t)tid * CH;
        for (uint32_t w = w0; w < w0 + CH && w < W; w++) {
            pref[w] = run;
            run += popc32(bm[w]);
        }
        if (tid == 0) pref[W] = (uint32_t)ex.sh.tot[0];
    });
    ex.par([&](int tid, EncRegs&) {
        const uint32_t nidx = nbits / 128;  // bitmap.rs:70
        if (tid == 0) {
            gstore32u(dst, __builtin_bswap32(nbits));
            gstore32u(dst + 4, __builtin_bswap32(4u));  // k, bitmap.rs:69,130
        }
        for (uint32_t b = (uint32_t)tid; b < nidx; b += NT) gstore32u(dst + 8 + 4 * b, __builtin_bswap32(pref[4 * (b + 1)]));
        uint8_t* wd = dst + 8 + 4 * nidx;
        for (uint32_t w = (uint32_t)tid; w < W; w += NT) gstore32u(wd + 4 * w, __builtin_bswap32(bm[w]));
    });
}

// Several LDS bitmaps finished in one go: one multi-field workgroup scan of per-thread popcounts, then every thread
// serializes its own word range of each bitmap -- header, rank index (bitmap.rs:97-104) and big-endian words
// (bitmap.rs:128-138) -- and leaves the per-word rank prefix in `pref` where one is asked for.
struct BmJob {
    const uint32_t* bm;  // LDS words
    uint32_t nbits;
    uint32_t* pref;      // LDS, may be null
    uint32_t prefcap;    // prefixes are kept for words [0, prefcap) only
    uint8_t* dst;        // where the serialized BitMap goes; null = bitmap absent
};
template <class C, int NB, class EX>
K2R_HD void bitmaps_finish(EX& ex, const BmJob (&J)[NB]) {
    constexpr int NT = C::NT;
    ex.par_nosync([&](int tid, EncRegs& r) {
#pragma unroll
        for (int f = 0; f < NB; f++) {
            uint32_t sum = 0;
            if (J[f].dst) {
                const uint32_t W = (J[f].nbits + 31) / 32, CH = (W + NT - 1) / NT, w0 = (uint32_t)tid * CH;
                for (uint32_t w = w0; w < w0 + CH && w < W; w++) sum += popc32(J[f].bm[w]);
            }
            r.sc[f] = sum;
        }
    });
    ex.template scan<NB, false>();
    ex.par([&](int tid, EncRegs& r) {
#pragma unroll
        for (int f = 0; f < NB; f++) {
            if (!J[f].dst) continue;
            const uint32_t nbits = J[f].nbits, W = (nbits + 31) / 32, CH = (W + NT - 1) / NT, w0 = (uint32_t)tid * CH;
            const uint32_t nidx = nbits / 128;  // bitmap.rs:70
            uint8_t* const dst = J[f].dst;
            uint8_t* const wd = dst + 8 + 4 * nidx;
            if (tid == 0) {
                gstore32u(dst, __builtin_bswap32(nbits));
                gstore32u(dst + 4, __builtin_bswap32(4u));  // k, bitmap.rs:69,130
            }
            uint32_t run = r.sc[f];
            for (uint32_t w = w0; w < w0 + CH && w < W; w++) {
                const uint32_t x = J[f].bm[w];
                if (J[f].pref && w < J[f].prefcap) J[f].pref[w] = run;
                run += popc32(x);
                gstore32u(wd + 4 * w, __builtin_bswap32(x));
                if ((w & 3u) == 3u && (w >> 2) < nidx) gstore32u(dst + 8 + 4 * (w >> 2), __builtin_bswap32(run));
            }
            if (J[f].pref && w0 < W && w0 + CH >= W && W < J[f].prefcap) J[f].pref[W] = run;  // total, for rank(len)
        }
    });
}

// ---- emission sink ---------------------------------------------------------------------------------
// One Dac being written: plane-0 bytes go straight to memory, longer values go to the overflow list.
struct DacSink {
    uint8_t* plane0;   // where byte 0 of value #pos goes
    uint32_t* bm0;     // LDS continuation bitmap of plane 0
    uint64_t* list;    // overflow entries (pos << 32 | remaining bytes)
    uint32_t* nlist;   // LDS counter
    uint32_t n0;       // number of values of this Dac (positions are < n0)
    uint32_t n1;       // number of values with more than one byte (list entries are < n1)
    uint32_t inst;     // for diagnostics
    uint32_t code;     // guard code base
    uint8_t* plane1;         // two-pass mode: where byte 1 of the r-th long value goes
    const uint32_t* pref;    // two-pass mode: per-word rank prefixes of bm0
};
// Emission modes.
//   EM_LIST : one pass; plane-0 bytes are stored, values longer than a byte are appended to the overflow list and
//             placed by dac_finish (any number of planes).
//   EM_P0   : for the few nodes of heights >= 3 of a two-plane Dac: plane-0 byte and continuation bit now; the second
//             byte is placed later, once the bitmap's rank prefixes exist, at rank1(continuation, pos) -- the decoder's
//             own hop (dac.rs:83-90).
//   EM_ONE  : one pass, for the same two-plane Dacs, when the caller already knows `lpos` = that rank (phase 1
//             counted the long values per level and a scan turned the counts into positions): plane-0 bytes,
//             continuation bits and second bytes are all stored at once.
enum : int { EM_LIST = 0, EM_P0 = 1, EM_ONE = 3 };
template <int V>
struct EmTag {
    static constexpr int value = V;
};

// global-memory byte store (the pointers travel through structs, so the compiler would otherwise have to
// emit flat_store_byte)
K2R_HD void gstore8(uint8_t* p, uint8_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    *(__attribute__((address_space(1))) uint8_t*)p = v;
#else
    *p = v;
#endif
}
K2R_HD void gstore32u(uint8_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
    *(__attribute__((address_space(1))) u32_unaligned*)p = v;
#else
    __builtin_memcpy(p, &v, 4);
#endif
}
// stash records that overflowed to global scratch: N consecutive 32-bit words (4-byte aligned)
K2R_HD void gstore_words3(uint32_t* p, uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t u3 __attribute__((ext_vector_type(3), aligned(4)));
    *(__attribute__((address_space(1))) u3*)p = u3{a, b, c};
#else
    p[0] = a; p[1] = b; p[2] = c;
#endif
}
K2R_HD uint64_t gload_u64(const uint64_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *(__attribute__((address_space(1))) const uint64_t*)p;
#else
    return *p;
#endif
}
K2R_HD void gstore_u64(uint64_t* p, uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    *(__attribute__((address_space(1))) uint64_t*)p = v;
#else
    *p = v;
#endif
}
K2R_HD void gstore_words5(uint32_t* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t u4 __attribute__((ext_vector_type(4), aligned(4)));
    *(__attribute__((address_space(1))) u4*)p = u4{a, b, c, d};
    *(__attribute__((address_space(1))) uint32_t*)(p + 4) = e;
#else
    p[0] = a; p[1] = b; p[2] = c; p[3] = d; p[4] = e;
#endif
}
template <int N>
K2R_HD void gload_words(const uint32_t* p, uint32_t (&w)[N]) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(N == 3 || N == 5, "record sizes");
    if (N == 3) {
        typedef uint32_t u3 __attribute__((ext_vector_type(3), aligned(4)));
        const u3 v = *(__attribute__((address_space(1))) const u3*)p;
        w[0] = v.x; w[1] = v.y; w[2] = v.z;
    } else {
        typedef uint32_t u4 __attribute__((ext_vector_type(4), aligned(4)));
        const u4 v = *(__attribute__((address_space(1))) const u4*)p;
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        w[N - 1] = *(__attribute__((address_space(1))) const uint32_t*)(p + 4);
    }
#else
    for (int i = 0; i < N; i++) w[i] = p[i];
#endif
}
// WHICH = 0: Lmax Dac (sh.bmV0, sh.nlistV); 1: Lmin Dac (sh.bmM[0], sh.nlistM)
template <int WHICH, int MODE = EM_LIST, class EX>
K2R_HD void emit_val(EX& ex, const DacSink& d, uint32_t pos, uint32_t zz, int tid, uint32_t lpos = 0) {
    pos = guard_pos(ex, pos, 1, d.n0, d.code);
    uint32_t* const bm0 = WHICH ? ex.sh.bmM[0] : ex.sh.bmV0;
    gstore8(d.plane0 + pos, (uint8_t)zz);
    if (zz > 0xffu) {
        bm_set(ex, bm0, pos);
        if (MODE == EM_LIST) {
            const uint32_t slot = ex.lds_add(WHICH ? &ex.sh.nlistM : &ex.sh.nlistV, 1u);
            d.list[guard_pos(ex, slot, 1, d.n1, d.code + 1)] = ((uint64_t)pos << 32) | (uint64_t)(zz >> 8);
        }
        if (MODE == EM_ONE) gstore8(d.plane1 + guard_pos(ex, lpos, 1, d.n1, d.code + 1), (uint8_t)(zz >> 8));
    }
}

// Four values at consecutive positions pos..pos+3 (the children of one internal node are always adjacent in
// level order): one unaligned 4-byte store of the plane-0 bytes; values longer than a byte take the slow path.
template <int WHICH, int MODE = EM_LIST, class EX>
K2R_HD void emit4(EX& ex, const DacSink& d, uint32_t pos, uint32_t z0, uint32_t z1, uint32_t z2, uint32_t z3, int tid,
                  uint32_t lpos = 0) {
    pos = guard_pos(ex, pos, 4, d.n0, d.code);
    uint32_t* const bm0 = WHICH ? ex.sh.bmM[0] : ex.sh.bmV0;
    gstore32u(d.plane0 + pos, (z0 & 0xffu) | ((z1 & 0xffu) << 8) | ((z2 & 0xffu) << 16) | (z3 << 24));
    if ((z0 | z1 | z2 | z3) > 0xffu) {
        if (MODE == EM_P0 || MODE == EM_ONE) {
            const uint32_t run = (z0 > 0xffu ? 8u : 0u) | (z1 > 0xffu ? 4u : 0u) | (z2 > 0xffu ? 2u : 0u) | (z3 > 0xffu ? 1u : 0u);
            bm_or_run(ex, bm0, pos, 4, run);
            if (MODE == EM_ONE) {
                const uint32_t z[4] = {z0, z1, z2, z3};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (z[i] > 0xffu) {
                        gstore8(d.plane1 + guard_pos(ex, lpos, 1, d.n1, d.code + 1), (uint8_t)(z[i] >> 8));
                        lpos++;
                    }
                }
            }
        } else {
            const uint32_t z[4] = {z0, z1, z2, z3};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (z[i] > 0xffu) {
                    bm_set(ex, bm0, pos + i);
                    const uint32_t slot = ex.lds_add(WHICH ? &ex.sh.nlistM : &ex.sh.nlistV, 1u);
                    d.list[guard_pos(ex, slot, 1, d.n1, d.code + 1)] = ((uint64_t)(pos + i) << 32) | (uint64_t)(z[i] >> 8);
                }
            }
        }
    }
}

// The BitMap of `nbits` zero bits (bitmap.rs:128-138): the last level of a Dac never continues.
template <class C, class EX>
K2R_HD void bitmap_write_zero(EX& ex, uint32_t nbits, uint8_t* dst) {
    const uint32_t nw = 2u + nbits / 128u + (nbits + 31u) / 32u;  // header, rank index, words
    ex.par_nosync([&](int tid, EncRegs&) {
        for (uint32_t w = (uint32_t)tid; w < nw; w += C::NT)
            gstore32u(dst + 4 * w, w == 0 ? __builtin_bswap32(nbits) : (w == 1 ? __builtin_bswap32(4u) : 0u));
    });
}

// Planes 1..nlev-1 of one Dac from its overflow list; also writes every level's BitMap.
// bmA = continuation bitmap of plane 0 (already complete), bmB = scratch for the next plane.
template <class C, class EX>
K2R_HD void dac_finish(EX& ex, const DacLayout& L, uint8_t* inst_out, uint32_t* bmA, uint32_t* bmB, uint32_t* pref,
                       uint64_t* list, const uint32_t* nlist_p) {
    constexpr int NT = C::NT;
    uint32_t* cur = bmA;
    uint32_t* nxt = bmB;
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
        if (j >= L.nlev) break;
        bitmap_finish_write<C>(ex, cur, L.n[j], pref, inst_out + L.bm_off[j]);
        if (j + 1 >= L.nlev) break;
        const uint32_t Wn = (L.n[j + 1] + 31) / 32;
        ex.par([&](int tid, EncRegs&) {
            for (uint32_t w = (uint32_t)tid; w <= Wn; w += NT) nxt[w] = 0;
        });
        ex.barrier_global();  // the overflow list lives in global scratch and was appended to by other threads
        const uint32_t nl = ex.uni(*nlist_p);
        uint8_t* plane = inst_out + L.by_off[j + 1];
        ex.par([&](int tid, EncRegs&) {
            // four entries per trip, their loads issued together: with one load in flight per thread the loop ran at the memory
            // latency (1300 cycles per trip, 85 trips per plane of an all-Snapshot instant)
            constexpr uint32_t U = 4;
            for (uint32_t e0 = (uint32_t)tid; e0 < nl; e0 += U * NT) {
                uint64_t ents[U];
#pragma unroll
                for (uint32_t u = 0; u < U; u++) ents[u] = e0 + u * NT < nl ? gload_u64(list + e0 + u * NT) : 0;
#pragma unroll
                for (uint32_t u = 0; u < U; u++) {
                    const uint64_t ent = ents[u];
                    const uint32_t rem = (uint32_t)ent;
                    if (rem == 0) continue;  // value ended on an earlier plane (or no entry)
                    const uint32_t pos = (uint32_t)(ent >> 32);
                    const uint32_t posg = guard_pos(ex, pos, 1, L.n[j], kGuardListPos + j);
                    const uint32_t q = guard_pos(ex, bm_rank(cur, pref, posg), 1, L.n[j + 1], kGuardListRank + j);  // dac.rs:86
                    gstore8(plane + q, (uint8_t)rem);
                    const uint32_t rest = rem >> 8;
                    if (rest) bm_set(ex, nxt, q);
                    gstore_u64(list + e0 + u * NT, ((uint64_t)q << 32) | (uint64_t)rest);
                }
            }
            guard_flush(ex);
        });
        uint32_t* t = cur;
        cur = nxt;
        nxt = t;
    }
}


// ======================================================================================================
// The chunk encoder.  PADDED: rows or cols < sidelen (or not a multiple of 8 blocks).  VEC: int32 (1), float32 (2) or
// int64 (3) / float64 (4) input with unit column stride and 16-byte aligned rows (=> vector loads), implies !PADDED.
// ======================================================================================================
template <class C, bool PADDED, int VEC, class EX>
K2R_HD void encode_chunk(EX& ex, const TileArgs& ta, TileResult* res, uint64_t* listV, uint64_t* listM, uint32_t* scmp, uint32_t* ovf) {
    constexpr int H = C::H;
    constexpr int NT = C::NT;
    auto& sh = ex.sh;
    using SH = EncShared<C>;
    static_assert(!(PADDED && VEC != 0), "vector loads need an unpadded tile");
    // words of the LDS pool the stash may use (ta.stash_words: 0 = all of it; tests shrink it to force the fallback)
    // ta.stash_words (tests, A/B runs): an instant whose records need more words than this takes the re-reading fallback
    // passes instead of the stash; 0 = never (what does not fit the LDS pool overflows to global scratch: EncPool)
    const uint32_t stash_cap = ta.stash_words != 0 ? ta.stash_words : 0xffffffffu;
    uint32_t* const ovI = ovf;                    // I records number CAPI_REC and up
    uint32_t* const ovQ = ovf + SH::OVI_WORDS;    // Q records number CAPQ_REC and up

    uint8_t* const out = ta.out;
    const uint64_t cap = ta.out_cap;

    // ---- geometry of an in-thread / top node ---------------------------------------------------------
    auto blk_origin = [&](int tid, uint32_t& r0, uint32_t& c0) {
        uint32_t br, bc;
        morton_decode((uint32_t)tid, br, bc);
        r0 = br * 8;
        c0 = bc * 8;
#if defined(__HIP_DEVICE_COMPILE__)
        // Opaque to the optimizer on purpose: everything derived from the block origin (32 load addresses per
        // instant) is loop-invariant across instants, and LICM would otherwise keep it all live in registers for the
        // whole chunk -- i.e. spill it to scratch and reload it every instant.  Recomputing costs a few VALU ops.
        asm volatile("" : "+v"(r0), "+v"(c0));
#endif
    };
    // all-invalid test of the node whose top-left cell is (r,c)  (snapshot.rs:453-457)
    auto inval = [&](uint32_t r, uint32_t c) -> bool { return PADDED && (r >= ta.rows || c >= ta.cols); };
    auto top_inval = [&](int h, uint32_t j) -> bool {
        if (!PADDED) return false;
        uint32_t br, bc;
        morton_decode(j, br, bc);
        return inval(br << h, bc << h);
    };

    // Speculative parts (near-linear scaling when a GPU holds few chunks per CU): a chunk's instants may be encoded by several
    // work items.  The first, [0, m), is an ordinary encode of those instants -- except that it leaves instant 0's compact
    // snapshot copy in a buffer of the CHUNK (ta.shared_cmp) instead of this workgroup's scratch, and says so (ta.shared_flag).
    // Every other part, [m, m'), is a CONTINUATION: it assumes what holds for 31 instants in 32 on the benchmark's data -- that
    // no block boundary falls into [1, m), so that instant 0 is still the open block's snapshot and the block holds m instants
    // -- waits for that copy, and runs chunk.rs:55-74 from instant m on, writing its Logs and Blocks from byte 0 of its own
    // slot.  k_stitch checks the assumption against the earlier parts' results (one snapshot in the first, none in the
    // middle ones) and splices the bytes; otherwise the chunk is re-encoded whole.  The work queue holds every first part
    // before any continuation, so a continuation's wait is for a workgroup that is already running.
    const uint32_t i_begin = ta.inst_begin, i_end = ta.inst_end != 0 ? ta.inst_end : ta.instants;
    const bool cont = i_begin > 0;
    const bool head = !cont && ta.shared_flag != nullptr;  // the first of several parts
    bool published = false;
    uint32_t* cmp = scmp;  // where the open block's compact snapshot copy lives
    constexpr uint32_t NO_HDR = 0xffffffffu;  // the open block's count byte lives in the first half's bytes
    uint32_t carry = 0;
    uint32_t off = cont ? 0u : 6u;            // chunk header: encoding, fractional_bits, n_blocks (chunk.rs:236-238)
    uint32_t n_blocks = 0;
    uint32_t blk_hdr = cont ? NO_HDR : 6u;    // where the open block's n_instants byte goes (block.rs:89)
    uint32_t blk_count = 0;      // instants in the open block
    uint32_t s_idx = 0;          // instant of the open block's snapshot (chunk.rs:52)
    // The open block's snapshot instant is compared with every later instant of the block, so it is read 30 times as
    // often as anything else.  When the instant's range fits 16 bits, a compact copy (cell - minimum as uint16, laid out
    // [sub-block][thread][16 cells] so that a wave reads 2 KB contiguous) is kept in global scratch and read instead:
    // 2 bytes per cell whatever the input type, already converted to fixed point.
    bool s_cmp = false;
    int32_t s_base = 0;
    uint32_t n_snap = 0, n_log = 0, n_stash = 0;
    int32_t status = ST_OK;

    ex.par([&](int tid, EncRegs&) {
        if (tid == 0) {
            sh.err = 0;
            sh.stI = 0;
            sh.stQ = 0;
            for (int i = 0; i < 6; i++) sh.fault[i] = 0;
            for (int i = 0; i < NPROF; i++) sh.prof[i] = 0;
            if (cap >= 6 && !cont) {
                out[0] = (uint8_t)ta.dtype;
                out[1] = (uint8_t)ta.fbits;
            }
        }
    });
    if (cap < 7 && !cont) status = ST_OUT_CAPACITY;
#ifdef K2R_PROFILE
    if (!EX::kSim) {
        ex.par([&](int tid, EncRegs&) {
            if (tid == 0) sh.prof_last = clock64();
            if ((tid & 63) == 0)
                for (int k = 0; k < NPROF; k++) sh.pw[tid >> 6][k][0] = sh.pw[tid >> 6][k][1] = 0;
        });
        ex.pw_start();
    }
#endif

    // the compact copy of instant `from` (the open block's snapshot) for the logs that follow
#ifdef K2R_REGCOPY
    constexpr bool kRegCopy = !PADDED;
#else
    constexpr bool kRegCopy = false;
#endif
    // sub-block (of a thread's phase-1 pass p) -> height-2 node and owner block, in the mapping of the lean phase 1
    [[maybe_unused]] auto lean_node = [&](uint32_t ltid, int p, uint32_t& B, uint32_t& j) {
        if (EX::kSim) {
            B = ltid;
            j = (uint32_t)p;
        } else {
            constexpr uint32_t NTW = NT < 64 ? NT : 64, BPP = NTW / 4;  // lanes per wave, blocks per pass and wave
            const uint32_t lane = ltid & 63u;
            B = (ltid & ~63u) + BPP * (uint32_t)p + (lane >> 2);
            j = lane & 3u;
        }
    };
    auto compact_pass = [&](uint32_t from, bool to_global) {
        if (kRegCopy) {
#ifdef K2R_REGCOPY
            ex.par_nosync([&](int tid, EncRegs& r) {
                const uint32_t ltid = opaque((uint32_t)tid);
                int32_t lerr = 0;
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    uint32_t B, j, br, bc;
                    lean_node(ltid, p, B, j);
                    morton_decode(4u * B + j, br, bc);
                    int32_t t16[16];
                    load_sub16<false, VEC>(ta, from, 4u * br, 4u * bc, 0, t16, lerr);
#pragma unroll
                    for (int q = 0; q < 2; q++)
#pragma unroll
                        for (int c = 0; c < 4; c++)
                            r.cw[8 * p + 4 * q + c] = (uint32_t)(t16[8 * q + c] - s_base) | ((uint32_t)(t16[8 * q + 4 + c] - s_base) << 16);
                    if (to_global) {  // (the first part of a chunk encoded in parts: the continuations fetch their copy from here)
                        uint32_t* g = cmp + (size_t)compact_slot<C>((int)B, (int)j) * 8;
                        for (int i = 0; i < 8; i++) g[i] = r.cw[8 * p + i];
                    }
                }
            });
            if (to_global) ex.barrier_global();
#endif
        } else {
            ex.par_nosync([&](int tid, EncRegs&) {
                uint32_t r0, c0;
                blk_origin(tid, r0, c0);
                int32_t lerr = 0;
#pragma unroll 1
                for (int j = 0; j < 4; j++) {
                    int32_t t16[16];
                    load_sub16<PADDED, VEC>(ta, from, r0, c0, j, t16, lerr);
                    store_compact<C>(cmp, tid, j, s_base, t16);
                }
            });
            ex.barrier_global();  // the stores are read back (by the same threads) in the next instant's phase 1
        }
    };

    if (cont) {  // the open block as the first part left it: instant 0 is its snapshot
        const uint32_t f = ex.await(ta.shared_flag, PART_FAILED);
        if (f == PART_FAILED) status = ST_RESPLIT;
        s_cmp = f == PART_CMP;
        s_base = (int32_t)ex.flag_payload(ta.shared_flag);
        s_idx = 0;
        blk_count = i_begin;
        cmp = ta.shared_cmp;
        ex.par([&](int tid, EncRegs& r) {
            if (tid == 0) sh.err = 0;
#ifdef K2R_REGCOPY
            if (kRegCopy && s_cmp) {
                const uint32_t ltid = opaque((uint32_t)tid);
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    uint32_t B, j, w[8];
                    lean_node(ltid, p, B, j);
                    load_compact_raw<C>(cmp, (int)B, (int)j, w);
#pragma unroll
                    for (int i = 0; i < 8; i++) r.cw[8 * p + i] = w[i];
                }
            }
#else
            (void)r;
#endif
        });
    }

    for (uint32_t inst = i_begin; inst < i_end && status == ST_OK; inst++) {
        const bool have_s = inst > 0;
        // ================= phase 1: stream the tile in 4x4 sub-blocks; thread-local counts ==============
        // Nothing but four per-height-2 summaries survives this phase in registers: cells are re-read on
        // demand at emission, and only under visited subtrees (sparse for logs).
        //
        // Two forms.  The LEAN one runs whenever the instant is compared with a compact snapshot copy on an unpadded tile
        // (every log of the benchmark): straight-line code on the copy's packed 16-bit cells -- differences, extremes and
        // node values are all taken relative to the copy's base (the constant is removed from the few values that are
        // emitted, EncShared::lq / rec_corr) -- with ONE stash allocation per wave and sub-block (ex.stash_alloc) instead
        // of one LDS atomic per record inside a divergent branch.  The general form covers the first instant of a block
        // (no snapshot to compare with), snapshots beyond 16 bits of range and padded tiles.
#ifndef K2R_NO_LEAN
        const bool lean = !PADDED && have_s && s_cmp;
#else
        const bool lean = false;
#endif
        // -- the analysis of one 4x4 sub-block (a height-2 node): t = its 16 cells in local Morton order, w = the copy's --
        struct LeanSub {
            uint32_t pq[4];    // per quad: internal for the log (0 / 1)
            uint32_t recw[4];  // per quad: Lmax | Lmin << 16 of the log (log.rs:133,148), both s_base too large
            uint32_t qw[8];    // per quad: its four cell differences as two packed words (+ s_base each)
            uint32_t tb1, e4, su, nq;  // T bits, "equal" bits (first quad = bit 3), non-uniform quads, internal quads
            uint32_t ne2, nu2, p2;     // the node: not "equal", not uniform, internal for the log (0 / 1 each)
            uint32_t d2w;              // the node's own Lmax | Lmin << 16 (+ s_base)
            uint32_t smn2, smx2;       // the snapshot's extremes over the node, as offsets from s_base
            int32_t mn2, mx2, d0;      // the instant's extremes; the first cell's difference (+ s_base)
        };
        auto lean_analyse = [&](const int32_t (&t)[16], const uint32_t (&w)[8], LeanSub& o) {
            // extremes of the instant's quads and of the node (snapshot.rs:476-497)
            int32_t mn1[4], mx1[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                mn1[q] = min4(t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]);
                mx1[q] = max4(t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]);
            }
            o.mn2 = min4(mn1[0], mn1[1], mn1[2], mn1[3]);
            o.mx2 = max4(mx1[0], mx1[1], mx1[2], mx1[3]);
            // the same for the snapshot, two quads per packed operation: sn[p] = minima of quads 2p | 2p + 1 << 16
            uint32_t sn[2], sx[2];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                sn[p] = pk_min_u16(pk_min_u16(w[4 * p], w[4 * p + 1]), pk_min_u16(w[4 * p + 2], w[4 * p + 3]));
                sx[p] = pk_max_u16(pk_max_u16(w[4 * p], w[4 * p + 1]), pk_max_u16(w[4 * p + 2], w[4 * p + 3]));
            }
            const uint32_t sn2p = pk_min_u16(sn[0], sn[1]), sx2p = pk_max_u16(sx[0], sx[1]);
            // (opaque: the reductions happen here, not at the end of the block with their inputs spilled in between)
            o.smn2 = opaque((sn2p & 0xffffu) < (sn2p >> 16) ? (sn2p & 0xffffu) : (sn2p >> 16));
            o.smx2 = opaque((sx2p & 0xffffu) > (sx2p >> 16) ? (sx2p & 0xffffu) : (sx2p >> 16));
            // cell differences against the copy (all of them s_base too large: equality tests do not care)
            int32_t d[16];
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const uint32_t ww = w[4 * (q >> 1) + c];
                    d[4 * q + c] = t[4 * q + c] - (int32_t)((q & 1) ? (ww >> 16) : (ww & 0xffffu));
                }
            // per quad: "all four differences equal" <=> x == 0 (log.rs:780,805); internal <=> not uniform and not equal.
            // Flags are 0 / 1 integers made without compares (nz): no mask registers, nothing for the compiler to re-derive.
            uint32_t x[4];
            o.tb1 = 0;
            o.e4 = 0;
            o.su = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                x[q] = (uint32_t)((d[4 * q] ^ d[4 * q + 1]) | (d[4 * q] ^ d[4 * q + 2]) | (d[4 * q] ^ d[4 * q + 3]));
                const uint32_t u1 = nz((uint32_t)(mn1[q] ^ mx1[q]));
                o.pq[q] = u1 & nz(x[q]);               // log.rs:137-152
                o.tb1 = (o.tb1 << 1) | o.pq[q];
                o.e4 = (o.e4 << 1) | (u1 ^ o.pq[q]);  // T = 0 and not uniform: "equal" (log.rs:137-144)
                o.su += u1;
                const uint32_t sxq = (q & 1) ? (sx[q >> 1] >> 16) : (sx[q >> 1] & 0xffffu);
                const uint32_t snq = (q & 1) ? (sn[q >> 1] >> 16) : (sn[q >> 1] & 0xffffu);
                o.recw[q] = pk_lo16((uint32_t)(mx1[q] - (int32_t)sxq), (uint32_t)(mn1[q] - (int32_t)snq));  // log.rs:133,148
                o.qw[2 * q] = pk_lo16((uint32_t)d[4 * q], (uint32_t)d[4 * q + 1]);
                o.qw[2 * q + 1] = pk_lo16((uint32_t)d[4 * q + 2], (uint32_t)d[4 * q + 3]);
            }
            const uint32_t X = x[0] | x[1] | x[2] | x[3] | (uint32_t)((d[0] ^ d[4]) | (d[0] ^ d[8]) | (d[0] ^ d[12]));
            // (flags through nz: "== 0" of a long OR-sum is rewritten by the optimizer into a conjunction of compares over every
            // term, all of them kept alive -- spilled -- until the end of the block)
            o.ne2 = nz(X);
            o.nu2 = nz((uint32_t)(o.mn2 ^ o.mx2));
            o.p2 = o.ne2 & o.nu2;  // log.rs:137-152 for the node
            o.nq = popc32(o.tb1);
            o.d0 = d[0];
            o.d2w = pk_lo16((uint32_t)(o.mx2 - (int32_t)o.smx2), (uint32_t)(o.mn2 - (int32_t)o.smn2));  // log.rs:133,148
        };
        // -- its stash records: Q = owner, ordinal among the owner's internal quads, the four cell differences; I = owner, ordinals,
        //    T bits, "equal" bits and the Lmax|Lmin pairs of the four quads.  ONE allocation for the whole wave. --
        auto lean_put = [&](const LeanSub& o, uint32_t owner, uint32_t lI1b, uint32_t lI2b) {
            uint32_t slotQ, slotI, endQ, endI;
            ex.stash_alloc(o.nq | (o.p2 << 16), &sh.stQ, &sh.stI, slotQ, slotI, endQ, endI);
            uint32_t ord = lI1b;
            const uint32_t p0 = owner | (lI2b << 10) | (lI1b << 12) | (o.tb1 << 16) | (o.e4 << 20);
            if (endQ <= (uint32_t)SH::CAPQ_REC && endI <= (uint32_t)SH::CAPI_REC) {  // (wave-uniform) nothing of this wave overflows
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint32_t* d = sh.pool + ((uint32_t)SH::QBASE + mul24(slotQ, 3u));
                    if (o.pq[q]) {
                        d[0] = owner | (ord << 10);
                        d[1] = o.qw[2 * q];
                        d[2] = o.qw[2 * q + 1];
                    }
                    slotQ += o.pq[q];
                    ord += o.pq[q];
                }
                if (o.p2) {
                    uint32_t* d = sh.pool + mul24(slotI, 5u);
                    d[0] = p0;
                    d[1] = o.recw[0];
                    d[2] = o.recw[1];
                    d[3] = o.recw[2];
                    d[4] = o.recw[3];
                }
            } else {  // (separate code paths on purpose: an LDS-or-global pointer would turn both into FLAT stores)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (o.pq[q]) {
                        if (slotQ < (uint32_t)SH::CAPQ_REC) {
                            uint32_t* d = sh.pool + (SH::QBASE + 3u * slotQ);
                            d[0] = owner | (ord << 10);
                            d[1] = o.qw[2 * q];
                            d[2] = o.qw[2 * q + 1];
                        } else {
                            gstore_words3(ovQ + 3u * (slotQ - (uint32_t)SH::CAPQ_REC), owner | (ord << 10), o.qw[2 * q], o.qw[2 * q + 1]);
                        }
                    }
                    slotQ += o.pq[q];
                    ord += o.pq[q];
                }
                if (o.p2) {
                    if (slotI < (uint32_t)SH::CAPI_REC) {
                        uint32_t* d = sh.pool + 5u * slotI;
                        d[0] = p0;
                        d[1] = o.recw[0];
                        d[2] = o.recw[1];
                        d[3] = o.recw[2];
                        d[4] = o.recw[3];
                    } else {
                        gstore_words5(ovI + 5u * (slotI - (uint32_t)SH::CAPI_REC), p0, o.recw[0], o.recw[1], o.recw[2], o.recw[3]);
                    }
                }
            }
        };
        // what the owner of a block keeps of it, and what goes to the top arrays
        auto lean_block_out = [&](uint32_t B, int32_t mn3, int32_t mx3, uint32_t smn3, uint32_t smx3, int32_t dref, uint32_t Zb, uint32_t& wide,
                                  int32_t& err, bool store) {
            // raw int32 rows are not range-checked cell by cell: the block's extremes decide (value-range contract)
            if (VEC == 1 && (mn3 < -VALUE_LIMIT || mx3 >= VALUE_LIMIT) && err == 0) err = ERR_RANGE;
            const int32_t smn3t = s_base + (int32_t)smn3, smx3t = s_base + (int32_t)smx3;
            // every in-block log value lies in [mn3 - smx3, mx3 - smn3]: below 2^15 in magnitude none of them needs a
            // third byte (and the 16 bits kept of each are all of it); otherwise the exact classes_pass is requested
            wide = (mn3 - smx3t < -32768 || mx3 - smn3t > 32767) ? 1u : 0u;
            if (store) {
                sh.tmin[B] = mn3;
                sh.tmax[B] = mx3;
                sh.smin[B] = smn3t;
                sh.smax[B] = smx3t;
                sh.diff[B] = dref - s_base;
                sh.eq[B] = Zb == 0 ? 1u : 0u;
                sh.lq[B][0] = 0;
                sh.lq[B][1] = 0;
                sh.li[B] = 0;
            }
        };
        if (lean && !EX::kSim) {
            // The GPU form.  A lane owns a SUB-BLOCK: in pass p = 0..3 the wave covers 64 Morton-consecutive height-2 nodes, lanes
            // 4b .. 4b+3 the four of one block -- so eight lanes read 128 contiguous bytes of a row (a lane that owns an 8x8 block
            // reads 16 bytes of every other 32, and each line is fetched twice: profiles/r04d), and the wave's records enter the
            // stash in level order, which is what lets the emission passes' stores coalesce (profiles/r04e: 16 instead of up to 350
            // cycles per store instruction).  Block-level results are reduced over the quad with DPP and pulled by the lane that
            // owns the block in the rest of the kernel (thread = block) with ds_bpermute.
            if constexpr (!EX::kSim) ex.par([&](int tid, EncRegs& r) {
                constexpr uint32_t NTW = NT < 64 ? NT : 64, BPP = NTW / 4;  // lanes per wave, blocks per pass and wave
                const uint32_t ltid = opaque((uint32_t)tid), lane = ltid & 63u, j = lane & 3u;
                int32_t err = 0;
                // -DK2R_LEAN_PREFETCH=1: the next pass's cells AND copy are requested before this pass's are analysed; =2: the cells
                // only (16 instead of 24 registers in flight across the analysis; the copy is an L2 hit, the cells come from HBM)
                auto request_t = [&](int p, int32_t (&tt)[16]) {
                    const uint32_t Bp = (ltid & ~63u) + BPP * (uint32_t)p + (lane >> 2);
                    uint32_t br, bc;
                    morton_decode(4u * Bp + j, br, bc);
                    load_sub16<false, VEC>(ta, inst, 4u * br, 4u * bc, 0, tt, err);
                    // (settled here: left symbolic, the range / conversion checks of all four passes are evaluated at the end of
                    // the phase, with every raw cell they look at kept alive -- spilled -- until then)
                    if (VEC >= 2) err = (int32_t)opaque((uint32_t)err);
                };
                auto request_w = [&](int p, uint32_t (&ww)[8]) {
                    const uint32_t Bp = (ltid & ~63u) + BPP * (uint32_t)p + (lane >> 2);
#ifdef K2R_REGCOPY
#pragma unroll
                    for (int i = 0; i < 8; i++) ww[i] = r.cw[8 * p + i];
                    (void)Bp;
#else
                    load_compact_raw<C>(cmp, (int)Bp, (int)j, ww);
#endif
                };
#ifdef K2R_LEAN_PREFETCH
                int32_t tn[16];
                uint32_t wn[8];
                request_t(0, tn);
                if (K2R_LEAN_PREFETCH == 1) request_w(0, wn);
#endif
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    sched_fence();
                    const uint32_t B = (ltid & ~63u) + BPP * (uint32_t)p + (lane >> 2);  // the block, and its owner's thread index
                    int32_t t[16];
                    uint32_t w[8];
#ifdef K2R_LEAN_PREFETCH
                    if (K2R_LEAN_PREFETCH != 1) request_w(p, w);
#pragma unroll
                    for (int i = 0; i < 16; i++) t[i] = tn[i];
                    if (K2R_LEAN_PREFETCH == 1) {
#pragma unroll
                        for (int i = 0; i < 8; i++) w[i] = wn[i];
                    }
                    if (p < 3) {
                        request_t(p + 1, tn);
                        if (K2R_LEAN_PREFETCH == 1) request_w(p + 1, wn);
                    }
                    sched_fence();
#else
                    request_t(p, t);
                    request_w(p, w);
#endif
                    LeanSub o;
                    lean_analyse(t, w, o);
                    sched_fence();
                    // the block: counts of the sub-blocks before this one, extremes, "equal"
                    const uint32_t cnt = o.nq | (o.p2 << 8) | (o.su << 16) | (o.nu2 << 24);
                    const uint32_t n0 = EX::template quad_bcast<0>(cnt), n1 = EX::template quad_bcast<1>(cnt), n2 = EX::template quad_bcast<2>(cnt),
                                   n3 = EX::template quad_bcast<3>(cnt);
                    const uint32_t pre = (j > 0 ? n0 : 0u) + (j > 1 ? n1 : 0u) + (j > 2 ? n2 : 0u), tot = n0 + n1 + n2 + n3;
                    lean_put(o, B, pre & 0xffu, (pre >> 8) & 0xffu);
                    const int32_t mn3 = EX::quad_min(o.mn2), mx3 = EX::quad_max(o.mx2);
                    const uint32_t smn3 = (uint32_t)EX::quad_min((int32_t)o.smn2), smx3 = (uint32_t)EX::quad_max((int32_t)o.smx2);
                    const int32_t dref = (int32_t)EX::template quad_bcast<0>((uint32_t)o.d0);
                    const uint32_t Zb = EX::quad_or(o.ne2 | nz((uint32_t)(o.d0 ^ dref)));
                    uint32_t wide;
                    lean_block_out(B, mn3, mx3, smn3, smx3, dref, Zb, wide, err, j == 0);
                    // the owner's registers (k2r EncRegs): flags = eq2 bits | internal quads per sub-block (snapshot, log) | wide
                    const uint32_t fw = EX::quad_or(((o.ne2 ^ 1u) << j) | (o.su << (4u + 3u * j)) | (o.nq << (16u + 3u * j))) | (wide << 28);
                    const uint32_t uw = EX::quad_or((o.nu2 ^ 1u) << j);
                    const uint32_t own_lane = lane - BPP * (uint32_t)p, src = (4u * own_lane) & 63u;  // (meaningful for the owners of this pass)
                    const uint32_t g0 = EX::lane_pull(src, o.d2w), g1 = EX::lane_pull(src + 1u, o.d2w), g2 = EX::lane_pull(src + 2u, o.d2w),
                                   g3 = EX::lane_pull(src + 3u, o.d2w);
                    const uint32_t gf = EX::lane_pull(src, fw), gu = EX::lane_pull(src, uw), gt = EX::lane_pull(src, tot);
                    if (lane / BPP == (uint32_t)p) {
                        r.d2[0] = g0;
                        r.d2[1] = g1;
                        r.d2[2] = g2;
                        r.d2[3] = g3;
                        r.flags = gf;
                        r.u2 = gu;
                        r.sc[0] = ((gt >> 16) & 0xffu) | ((gt >> 24) << 16);   // snapshot I1 | I2 << 16
                        r.sc[3] = (gt & 0xffu) | (((gt >> 8) & 0xffu) << 16);  // log I1 | I2 << 16
                    }
                }
                if (tid < C::TBW) {
                    sh.tbS[tid] = 0;
                    sh.tbL[tid] = 0;
                }
                if (err != 0) ex.lds_min(&sh.err, err);
            });
        } else if (lean) ex.par([&](int tid, EncRegs& r) {  // the same with thread = block (the sequential context of tests/sim)
            uint32_t r0, c0;
            blk_origin(tid, r0, c0);
            int32_t err = 0;
            uint32_t sI1 = 0, sI2 = 0, lI1 = 0, lI2 = 0, eqbits = 0, cntbits = 0, u2 = 0, Z = 0;
            int32_t mn3 = 0, mx3 = 0, dref = 0;
            uint32_t smn3 = 0, smx3 = 0;  // the snapshot's extremes as offsets from s_base, like the copy's cells
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int32_t t[16];
                uint32_t w[8];
                load_sub16<false, VEC>(ta, inst, r0, c0, j, t, err);
#ifdef K2R_REGCOPY
                for (int i = 0; i < 8; i++) w[i] = r.cw[8 * j + i];
#else
                load_compact_raw<C>(cmp, tid, j, w);
#endif
                LeanSub o;
                lean_analyse(t, w, o);
                if (j == 0) dref = o.d0;
                Z |= o.ne2 | nz((uint32_t)(o.d0 ^ dref));
                lean_put(o, (uint32_t)tid, lI1, lI2);
                sI1 += o.su;
                sI2 += o.nu2;
                cntbits |= (o.su << (4 + 3 * j)) | (o.nq << (16 + 3 * j));
                lI1 += o.nq;
                lI2 += o.p2;
                u2 |= (o.nu2 ^ 1u) << j;
                eqbits |= (o.ne2 ^ 1u) << j;
                r.d2[j] = o.d2w;
                mn3 = j == 0 ? o.mn2 : (o.mn2 < mn3 ? o.mn2 : mn3);
                mx3 = j == 0 ? o.mx2 : (o.mx2 > mx3 ? o.mx2 : mx3);
                smn3 = j == 0 ? o.smn2 : (o.smn2 < smn3 ? o.smn2 : smn3);
                smx3 = j == 0 ? o.smx2 : (o.smx2 > smx3 ? o.smx2 : smx3);
            }
            uint32_t wide;
            lean_block_out((uint32_t)tid, mn3, mx3, smn3, smx3, dref, Z, wide, err, true);
            r.sc[0] = sI1 | (sI2 << 16);
            r.flags = eqbits | cntbits | (wide << 28);
            r.u2 = u2;
            if (tid < C::TBW) {
                sh.tbS[tid] = 0;
                sh.tbL[tid] = 0;
            }
            r.sc[3] = lI1 | (lI2 << 16);
            if (err != 0) ex.lds_min(&sh.err, err);
        });
        else ex.par([&](int tid, EncRegs& r) {
            uint32_t r0, c0;
            blk_origin(tid, r0, c0);
            int32_t err = 0;
            const bool inv3 = inval(r0, c0);
            uint32_t sI1 = 0, sI2 = 0, lI1 = 0, lI2 = 0;
            // (which log values need a second byte is not this phase's business: the pre-pass over the stash records
            // works that out for the few values that are actually emitted, see EncShared::lq)
            if (have_s) {
                sh.lq[tid][0] = 0;
                sh.lq[tid][1] = 0;
                sh.li[tid] = 0;
            }
            int32_t df2_0 = 0, mn3 = 0, mx3 = 0, smn3 = 0, smx3 = 0;
            uint32_t eqbits = 0, eqall = 1, cntbits = 0, wide = 0, u2 = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                sched_fence();
                const uint32_t rj = r0 + 4 * (j >> 1), cj = c0 + 4 * (j & 1);
                const bool inv2 = inval(rj, cj);
                int32_t t16[16];
                load_sub16<PADDED, VEC>(ta, inst, r0, c0, j, t16, err);
                int32_t mn1[4], mx1[4];
                bool inv1[4], P1S[4];
#pragma unroll
                for (int qq = 0; qq < 4; qq++) {
                    mn1[qq] = min4(t16[4 * qq], t16[4 * qq + 1], t16[4 * qq + 2], t16[4 * qq + 3]);
                    mx1[qq] = max4(t16[4 * qq], t16[4 * qq + 1], t16[4 * qq + 2], t16[4 * qq + 3]);
                    inv1[qq] = inval(rj + 2 * (qq >> 1), cj + 2 * (qq & 1));
                    P1S[qq] = !inv1[qq] && mn1[qq] != mx1[qq];
                    sI1 += P1S[qq] ? 1u : 0u;
                    cntbits += (P1S[qq] ? 1u : 0u) << (4 + 3 * j);  // bits 4..15: internal quads per j, snapshot
                }
                const int32_t mn2 = min4(mn1[0], mn1[1], mn1[2], mn1[3]);
                const int32_t mx2 = max4(mx1[0], mx1[1], mx1[2], mx1[3]);
                mn3 = j == 0 ? mn2 : (mn2 < mn3 ? mn2 : mn3);
                mx3 = j == 0 ? mx2 : (mx2 > mx3 ? mx2 : mx3);
                u2 |= ((inv2 || mn2 == mx2) ? 1u : 0u) << j;
                const bool P2S = !inv2 && mn2 != mx2;
                sI2 += P2S ? 1u : 0u;
                // (the snapshot candidate's byte classes are computed lazily: see classes_pass below)
                // ---- log candidate vs. the open block's snapshot (log.rs:112-165, 725-817) ----
                if (have_s) {
                    int32_t s16[16];
                    if (s_cmp) load_compact<C>(cmp, tid, j, s_base, s16);
                    else load_sub16<PADDED, VEC>(ta, s_idx, r0, c0, j, s16, err);
                    int32_t smn1[4], smx1[4], df1[4];
                    bool eq1[4];
                    uint32_t recw[4], tb1 = 0, e4 = 0;  // the I record of this node (see EncShared::pool)
                    const uint32_t pre1 = lI1;
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) {
                        const uint32_t rq = rj + 2 * (qq >> 1), cq = cj + 2 * (qq & 1);
                        int32_t d[4];
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const bool inv0 = inval(rq + (i >> 1), cq + (i & 1));
                            d[i] = inv0 ? 0 : t16[4 * qq + i] - s16[4 * qq + i];  // log.rs:751
                        }
                        smn1[qq] = min4(s16[4 * qq], s16[4 * qq + 1], s16[4 * qq + 2], s16[4 * qq + 3]);
                        smx1[qq] = max4(s16[4 * qq], s16[4 * qq + 1], s16[4 * qq + 2], s16[4 * qq + 3]);
                        eq1[qq] = d[0] == d[1] && d[0] == d[2] && d[0] == d[3];  // log.rs:780,805
                        df1[qq] = d[0];
                        const bool P1L = !inv1[qq] && mn1[qq] != mx1[qq] && !eq1[qq];  // log.rs:137-152
                        const int32_t vx1 = inv1[qq] ? 0 : mx1[qq] - smx1[qq];          // log.rs:133
                        const int32_t vn1 = mn1[qq] - smn1[qq];                          // log.rs:148
                        recw[qq] = ((uint32_t)vx1 & 0xffffu) | ((uint32_t)vn1 << 16);
                        tb1 = (tb1 << 1) | (P1L ? 1u : 0u);
                        // a T = 0 quad has one eqB bit, set iff "equal" rather than uniform (log.rs:137-144); kept per quad here
                        // (bit 3 - qq), packed to a run of the T = 0 quads when the record is emitted
                        e4 = (e4 << 1) | ((!P1L && !(inv1[qq] || mn1[qq] == mx1[qq])) ? 1u : 0u);
                        if (P1L) {  // Q record: owner, ordinal among the owner's internal quads, the four cell diffs
                            const uint32_t m = ex.lds_add(&sh.stQ, 1u);
                            const uint32_t q0 = (uint32_t)tid | (lI1 << 10);
                            const uint32_t q1 = ((uint32_t)d[0] & 0xffffu) | ((uint32_t)d[1] << 16);
                            const uint32_t q2 = ((uint32_t)d[2] & 0xffffu) | ((uint32_t)d[3] << 16);
                            if (m < (uint32_t)SH::CAPQ_REC) {
                                uint32_t* q = sh.pool + (SH::QBASE + 3u * m);
                                q[0] = q0;
                                q[1] = q1;
                                q[2] = q2;
                            } else {  // (separate code paths on purpose: an LDS-or-global pointer would turn both into FLAT stores)
                                gstore_words3(ovQ + 3u * (m - (uint32_t)SH::CAPQ_REC), q0, q1, q2);
                            }
                        }
                        lI1 += P1L ? 1u : 0u;
                        cntbits += (P1L ? 1u : 0u) << (16 + 3 * j);  // bits 16..27: internal quads per j, log
                    }
                    const int32_t smn2 = min4(smn1[0], smn1[1], smn1[2], smn1[3]);
                    const int32_t smx2 = max4(smx1[0], smx1[1], smx1[2], smx1[3]);
                    const bool eq2 = eq1[0] && eq1[1] && eq1[2] && eq1[3] && df1[0] == df1[1] && df1[0] == df1[2] &&
                                     df1[0] == df1[3];
                    if (j == 0) df2_0 = df1[0];
                    eqall &= df1[0] == df2_0 ? 1u : 0u;
                    const bool P2L = !inv2 && mn2 != mx2 && !eq2;
                    if (P2L) {  // I record: owner, ordinals, T / eqB runs and the Lmax|Lmin pairs of the four quads
                        const uint32_t k = ex.lds_add(&sh.stI, 1u);
                        const uint32_t p0 = (uint32_t)tid | (lI2 << 10) | (pre1 << 12) | (tb1 << 16) | (e4 << 20);
                        if (k < (uint32_t)SH::CAPI_REC) {
                            uint32_t* p = sh.pool + 5u * k;
                            p[0] = p0;
                            p[1] = recw[0];
                            p[2] = recw[1];
                            p[3] = recw[2];
                            p[4] = recw[3];
                        } else {
                            gstore_words5(ovI + 5u * (k - (uint32_t)SH::CAPI_REC), p0, recw[0], recw[1], recw[2], recw[3]);
                        }
                    }
                    lI2 += P2L ? 1u : 0u;
                    smn3 = j == 0 ? smn2 : (smn2 < smn3 ? smn2 : smn3);
                    smx3 = j == 0 ? smx2 : (smx2 > smx3 ? smx2 : smx3);
                    r.d2[j] = ((uint32_t)(inv2 ? 0 : mx2 - smx2) & 0xffffu) | ((uint32_t)(mn2 - smn2) << 16);  // log.rs:133,148
                    eqbits |= (eq2 ? 1u : 0u) << j;
                    eqall &= eq2 ? 1u : 0u;
                }
            }
            const bool P3S = !inv3 && mn3 != mx3;
            // raw int32 rows are not range-checked cell by cell: the block's extremes decide (value-range contract)
            if (VEC == 1 && (mn3 < -VALUE_LIMIT || mx3 >= VALUE_LIMIT) && err == 0) err = ERR_RANGE;
            sh.tmin[tid] = mn3;
            sh.tmax[tid] = mx3;
            // reduce words (completed in phase 3): [0] snapshot I1 | I2 << 16, [3] log I1 | I2 << 16,
            //                                      [7] cL0, [8] cL1 | cL2 << 16, [9] mL1 | mL2 << 16
            r.sc[0] = sI1 | (sI2 << 16);
            (void)P3S;
            if (have_s) {
                const bool eq3 = eqall != 0;
                // every in-block log value lies in [mn3 - smx3, mx3 - smn3]: below 2^15 in magnitude none of them
                // needs a third byte; otherwise the exact classes_pass is requested (bit 60 of the top pack)
                const int32_t lo_b = mn3 - smx3, hi_b = mx3 - smn3;
                wide = (lo_b < -32768 || hi_b > 32767) ? 1u : 0u;
                sh.smin[tid] = smn3;
                sh.smax[tid] = smx3;
                sh.diff[tid] = df2_0;
                sh.eq[tid] = eq3 ? 1u : 0u;
            }
            r.flags = eqbits | cntbits | (wide << 28);
            r.u2 = u2;
            if (tid < C::TBW) {
                sh.tbS[tid] = 0;
                sh.tbL[tid] = 0;
            }
            r.sc[3] = lI1 | (lI2 << 16);
            if (err != 0) ex.lds_min(&sh.err, err);
        });
        const int32_t perr = ex.uni(sh.err);
        const uint32_t stI = ex.uni(sh.stI), stQ = ex.uni(sh.stQ);  // (zeroed again in phase 3)
        if (perr != 0) {
            status = perr == ERR_RANGE ? (int32_t)ST_UNSUPPORTED : perr;
            break;
        }
        ex.stamp(0);  // phase 1: load + thread-local analysis

        // ================= pre-pass over the stash records: which of their values need a second byte ==========
        // Sparse work (about an eighth of the quads and two fifths of the height-2 nodes of a benchmark log have a record),
        // taken out of the dense phase 1.  Records are in arrival order; each ORs its flags into its owner's words at the
        // place its ordinal gives.  rec_corr: what phase 1 left added to every 16-bit value of a record (see lean phase 1).
        const bool rec_ovf = stI > (uint32_t)SH::CAPI_REC || stQ > (uint32_t)SH::CAPQ_REC;
        const uint32_t rec_corr = lean ? (uint32_t)s_base & 0xffffu : 0u;
        auto rec16 = [&](uint32_t half) -> int32_t { return (int32_t)(int16_t)(uint16_t)(half - rec_corr); };
        auto lng = [](int32_t v) -> uint32_t { return ((uint32_t)v + 128u) > 255u ? 1u : 0u; };  // zig-zag(v) > 0xff
        // one node of the tree top from its four children (snapshot.rs:476-497, log.rs:776-806)
        auto top_node = [&](int h, int j) {
            const int c = C::top_off(h - 1) + 4 * j, a = C::top_off(h) + j;
            sh.tmin[a] = min4(sh.tmin[c], sh.tmin[c + 1], sh.tmin[c + 2], sh.tmin[c + 3]);
            sh.tmax[a] = max4(sh.tmax[c], sh.tmax[c + 1], sh.tmax[c + 2], sh.tmax[c + 3]);
            if (have_s) {
                sh.smin[a] = min4(sh.smin[c], sh.smin[c + 1], sh.smin[c + 2], sh.smin[c + 3]);
                sh.smax[a] = max4(sh.smax[c], sh.smax[c + 1], sh.smax[c + 2], sh.smax[c + 3]);
                const int32_t d0 = sh.diff[c];
                sh.diff[a] = d0;
                sh.eq[a] = (sh.eq[c] & sh.eq[c + 1] & sh.eq[c + 2] & sh.eq[c + 3]) && d0 == sh.diff[c + 1] && d0 == sh.diff[c + 2] &&
                                   d0 == sh.diff[c + 3]
                               ? 1u
                               : 0u;
            }
        };
        // the pre-pass over the records first + first + step, first + 2 step, ...
        auto prepass = [&](uint32_t first, uint32_t step) {
            for (uint32_t m = first; m < stQ; m += step) {
                uint32_t q[3];
                if (m < (uint32_t)SH::CAPQ_REC) {
#pragma unroll
                    for (int i = 0; i < 3; i++) q[i] = sh.pool[SH::QBASE + 3u * m + i];
                } else {
                    gload_words<3>(ovQ + 3u * (m - (uint32_t)SH::CAPQ_REC), q);
                }
                const uint32_t own = q[0] & 1023u, ord = (q[0] >> 10) & 15u;
                const uint32_t f = lng(rec16(q[1] & 0xffffu)) | (lng(rec16(q[1] >> 16)) << 1) | (lng(rec16(q[2] & 0xffffu)) << 2) |
                                   (lng(rec16(q[2] >> 16)) << 3);
                if (f) ex.lds_or(&sh.lq[own][ord >> 3], f << (4u * (ord & 7u)));
            }
            for (uint32_t k = first; k < stI; k += step) {
                uint32_t rec[5];
                if (k < (uint32_t)SH::CAPI_REC) {
#pragma unroll
                    for (int i = 0; i < 5; i++) rec[i] = sh.pool[5u * k + i];
                } else {
                    gload_words<5>(ovI + 5u * (k - (uint32_t)SH::CAPI_REC), rec);
                }
                const uint32_t own = rec[0] & 1023u, ord2 = (rec[0] >> 10) & 3u, tb1 = (rec[0] >> 16) & 15u;
                uint32_t f = 0;
#pragma unroll
                for (int qq = 0; qq < 4; qq++) {
                    f |= lng(rec16(rec[1 + qq] & 0xffffu)) << qq;                                   // Lmax of quad qq
                    f |= (((tb1 >> (3 - qq)) & 1u) & lng(rec16(rec[1 + qq] >> 16))) << (4 + qq);     // Lmin, internal quads only
                }
                if (f) ex.lds_or(&sh.li[own], f << (8u * ord2));
            }
        };
        // records that overflowed to global scratch are read by other threads: their stores must have landed
        if (have_s && rec_ovf) ex.barrier_global();
#ifdef K2R_TOPWAVE  // (built, bit-exact and 2.3 % SLOWER than the phased form, 11.48 against 11.22 ms on the same box: off)
        constexpr bool kTopWave = !EX::kSim && NT > 64;
#else
        constexpr bool kTopWave = false;
#endif
#ifdef K2R_TOPDPP  // (built, bit-exact on the GPU, and 0.3 % slower than the phased form in an A/B on one box, 11.18 against 11.14 ms: off)
        constexpr bool kTopDpp = !EX::kSim && !kTopWave && H >= 7;
#else
        constexpr bool kTopDpp = false;
#endif
        if (kTopWave && have_s) {
            // The GPU form for logs (sidelen 128 and 256).  The tree top as five barrier-separated LDS phases kept fifteen waves idle
            // for most of five phases; the pre-pass is sparse work over a few thousand records; neither needs the other.  So ONE wave
            // builds the whole top -- lane = height-4 node (Morton order: 64 of them per trip, one trip at sidelen 128, four at 256),
            // the four children read from LDS, heights 5, 6 and 7 reduced in registers over quads, rows and the wave (DPP, readlane),
            // the root from the four trips' scalars -- while the other fifteen waves run the pre-pass.
            if constexpr (kTopWave && H >= 7) ex.par([&](int tid, EncRegs&) {
                if (tid < 64) {
                    constexpr int TRIPS = 1 << (2 * (H - 7));  // height-7 nodes
                    int32_t rmin = 0, rmax = 0, rsmin = 0, rsmax = 0, rdiff = 0;
                    uint32_t rz = 0;
#pragma unroll 1
                    for (int i = 0; i < TRIPS; i++) {
                        const int j4 = 64 * i + tid;
                        top_node(4, j4);
                        const int a4 = C::top_off(4) + j4;
                        int32_t mn = sh.tmin[a4], mx = sh.tmax[a4], sn = sh.smin[a4], sx = sh.smax[a4], df = sh.diff[a4];
                        uint32_t z = sh.eq[a4] ^ 1u;  // "not equal" so far
                        // height 5: quads
                        mn = EX::quad_min(mn); mx = EX::quad_max(mx); sn = EX::quad_min(sn); sx = EX::quad_max(sx);
                        int32_t d5 = (int32_t)EX::template quad_bcast<0>((uint32_t)df);
                        z = EX::quad_or(z | (df != d5 ? 1u : 0u));
                        if ((tid & 3) == 0) {
                            const int a = C::top_off(5) + (j4 >> 2);
                            sh.tmin[a] = mn; sh.tmax[a] = mx; sh.smin[a] = sn; sh.smax[a] = sx; sh.diff[a] = d5; sh.eq[a] = z ^ 1u;
                        }
                        // height 6: rows of sixteen lanes
                        mn = EX::row_min_of_quads(mn); mx = EX::row_max_of_quads(mx); sn = EX::row_min_of_quads(sn); sx = EX::row_max_of_quads(sx);
                        const int32_t d6 = (int32_t)EX::lane_pull((uint32_t)tid & 48u, (uint32_t)d5);
                        z = EX::row_or_of_quads(z | (d5 != d6 ? 1u : 0u));
                        if ((tid & 15) == 0) {
                            const int a = C::top_off(6) + (j4 >> 4);
                            sh.tmin[a] = mn; sh.tmax[a] = mx; sh.smin[a] = sn; sh.smax[a] = sx; sh.diff[a] = d6; sh.eq[a] = z ^ 1u;
                        }
                        // height 7: the wave (scalars)
                        const int32_t mn7 = EX::wave_min_of_rows(mn), mx7 = EX::wave_max_of_rows(mx), sn7 = EX::wave_min_of_rows(sn),
                                      sx7 = EX::wave_max_of_rows(sx), d7 = (int32_t)ex.lane_value((uint32_t)d6, 0);
                        const uint32_t z7 = EX::wave_or_of_rows(z | (d6 != d7 ? 1u : 0u));
                        if (tid == 0) {
                            const int a = C::top_off(7) + i;
                            sh.tmin[a] = mn7; sh.tmax[a] = mx7; sh.smin[a] = sn7; sh.smax[a] = sx7; sh.diff[a] = d7; sh.eq[a] = z7 ^ 1u;
                        }
                        // height 8: the root, from the trips
                        rmin = i == 0 ? mn7 : (mn7 < rmin ? mn7 : rmin);
                        rmax = i == 0 ? mx7 : (mx7 > rmax ? mx7 : rmax);
                        rsmin = i == 0 ? sn7 : (sn7 < rsmin ? sn7 : rsmin);
                        rsmax = i == 0 ? sx7 : (sx7 > rsmax ? sx7 : rsmax);
                        if (i == 0) rdiff = d7;
                        rz |= z7 | (d7 != rdiff ? 1u : 0u);
                    }
                    if (H == 8 && tid == 0) {
                        const int a = C::top_off(H);
                        sh.tmin[a] = rmin; sh.tmax[a] = rmax; sh.smin[a] = rsmin; sh.smax[a] = rsmax; sh.diff[a] = rdiff; sh.eq[a] = rz ^ 1u;
                    }
                } else if (!(K2R_DIAG_SKIP & 64)) {
                    prepass((uint32_t)tid - 64u, (uint32_t)NT - 64u);
                }
            });
        } else if constexpr (kTopDpp) {
            // ================= phase 2 on the GPU, sidelen 128 and 256: heights 4..6 in registers, every wave its own ==========
            // Thread tid owns height-3 node tid and the threads are in Morton order, so the four children of a height-4 node sit in
            // one quad of lanes, those of a height-5 node in one row of sixteen, those of a height-6 node in one wave: three levels
            // by DPP reductions, no LDS traffic but the results, no barrier -- all sixteen waves at once, behind their share of the
            // pre-pass.  One wave then finishes heights 7 and 8 from the height-6 nodes.  (As five barrier-separated LDS phases the
            // tree top keeps wave 0 busy for 8.1 K cycles of an instant and the other waves waiting; this form takes the phase from
            // 8.6 K to 7.0 K cycles per wave in the instrumented build -- and the launch from 11.14 to 11.18 ms: the kernel moves its
            // 46.8 GB at 86 % of the rate of a device-to-device copy, and cycles saved between two loads do not shorten it.)
            if (have_s && !(K2R_DIAG_SKIP & 64)) ex.par_nosync([&](int tid, EncRegs&) { prepass((uint32_t)tid, (uint32_t)NT); });
            ex.par([&](int tid, EncRegs&) {
                int32_t mn = sh.tmin[tid], mx = sh.tmax[tid], sn = 0, sx = 0, df = 0;
                uint32_t z = 0;  // "not equal"
                if (have_s) {
                    sn = sh.smin[tid];
                    sx = sh.smax[tid];
                    df = sh.diff[tid];
                    z = sh.eq[tid] ^ 1u;
                }
                auto put = [&](int a, int32_t d) {
                    sh.tmin[a] = mn;
                    sh.tmax[a] = mx;
                    if (have_s) {
                        sh.smin[a] = sn;
                        sh.smax[a] = sx;
                        sh.diff[a] = d;
                        sh.eq[a] = z ^ 1u;
                    }
                };
                // height 4: quads
                mn = EX::quad_min(mn); mx = EX::quad_max(mx); sn = EX::quad_min(sn); sx = EX::quad_max(sx);
                const int32_t d4 = (int32_t)EX::template quad_bcast<0>((uint32_t)df);
                z = EX::quad_or(z | (df != d4 ? 1u : 0u));
                if ((tid & 3) == 0) put(C::top_off(4) + (tid >> 2), d4);
                // height 5: rows of sixteen lanes
                mn = EX::row_min_of_quads(mn); mx = EX::row_max_of_quads(mx); sn = EX::row_min_of_quads(sn); sx = EX::row_max_of_quads(sx);
                const int32_t d5 = (int32_t)EX::lane_pull((uint32_t)tid & 48u, (uint32_t)d4);
                z = EX::row_or_of_quads(z | (d4 != d5 ? 1u : 0u));
                if ((tid & 15) == 0) put(C::top_off(5) + (tid >> 4), d5);
                // height 6: the wave
                const int32_t d6 = (int32_t)ex.lane_value((uint32_t)d5, 0);
                const uint32_t z6 = EX::wave_or_of_rows(z | (d5 != d6 ? 1u : 0u));
                mn = EX::wave_min_of_rows(mn); mx = EX::wave_max_of_rows(mx); sn = EX::wave_min_of_rows(sn); sx = EX::wave_max_of_rows(sx);
                z = z6;
                if ((tid & 63) == 0) put(C::top_off(6) + (tid >> 6), d6);
            });
            ex.par([&](int tid, EncRegs&) {
                if (tid >= 64) return;
                constexpr int N6 = 1 << (2 * (H - 6));  // height-6 nodes: 4 (one quad of lanes) or 16 (one row)
                const int a6 = C::top_off(6) + (tid < N6 ? tid : 0);
                int32_t mn = sh.tmin[a6], mx = sh.tmax[a6], sn = 0, sx = 0, df = 0;
                uint32_t z = 0;
                if (have_s) {
                    sn = sh.smin[a6];
                    sx = sh.smax[a6];
                    df = sh.diff[a6];
                    z = sh.eq[a6] ^ 1u;
                }
                auto put = [&](int a, int32_t d) {
                    sh.tmin[a] = mn;
                    sh.tmax[a] = mx;
                    if (have_s) {
                        sh.smin[a] = sn;
                        sh.smax[a] = sx;
                        sh.diff[a] = d;
                        sh.eq[a] = z ^ 1u;
                    }
                };
                // height 7: quads of the height-6 nodes
                mn = EX::quad_min(mn); mx = EX::quad_max(mx); sn = EX::quad_min(sn); sx = EX::quad_max(sx);
                const int32_t d7 = (int32_t)EX::template quad_bcast<0>((uint32_t)df);
                z = EX::quad_or(z | (df != d7 ? 1u : 0u));
                if ((tid & 3) == 0 && tid < N6) put(C::top_off(7) + (tid >> 2), d7);
                if (H == 8) {  // height 8: the row of sixteen
                    mn = EX::row_min_of_quads(mn); mx = EX::row_max_of_quads(mx); sn = EX::row_min_of_quads(sn); sx = EX::row_max_of_quads(sx);
                    const int32_t d8 = (int32_t)ex.lane_value((uint32_t)d7, 0);
                    z = EX::row_or_of_quads(z | (d7 != d8 ? 1u : 0u));
                    if (tid == 0) put(C::top_off(8), d8);
                }
            });
        } else {
            if (have_s) {
                if (!(K2R_DIAG_SKIP & 64)) ex.par_nosync([&](int tid, EncRegs&) { prepass((uint32_t)tid, (uint32_t)NT); });
                if (H < 4) ex.barrier();  // (phase 3 reads the flags; taller trees have phase 2's barriers in between)
            }
            // ================= phase 2: heights 4..H in LDS ==========
            for (int h = 4; h <= H; h++) {
                const int n_h = 1 << (2 * (H - h));
                ex.par([&](int tid, EncRegs&) {
                    for (int j = tid; j < n_h; j += NT) top_node(h, j);
                });
            }
        }

        ex.stamp(1);  // phase 2: top of the tree
        // node predicates on the top arrays (index = top_off(h) + j)
        auto PS = [&](int h, uint32_t j) -> bool {
            const int a = C::top_off(h) + (int)j;
            return !top_inval(h, j) && sh.tmin[a] != sh.tmax[a];
        };
        auto PL = [&](int h, uint32_t j) -> bool {
            const int a = C::top_off(h) + (int)j;
            return !top_inval(h, j) && sh.tmin[a] != sh.tmax[a] && sh.eq[a] == 0;
        };
        // Lmax / Lmin values of node (h,j), h >= 3 (root: absolute, snapshot.rs:123)
        auto snap_vmax = [&](int h, uint32_t j) -> int32_t {
            const int a = C::top_off(h) + (int)j;
            if (h == H) return sh.tmax[a];
            const int p = C::top_off(h + 1) + (int)(j >> 2);
            return top_inval(h, j) ? sh.tmax[p] : sh.tmax[p] - sh.tmax[a];  // snapshot.rs:139
        };
        auto snap_vmin = [&](int h, uint32_t j) -> int32_t {
            const int a = C::top_off(h) + (int)j;
            if (h == H) return sh.tmin[a];
            const int p = C::top_off(h + 1) + (int)(j >> 2);
            return sh.tmin[a] - sh.tmin[p];  // snapshot.rs:140
        };
        auto log_vmax = [&](int h, uint32_t j) -> int32_t {
            const int a = C::top_off(h) + (int)j;
            return top_inval(h, j) ? 0 : sh.tmax[a] - sh.smax[a];  // log.rs:133
        };
        auto log_vmin = [&](int h, uint32_t j) -> int32_t {
            const int a = C::top_off(h) + (int)j;
            return sh.tmin[a] - sh.smin[a];  // log.rs:148
        };

        // worker thread n < NTOPX looks after top node n (heights 4..H, level order); bit index == n
        auto top_decode = [&](uint32_t n, int& h, uint32_t& j) {
            h = 4;
#pragma unroll
            for (int k = 4; k < H; k++)
                if (n >= (uint32_t)(C::top_off(k + 1) - C::NBLK)) h = k + 1;
            j = n - (uint32_t)(C::top_off(h) - C::NBLK);
        };
        auto tbit = [&](int h, uint32_t j) -> uint32_t { return (uint32_t)(C::top_off(h) - C::NBLK) + j; };

        // ================= phase 3: own node + (for the first NTOPX threads) one top node each ==========
        // (no barrier of its own: the reduction that follows only takes this phase's per-thread counts)
        ex.par_nosync([&](int tid, EncRegs& r) {
            uint32_t sI3 = 0, lI3 = 0;
            uint32_t wide = (r.flags >> 28) & 1u;
            Cls lMax, lMin;
            uint32_t sTop = 0, lTop = 0;
            auto node = [&](int h, uint32_t j) {
                if (PS(h, j)) {
                    if (h == 3) sI3 = 1;
                    else {
                        ex.lds_or(&sh.tbS[tbit(h, j) >> 5], 1u << (tbit(h, j) & 31));
                        sTop += (uint32_t)packTop(h);
                    }
                }
                if (have_s) {
                    const bool visL = (h == H) ? true : PL(h + 1, j >> 2);
                    const bool pL = PL(h, j);
                    const int32_t vx = log_vmax(h, j), vn = log_vmin(h, j);
                    lMax.add1(vx, visL);
                    lMin.add1(vn, pL);
                    wide |= ((visL && (vx < -32768 || vx > 32767)) || (pL && (vn < -32768 || vn > 32767))) ? 1u : 0u;
                    if (pL) {
                        if (h == 3) lI3 = 1;
                        else {
                            ex.lds_or(&sh.tbL[tbit(h, j) >> 5], 1u << (tbit(h, j) & 31));
                            lTop += (uint32_t)packTop(h);
                        }
                    }
                }
            };
            const uint32_t wt = opaque((uint32_t)tid);
            node(3, wt);
            if (tid == 0) {
                sh.stI = 0;
                sh.stQ = 0;
            }
            if (wt < (uint32_t)C::NTOPX) {
                int h;
                uint32_t j;
                top_decode(wt, h, j);
                node(h, j);
            }
            // reduce words: [0] snapshot I1 | I2 << 16   [1] snapshot I3 | log I3 << 16   [2] snapshot top pack
            //               [3] log I1 | I2 << 16        [4] log top pack
            //               [5] long log values at heights >= 3: Lmax | Lmin << 16      [6] "wide" requests
            //               [7] cL0    [8] cL1 | cL2 << 16    [9] mL1 | mL2 << 16      (phase 1)
            r.sc[1] = sI3 | (lI3 << 16);
            r.sc[2] = sTop;
            r.sc[4] = lTop;
            r.sc[5] = lMax.c1 | (lMin.c1 << 16);
            r.sc[6] = wide;
            // second bytes of this block's log values per level: cells and height-1 values from the flags the pre-pass left,
            // height-2 values from the four pairs phase 1 kept (Lmax of all four when the block's own node is internal, Lmin of
            // the internal ones); meaningful for narrow logs only, like the pairs themselves
            uint32_t cL0 = 0, cL1 = 0, mL1 = 0, cL2 = 0, mL2 = 0;
            if (have_s) {
                const uint32_t fl = sh.li[tid];
                cL0 = popc32(sh.lq[tid][0]) + popc32(sh.lq[tid][1]);
                cL1 = popc32(fl & 0x0f0f0f0fu);
                mL1 = popc32(fl & 0xf0f0f0f0u);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int32_t vx = rec16(r.d2[j] & 0xffffu), vn = rec16(r.d2[j] >> 16);
                    r.d2[j] = ((uint32_t)vx & 0xffffu) | ((uint32_t)vn << 16);
                    const bool P2L = ((r.u2 >> j) & 1u) == 0 && ((r.flags >> j) & 1u) == 0;
                    cL2 += lI3 ? lng(vx) : 0u;
                    mL2 += P2L ? lng(vn) : 0u;
                }
            }
            r.sc[7] = cL0;
            r.sc[8] = cL1 | (cL2 << 16);
            r.sc[9] = mL1 | (mL2 << 16);
        });
        ex.stamp(2);  // phase 3: own/top nodes
        // one scan serves both the totals the planner needs and the prefixes emission needs (of both candidates:
        // the winner's are picked once it is known); it is finished inside the planning phase
        ex.template scan_begin<10>();
        ex.stamp(3);  // totals of the 4 packed fields

        // Exact byte classes of EVERY value of one candidate (0 = snapshot, 1 = log): a second streaming pass over
        // the tile with compact rolled loops, run only when the exact figure matters (see phase 4).  Totals end up
        // in sh.tot[10..12] (Lmax: > 1, > 2, > 3 bytes) and sh.tot[13..15] (Lmin).
        auto classes_pass = [&](const int which) {
#ifdef K2R_SIM_TRACE
            if (EX::kSim) __builtin_printf("classes_pass(%d) inst=%u\n", which, inst);
#endif
            ex.par_nosync([&](int tid, EncRegs& r) {
                uint32_t r0, c0;
                blk_origin(tid, r0, c0);
                int32_t lerr = 0;
                Cls vMax, vMin;
                auto cnode = [&](int h, uint32_t j) {
                    if (which == 0) {
                        vMax.add(zz32(snap_vmax(h, j)), (h == H) ? true : PS(h + 1, j >> 2));
                        vMin.add(zz32(snap_vmin(h, j)), PS(h, j));
                    } else {
                        vMax.add(zz32(log_vmax(h, j)), (h == H) ? true : PL(h + 1, j >> 2));
                        vMin.add(zz32(log_vmin(h, j)), PL(h, j));
                    }
                };
                cnode(3, (uint32_t)tid);
                if (tid < C::NTOPX) {
                    int h;
                    uint32_t j;
                    top_decode((uint32_t)tid, h, j);
                    cnode(h, j);
                }
                const int32_t mn3 = sh.tmin[tid], mx3 = sh.tmax[tid];
                const bool inv3 = inval(r0, c0);
                const bool P3 = !inv3 && mn3 != mx3 && (which == 0 || sh.eq[tid] == 0);
#pragma unroll 1
                for (int j = 0; j < 4; j++) {
                    const uint32_t rj = r0 + 4 * (j >> 1), cj = c0 + 4 * (j & 1);
                    const bool inv2 = inval(rj, cj);
                    int32_t t16[16], s16[16];
                    load_sub16<PADDED, VEC>(ta, inst, r0, c0, j, t16, lerr);
                    if (which != 0) load_sub16<PADDED, VEC>(ta, s_idx, r0, c0, j, s16, lerr);
                    int32_t mn2 = t16[0], mx2 = t16[0], smn2 = 0, smx2 = 0;
#pragma unroll
                    for (int i = 1; i < 16; i++) {
                        mn2 = t16[i] < mn2 ? t16[i] : mn2;
                        mx2 = t16[i] > mx2 ? t16[i] : mx2;
                    }
                    if (which != 0) {
                        smn2 = smx2 = s16[0];
#pragma unroll
                        for (int i = 1; i < 16; i++) {
                            smn2 = s16[i] < smn2 ? s16[i] : smn2;
                            smx2 = s16[i] > smx2 ? s16[i] : smx2;
                        }
                    }
                    const bool unif2 = inv2 || mn2 == mx2;
                    bool P2;
                    if (which == 0) {
                        P2 = !unif2;
                        vMax.add(zz32(inv2 ? mx3 : mx3 - mx2), P3);
                        vMin.add(zz32(mn2 - mn3), P2);
                    } else {
                        P2 = !unif2 && ((r.flags >> j) & 1u) == 0;
                        vMax.add(zz32(inv2 ? 0 : mx2 - smx2), P3);
                        vMin.add(zz32(mn2 - smn2), P2);
                    }
                    if (!P2) continue;
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) {
                        const uint32_t rq = rj + 2 * (qq >> 1), cq = cj + 2 * (qq & 1);
                        const bool inv1 = inval(rq, cq);
                        const int32_t mn1 = min4(t16[4 * qq], t16[4 * qq + 1], t16[4 * qq + 2], t16[4 * qq + 3]);
                        const int32_t mx1 = max4(t16[4 * qq], t16[4 * qq + 1], t16[4 * qq + 2], t16[4 * qq + 3]);
                        const bool unif1 = inv1 || mn1 == mx1;
                        if (which == 0) {
                            vMax.add(zz32(inv1 ? mx2 : mx2 - mx1), true);
                            vMin.add(zz32(mn1 - mn2), !unif1);
#pragma unroll
                            for (int i = 0; i < 4; i++)
                                vMax.add(zz32(inval(rq + (i >> 1), cq + (i & 1)) ? mx1 : mx1 - t16[4 * qq + i]), !unif1);
                        } else {
                            int32_t d[4];
#pragma unroll
                            for (int i = 0; i < 4; i++)
                                d[i] = inval(rq + (i >> 1), cq + (i & 1)) ? 0 : t16[4 * qq + i] - s16[4 * qq + i];
                            const int32_t smn1 = min4(s16[4 * qq], s16[4 * qq + 1], s16[4 * qq + 2], s16[4 * qq + 3]);
                            const int32_t smx1 = max4(s16[4 * qq], s16[4 * qq + 1], s16[4 * qq + 2], s16[4 * qq + 3]);
                            const bool P1 = !unif1 && !(d[0] == d[1] && d[0] == d[2] && d[0] == d[3]);
                            vMax.add(zz32(inv1 ? 0 : mx1 - smx1), true);
                            vMin.add(zz32(mn1 - smn1), P1);
#pragma unroll
                            for (int i = 0; i < 4; i++) vMax.add(zz32(d[i]), P1);
                        }
                    }
                }
                r.sc[10] = vMax.c1;
                r.sc[11] = vMax.c2;
                r.sc[12] = vMax.c3;
                r.sc[13] = vMin.c1;
                r.sc[14] = vMin.c2;
                r.sc[15] = vMin.c3;
            });
            ex.template reduce<6, 10>();
        };

        // ================= phase 4: sizes and the heuristic (chunk.rs:62) ===============================
        // Planned by ONE lane into sh.pl (see InstPlan).  Sizes are functions of counts; the log's byte classes
        // come inline from phase 1 unless a value may need 3+ bytes, the snapshot's are only counted when its
        // one-byte-per-value lower bound does not already lose against the log (snapshot.rs:87-92, log.rs:95-97).
        const bool cap254 = have_s && (blk_count - 1 == 254);  // chunk.rs:62 (checked first)
        auto plan = [&](const int stage) {
            ex.par([&](int tid, EncRegs&) {
                uint32_t T10[10];
                if (stage == 1) {
                    ex.template scan_finish<10>(T10);  // every thread: its prefixes; the totals for the planner
                    // the emission bitmaps are cleared here, while all threads but the planner would only wait
                    for (uint32_t w = (uint32_t)tid; w <= (uint32_t)C::WT; w += NT) {
                        sh.bmT[w] = 0;
                        sh.bmE[w] = 0;
                        sh.bmM[0][w] = 0;
                    }
                    for (uint32_t w = (uint32_t)tid; w <= (uint32_t)C::WV; w += NT) sh.bmV0[w] = 0;
                }
                // Stage 1 runs on two lanes of wave 0 at once where it can -- lane 0 plans the snapshot candidate, lane 1
                // the log candidate, the same instructions on different data -- and exchanges the two sizes with readlane;
                // the sequential context (and a 1-thread workgroup) lets thread 0 do both in turn.
                constexpr bool kTwoLanes = !EX::kSim && NT >= 2;
                if (tid >= (stage == 1 && kTwoLanes ? 2 : 1)) return;
                auto& pl = sh.pl;
                // the decision: W = the winner's totals, size = its serialized size
                auto choose = [&](bool snap, const Totals<C>& W, uint32_t size, uint32_t narrow) {
                    pl.as_snapshot = snap ? 1u : 0u;
                    pl.isize = size;
                    // a Log is emitted from the stash when phase 1 managed to record all of it
                    pl.use_stash = (!snap && narrow && 5u * stI + 3u * stQ <= stash_cap && stI == W.Ni[2] && stQ == W.Ni[1]) ? 1u : 0u;
                    uint32_t rr = 0;
#pragma unroll
                    for (int h = 4; h <= H; h++) {
                        sh.ttR[h] = rr;
                        rr += W.Ni[h];
                    }
                    sh.ttR[H + 1] = rr;
                };
                if (stage == 1) {
                    // everything is computed in registers and stored once: a chain of LDS round trips here would be paid by
                    // the whole workgroup waiting at the barrier
                    const uint32_t t0 = T10[0], t1 = T10[1], t2 = T10[2], t3 = T10[3], t4 = T10[4], t5 = T10[5], t6 = T10[6], t7 = T10[7],
                                   t8 = T10[8], t9 = T10[9];
                    const uint32_t narrow = (have_s && t6 == 0) ? 1u : 0u;
                    // second bytes come in level order too: first those of heights >= 3, then height 2, 1, 0
                    const uint32_t lv2 = t5 & 0xffffu, lv1 = lv2 + (t8 >> 16), lv0 = lv1 + (t8 & 0xffffu);
                    const uint32_t lm2 = t5 >> 16, lm1 = lm2 + (t9 >> 16);
                    // candidate c (0 snapshot, 1 log): level offsets, Dac layouts (the snapshot's with one byte per value:
                    // a lower bound; the log's with the inline "> 1 byte" counts, exact when narrow), serialized size
                    auto cand = [&](uint32_t c, Totals<C>& T, DacLayout& V, DacLayout& M, uint32_t& eo) {
                        T.from(c == 0 ? ((uint64_t)t0 | ((uint64_t)(t1 & 0xffffu) << 30)) : ((uint64_t)t3 | ((uint64_t)(t1 >> 16) << 30)),
                               c == 0 ? t2 : t4);
                        eo = 13 + bitmap_size(T.LT);
                        const uint32_t vbase = c == 0 ? eo : eo + bitmap_size(T.LT - T.M0);
                        const bool counts = c == 1 && narrow;
                        V = dac_layout(vbase, T.N0, counts ? lv0 + t7 : 0u, 0, 0);
                        M = dac_layout(V.end, T.M0, counts ? lm1 + (t9 & 0xffffu) : 0u, 0, 0);
                    };
                    uint32_t snap_lb, log_size, eq_off;
                    Totals<C> TW;  // the log candidate's totals, in the registers of the thread that may have to choose() it
                    bool i_hold_log;
                    if (kTwoLanes) {
                        DacLayout V, M;
                        uint32_t eo;
                        cand((uint32_t)tid, TW, V, M, eo);
                        if (tid == 0 || have_s) {
                            pl.T[tid] = TW;
                            pl.V[tid] = V;
                            pl.M[tid] = M;
                        }
                        snap_lb = ex.lane_value(M.end, 0);
                        log_size = have_s ? ex.lane_value(M.end, 1) : 0u;
                        eq_off = have_s ? ex.lane_value(eo, 1) : 0u;
                        i_hold_log = tid == 1;
                    } else {
                        Totals<C> TS;
                        DacLayout SV, SM, LV{}, LM{};
                        uint32_t eo0, eo1 = 0;
                        cand(0, TS, SV, SM, eo0);
                        pl.T[0] = TS;
                        pl.V[0] = SV;
                        pl.M[0] = SM;
                        if (have_s) {
                            cand(1, TW, LV, LM, eo1);
                            pl.T[1] = TW;
                            pl.V[1] = LV;
                            pl.M[1] = LM;
                        }
                        snap_lb = SM.end;
                        log_size = have_s ? LM.end : 0u;
                        eq_off = eo1;
                        i_hold_log = true;
                    }
                    uint32_t need = 0;
                    if (have_s && !narrow) {
                        need = 1;  // some log value may need 3+ bytes: count exactly first
                        log_size = 0;
                    } else if (!have_s || cap254 || snap_lb <= log_size) {
                        need = 2;  // the snapshot may win (or must be taken): count its byte classes exactly
                    } else if (i_hold_log) {
                        choose(false, TW, log_size, narrow);  // log.rs:95-97 vs snapshot.rs:87-92: the log wins
                    }
                    if (tid == 0) {
                        pl.lngV[0] = lv0;
                        pl.lngV[1] = lv1;
                        pl.lngV[2] = lv2;
                        pl.lngM[1] = lm1;
                        pl.lngM[2] = lm2;
                        pl.narrow = narrow;
                        pl.log_size = log_size;
                        pl.eq_off = eq_off;
                        pl.snap_lb = snap_lb;
                        pl.need = need;
                    }
                } else if (stage == 2) {  // exact classes of the log are in sh.tot[10..15]
                    pl.V[1] = dac_layout(pl.eq_off + bitmap_size(pl.T[1].LT - pl.T[1].M0), pl.T[1].N0, sh.tot[10], sh.tot[11], sh.tot[12]);
                    pl.M[1] = dac_layout(pl.V[1].end, pl.T[1].M0, sh.tot[13], sh.tot[14], sh.tot[15]);
                    pl.log_size = pl.M[1].end;
                    if (cap254 || pl.snap_lb <= pl.log_size) pl.need = 2;
                    else {
                        choose(false, pl.T[1], pl.log_size, pl.narrow);
                        pl.need = 0;
                    }
                } else {  // exact classes of the snapshot are in sh.tot[10..15]
                    const uint32_t sbase = 13 + bitmap_size(pl.T[0].LT);
                    pl.V[0] = dac_layout(sbase, pl.T[0].N0, sh.tot[10], sh.tot[11], sh.tot[12]);
                    pl.M[0] = dac_layout(pl.V[0].end, pl.T[0].M0, sh.tot[13], sh.tot[14], sh.tot[15]);
                    const bool snap = !have_s || cap254 || pl.M[0].end <= pl.log_size;
                    choose(snap, pl.T[snap ? 0 : 1], snap ? pl.M[0].end : pl.log_size, pl.narrow);
                    pl.need = 0;
                }
            });
        };
        plan(1);
        uint32_t need = ex.uni(sh.pl.need);
        if (need & 1u) {
            classes_pass(1);
            plan(2);
            need = ex.uni(sh.pl.need);
        }
        if (need & 2u) {
            classes_pass(0);
            plan(3);
        }
        const bool as_snapshot = ex.uni(sh.pl.as_snapshot) != 0;
        const bool use_stash = ex.uni(sh.pl.use_stash) != 0;
        const uint32_t isize = ex.uni(sh.pl.isize);
        const uint32_t log_eq_off = ex.uni(sh.pl.eq_off);
        // the winner's level offsets and Dac layouts: read from LDS where they are used
        const Totals<C>& TT = sh.pl.T[as_snapshot ? 0 : 1];
        const DacLayout& DV = sh.pl.V[as_snapshot ? 0 : 1];
        const DacLayout& DM = sh.pl.M[as_snapshot ? 0 : 1];

        uint32_t hdr_patch_off = 0, hdr_patch_val = 0;
        bool do_patch = false;
        if (as_snapshot) {
            if (have_s) {  // close the open block (chunk.rs:63-70)
                if (blk_hdr == NO_HDR) carry = blk_count;  // the inherited block: k_stitch patches its count byte
                else {
                    do_patch = true;
                    hdr_patch_off = blk_hdr;
                    hdr_patch_val = blk_count;
                }
                n_blocks++;
            }
            blk_hdr = off;
            off += 1;
            blk_count = 0;
            s_idx = inst;
            n_snap++;
            const int32_t rmin = ex.uni(sh.tmin[C::top_off(H)]), rmax = ex.uni(sh.tmax[C::top_off(H)]);
            s_cmp = (int64_t)rmax - (int64_t)rmin <= 65535 && inst + 1 < ta.instants;
            s_base = rmin;
            cmp = (head && inst == 0) ? ta.shared_cmp : scmp;
        } else {
            n_log++;
        }
        if ((uint64_t)off + isize > cap) {
            status = ST_OUT_CAPACITY;
            break;
        }
        uint8_t* const io = out + off;  // first byte of this Snapshot / Log
        n_stash += use_stash ? 1u : 0u;
        ex.stamp(12);  // sizes + heuristic (wave-uniform arithmetic, lazy class passes)
        // ================= phase 5: emission of the winner ===============================================
        // 5a. save prefixes, header (the bitmaps were cleared during planning)
        ex.par([&](int tid, EncRegs& r) {
            // exclusive prefixes of the winner's counts (phase 3 scanned both candidates')
            const uint32_t pI12 = as_snapshot ? r.sc[0] : r.sc[3];                      // I1 | I2 << 16
            const uint32_t pI3 = as_snapshot ? (r.sc[1] & 0xffffu) : (r.sc[1] >> 16);
            const uint32_t pL0 = r.sc[7], pL12 = r.sc[8], pM12 = r.sc[9];               // second bytes per level (logs)
            r.pf_lo = (uint64_t)pI12 | ((uint64_t)pI3 << 30);  // the lo pack of unpackI
            r.pf_l2 = (pL12 >> 16) | (pM12 & 0xffff0000u);     // second bytes before this block's height-2 group: Lmax | Lmin << 16
            sh.pfx[0][tid] = pI12;
            sh.pfx[1][tid] = pL0;
            sh.pfx[2][tid] = (pL12 & 0xffffu) | (pM12 << 16);
            if (tid <= C::TBW) {  // rank prefix over the winner's top-node flags (<= 12 words)
                const uint32_t* const tbw = as_snapshot ? sh.tbS : sh.tbL;
                uint32_t run = 0;
                for (int w = 0; w < tid && w < C::TBW; w++) run += popc32(tbw[w]);
                sh.tbP[tid] = run;
            }
            if (tid == 0) {
                sh.nlistV = 0;
                sh.nlistM = 0;
                if (do_patch) out[hdr_patch_off] = (uint8_t)hdr_patch_val;
                io[0] = 2;  // k  (snapshot.rs:49, log.rs:54)
                store_be32(io + 1, ta.rows);
                store_be32(io + 5, ta.cols);
                store_be32(io + 9, (uint32_t)C::S);
                io[DV.bm_off[0] - 1] = (uint8_t)DV.nlev;  // dac.rs:38
                io[DM.bm_off[0] - 1] = (uint8_t)DM.nlev;
                if (ta.minmax) {
                    ta.minmax[2 * inst] = sh.tmin[C::top_off(H)];
                    ta.minmax[2 * inst + 1] = sh.tmax[C::top_off(H)];
                }
            }
        });

        // (records that overflowed to global scratch: the pre-pass already waited for their stores to land)
        ex.stamp(4);  // sizes, heuristic, clears, header
        const uint32_t nlevV = ex.uni(DV.nlev), nlevM = ex.uni(DM.nlev);
        const DacSink sinkV{io + ex.uni(DV.by_off[0]), sh.bmV0, listV, &sh.nlistV, ex.uni(DV.n[0]), ex.uni(DV.n[1]), inst, kGuardVPos,
                            io + ex.uni(DV.by_off[1]), use_stash ? sh.prefTopV : sh.prefV()};
        const DacSink sinkM{io + ex.uni(DM.by_off[0]), sh.bmM[0], listM, &sh.nlistM, ex.uni(DM.n[0]), ex.uni(DM.n[1]), inst, kGuardMPos,
                            io + ex.uni(DM.by_off[1]), sh.prefM};

        // 5b. plane 0 of both Dacs, T (and eqB) bits.  Pass A covers the nodes of heights >= 2; it is instantiated
        // per emission mode.
        auto passA = [&](auto mode_tag) {  // (no barrier at its end: the caller decides)
          constexpr int MODE = decltype(mode_tag)::value;
          ex.par_nosync([&](int tid, EncRegs& r) {
            uint32_t r0, c0;
            blk_origin(tid, r0, c0);
            const uint64_t pLo = r.pf_lo;

            // -- this thread's own height-3 node and, for the first NTOPX threads, one top node each --
            const uint32_t* const tb = as_snapshot ? sh.tbS : sh.tbL;
            if (MODE == EM_P0) r.pa[2] = 0;
            // nodes of heights >= 3 are placed by bitmap rank (EM_P0 now, second bytes replayed later); the height-2
            // group below knows where its second bytes go (EM_ONE)
            constexpr int GMODE = (MODE == EM_P0) ? (int)EM_ONE : MODE;
            auto enode = [&](int h, uint32_t j, const int slot) {
                const bool vis = (h == H) ? true : bit_test(tb, tbit(h + 1, j >> 2));
                if (!vis) return;
                const bool p = (h == 3) ? (as_snapshot ? PS(3, j) : PL(3, j)) : bit_test(tb, tbit(h, j));
                // level-order index: four children per internal parent, parents ranked among the internal nodes
                auto trank = [&](uint32_t b) -> uint32_t {  // set flags before bit b
                    return sh.tbP[b >> 5] + popc32(tb[b >> 5] & ((1u << (b & 31u)) - 1u));
                };
                uint32_t vrank = 0;
                if (h < H) vrank = 4 * (trank(tbit(h + 1, j >> 2)) - sh.ttR[h + 1]) + (j & 3);
                const uint32_t idx = TT.offV[h] + vrank;
                const uint32_t irank = (h == 3) ? unpackI(3, pLo, 0) : trank(tbit(h, j)) - sh.ttR[h];
                const uint32_t zv = zz32(as_snapshot ? snap_vmax(h, j) : log_vmax(h, j));
                emit_val<0, MODE>(ex, sinkV, idx, zv, tid);
                if (MODE == EM_P0) {
                    r.pa[3 * slot] = idx;
                    r.pa[2] |= (zv >> 8) << (16 * slot);
                }
                if (p) {
                    bm_set(ex, sh.bmT, guard_pos(ex, idx, 1, TT.LT, kGuardTOwn));
                    const uint32_t zm = zz32(as_snapshot ? snap_vmin(h, j) : log_vmin(h, j));
                    emit_val<1, MODE>(ex, sinkM, TT.offI[h] + irank, zm, tid);
                    if (MODE == EM_P0) {
                        r.pa[3 * slot + 1] = TT.offI[h] + irank;
                        r.pa[2] |= (zm >> 8) << (16 * slot + 8);
                    }
                } else if (!as_snapshot) {
                    const int a = C::top_off(h) + (int)j;
                    const bool e = !top_inval(h, j) && sh.tmin[a] != sh.tmax[a];  // not uniform => equal
                    if (e) bm_set(ex, sh.bmE, guard_pos(ex, TT.offZ[h] + vrank - irank, 1, TT.LT - TT.M0, kGuardEOwn));
                }
            };
            const uint32_t wt = opaque((uint32_t)tid);
            enode(3, wt, 0);
            if (wt < (uint32_t)C::NTOPX) {
                int h;
                uint32_t j;
                top_decode(wt, h, j);
                enode(h, j, 1);
            }

            // -- the four height-2 children of this thread's block + the work list of internal height-2 nodes --
            const int32_t mn3 = sh.tmin[tid], mx3 = sh.tmax[tid];
            const bool inv3 = inval(r0, c0);
            const uint32_t E1 = unpackI(1, pLo, 0), E2 = unpackI(2, pLo, 0), E3 = unpackI(3, pLo, 0);
            const bool P3 = !inv3 && mn3 != mx3 && (as_snapshot || sh.eq[tid] == 0);
            if (P3) {
                const uint32_t p2 = TT.offV[2] + 4 * E3;
                uint32_t tb2 = 0, z2v[4], zm2[4], erun = 0, elen = 0;
                bool P2[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const bool inv2 = inval(r0 + 4 * (j >> 1), c0 + 4 * (j & 1));
                    const bool unif2 = ((r.u2 >> j) & 1u) != 0;
                    int32_t vx, vn;  // the node's Lmax / Lmin values
                    if (MODE == EM_LIST) {  // snapshots and wide logs: from the tile again (see EncRegs)
                        int32_t lerr = 0, t16[16];
                        load_sub16<PADDED, VEC>(ta, inst, r0, c0, j, t16, lerr);
                        int32_t mn2 = t16[0], mx2 = t16[0];
#pragma unroll
                        for (int i = 1; i < 16; i++) {
                            mn2 = t16[i] < mn2 ? t16[i] : mn2;
                            mx2 = t16[i] > mx2 ? t16[i] : mx2;
                        }
                        if (as_snapshot) {
                            vx = inv2 ? mx3 : mx3 - mx2;  // snapshot.rs:139
                            vn = mn2 - mn3;               // snapshot.rs:140
                        } else {
                            int32_t s16[16];
                            load_sub16<PADDED, VEC>(ta, s_idx, r0, c0, j, s16, lerr);
                            int32_t smn2 = s16[0], smx2 = s16[0];
#pragma unroll
                            for (int i = 1; i < 16; i++) {
                                smn2 = s16[i] < smn2 ? s16[i] : smn2;
                                smx2 = s16[i] > smx2 ? s16[i] : smx2;
                            }
                            vx = inv2 ? 0 : mx2 - smx2;  // log.rs:133
                            vn = mn2 - smn2;             // log.rs:148
                        }
                    } else {  // narrow log: the int16 pair phase 1 kept
                        vx = (int32_t)(int16_t)(r.d2[j] & 0xffffu);
                        vn = (int32_t)r.d2[j] >> 16;
                    }
                    if (as_snapshot) {
                        P2[j] = !unif2;
                        z2v[j] = zz32(vx);
                        zm2[j] = zz32(vn);
                    } else {
                        P2[j] = !unif2 && ((r.flags >> j) & 1u) == 0;
                        z2v[j] = zz32(vx);
                        zm2[j] = zz32(vn);
                        if (!P2[j]) {  // T = 0: one eqB bit, set iff "equal" (log.rs:137-144)
                            erun = (erun << 1) | (unif2 ? 0u : 1u);
                            elen++;
                        }
                    }
                    tb2 = (tb2 << 1) | (P2[j] ? 1u : 0u);
                }
                emit4<0, GMODE>(ex, sinkV, p2, z2v[0], z2v[1], z2v[2], z2v[3], tid, sh.pl.lngV[2] + (r.pf_l2 & 0xffffu));
                bm_or_run(ex, sh.bmT, guard_pos(ex, p2, 4, TT.LT, kGuardTRun2), 4, tb2);
                if (!as_snapshot) bm_or_run(ex, sh.bmE, guard_pos<false>(ex, TT.offZ[2] + 4 * E3 - E2, elen, TT.LT - TT.M0, kGuardE2), elen, erun);
                uint32_t n2 = 0, pre = E1, lm = sh.pl.lngM[2] + (r.pf_l2 >> 16);
                const uint32_t cshift = as_snapshot ? 4u : 16u;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (P2[j]) {
                        emit_val<1, GMODE>(ex, sinkM, TT.offI[2] + E2 + n2, zm2[j], tid, lm);
                        lm += zm2[j] > 0xffu ? 1u : 0u;
                        if (MODE == EM_LIST)
                            sh.L2()[guard_pos(ex, E2 + n2, 1, 4 * C::NBLK, kGuardList2)] = ((uint32_t)tid << 2) | (uint32_t)j | (pre << 12);
                        n2++;
                        pre += (r.flags >> (cshift + 3 * j)) & 7u;
                    }
                }
            }
            guard_flush(ex);
        });
        };
        const uint32_t nI2 = ex.uni(TT.Ni[2]), nI1 = ex.uni(TT.Ni[1]);
        if (use_stash) {
            // ---- a Log whose values all fit two bytes, emitted from the stash: every byte of both Dacs is stored in
            // one visit of each source (EM_ONE), except the second bytes of the few nodes of heights >= 3 ----
            // Pass A and the two stash passes only write output bytes and OR bits into the LDS bitmaps: nothing one of them
            // produces is read by another, so a wave runs through all three without waiting for the others.
            if (!(K2R_DIAG_SKIP & 1)) passA(EmTag<EM_P0>{});
            ex.stamp(10);
            // one work item per I record: the four height-1 children of an internal height-2 node.  Records are in
            // arrival order; their place in level order comes from the owner's prefixes.
            // (two instantiations of each stash pass: when no record overflowed the pool -- 97 % of the instants, known from the
            // counts -- the loop body has no LDS-or-global branch)
            auto passI = [&](auto ovf_tag) {
              constexpr bool OVF = decltype(ovf_tag)::value != 0;
              ex.par_nosync([&](int tid, EncRegs&) {
                const uint32_t offV1 = TT.offV[1], offZ1 = TT.offZ[1], offI1 = TT.offI[1], lt = TT.LT, ne = TT.LT - TT.M0;
                const uint32_t lv1 = sh.pl.lngV[1], lm1 = sh.pl.lngM[1];
                for (uint32_t k = (uint32_t)tid; k < nI2; k += NT) {
                    uint32_t rec[5];
                    if (!OVF || k < (uint32_t)SH::CAPI_REC) {
#pragma unroll
                        for (int i = 0; i < 5; i++) rec[i] = sh.pool[5u * k + i];
                    } else {
                        gload_words<5>(ovI + 5u * (k - (uint32_t)SH::CAPI_REC), rec);
                    }
                    const uint32_t hdr = rec[0];
                    const uint32_t own = hdr & 1023u;
                    const uint32_t pf = sh.pfx[0][own], pl = sh.pfx[2][own];
                    const uint32_t kk = ((pf >> 16) & 0x3fffu) + ((hdr >> 10) & 3u);  // rank among internal height-2 nodes
                    const uint32_t pre = (pf & 0xffffu) + ((hdr >> 12) & 15u);        // internal quads before this node
                    const uint32_t tb1 = (hdr >> 16) & 15u, e4 = (hdr >> 20) & 15u, elen = 4u - popc32(tb1);
                    // the eqB bits of the T = 0 quads, packed to a run (first quad = most significant bit)
                    uint32_t erun = 0;
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) {
                        const uint32_t t0 = ((tb1 >> (3 - qq)) & 1u) ^ 1u;
                        erun = (erun << t0) | ((e4 >> (3 - qq)) & t0);
                    }
                    // second bytes of this block's height-1 values that come before this record's (EncShared::li)
                    const uint32_t below = sh.li[own] & ((1u << (8u * ((hdr >> 10) & 3u))) - 1u);
                    uint32_t w[4];
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) w[qq] = rec[1 + qq];
                    const uint32_t p1 = offV1 + 4 * kk;
                    emit4<0, EM_ONE>(ex, sinkV, p1, zz32(rec16(w[0] & 0xffffu)), zz32(rec16(w[1] & 0xffffu)),
                                     zz32(rec16(w[2] & 0xffffu)), zz32(rec16(w[3] & 0xffffu)), tid,
                                     lv1 + (pl & 0xffffu) + popc32(below & 0x0f0f0f0fu));
                    bm_or_run(ex, sh.bmT, guard_pos(ex, p1, 4, lt, kGuardTRun1), 4, tb1);
                    bm_or_run(ex, sh.bmE, guard_pos<false>(ex, offZ1 + 4 * kk - pre, elen, ne, kGuardE1), elen, erun);
                    uint32_t n1 = 0, lm = lm1 + (pl >> 16) + popc32(below & 0xf0f0f0f0u);
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) {
                        if ((tb1 >> (3 - qq)) & 1u) {
                            const uint32_t zm = zz32(rec16(w[qq] >> 16));
                            emit_val<1, EM_ONE>(ex, sinkM, offI1 + pre + n1, zm, tid, lm);
                            lm += zm > 0xffu ? 1u : 0u;
                            n1++;
                        }
                    }
                }
                guard_flush(ex);
              });
            };
            if (K2R_DIAG_SKIP & 2) {
            } else if (rec_ovf) passI(EmTag<1>{});
            else passI(EmTag<0>{});
            ex.stamp(11);
            // one work item per Q record: the four cells of an internal quad
            auto passQ = [&](auto ovf_tag) {
              constexpr bool OVF = decltype(ovf_tag)::value != 0;
              ex.par([&](int tid, EncRegs&) {
                const uint32_t offV0 = TT.offV[0], lv0 = sh.pl.lngV[0];
                for (uint32_t m = (uint32_t)tid; m < nI1; m += NT) {
                    uint32_t q[3];
                    if (!OVF || m < (uint32_t)SH::CAPQ_REC) {
#pragma unroll
                        for (int i = 0; i < 3; i++) q[i] = sh.pool[SH::QBASE + 3u * m + i];
                    } else {
                        gload_words<3>(ovQ + 3u * (m - (uint32_t)SH::CAPQ_REC), q);
                    }
                    const uint32_t hdr = q[0], a = q[1], b = q[2];
                    const uint32_t own = hdr & 1023u;
                    const uint32_t ord = (hdr >> 10) & 15u;
                    const uint32_t pos = (sh.pfx[0][own] & 0xffffu) + ord;  // rank among internal quads
                    // second bytes of this block's cell values that come before this record's (EncShared::lq)
                    const uint32_t f0 = sh.lq[own][0], f1 = sh.lq[own][1], cut = (1u << (4u * (ord & 7u))) - 1u;
                    const uint32_t before = ord < 8u ? popc32(f0 & cut) : popc32(f0) + popc32(f1 & cut);
                    emit4<0, EM_ONE>(ex, sinkV, offV0 + 4 * pos, zz32(rec16(a & 0xffffu)), zz32(rec16(a >> 16)),
                                     zz32(rec16(b & 0xffffu)), zz32(rec16(b >> 16)), tid, lv0 + sh.pfx[1][own] + before);
                }
                guard_flush(ex);
              });
            };
            if (K2R_DIAG_SKIP & 4) ex.barrier();
            else if (rec_ovf) passQ(EmTag<1>{});
            else passQ(EmTag<0>{});
            ex.stamp(6);
            // T, eqB and the continuation bitmaps of both Dacs (serialized; rank prefixes kept for the top nodes)
            {
                const uint32_t lt = ex.uni(TT.LT), m0 = ex.uni(TT.M0);
                const BmJob jobs[4] = {{sh.bmT, lt, nullptr, 0, io + 13},
                                       {sh.bmE, lt - m0, nullptr, 0, io + log_eq_off},
                                       {sh.bmV0, sinkV.n0, sh.prefTopV, SH::PREFTOP, nlevV > 0 ? io + ex.uni(DV.bm_off[0]) : nullptr},
                                       {sh.bmM[0], sinkM.n0, sh.prefM, SH::PREFTOP, nlevM > 0 ? io + ex.uni(DM.bm_off[0]) : nullptr}};
                if (!(K2R_DIAG_SKIP & 8)) bitmaps_finish<C, 4>(ex, jobs);
            }
            ex.stamp(8);
            if (nlevV > 1 || nlevM > 1) {
                // second bytes of the nodes of heights >= 3, replayed from the registers pass A left behind
                if (!(K2R_DIAG_SKIP & 16)) ex.par_nosync([&](int tid, EncRegs& r) {
                    auto put = [&](const DacSink& d, uint32_t* bm0, uint32_t pos, uint32_t hi) {
                        if (hi) {
                            pos = guard_pos(ex, pos, 1, d.n0 < 32u * SH::PREFTOP ? d.n0 : 32u * SH::PREFTOP, d.code);
                            gstore8(d.plane1 + guard_pos(ex, bm_rank(bm0, d.pref, pos), 1, d.n1, d.code + 1), (uint8_t)hi);
                        }
                    };
                    const uint32_t hb = r.pa[2];
                    put(sinkV, sh.bmV0, r.pa[0], hb & 0xffu);
                    put(sinkM, sh.bmM[0], r.pa[1], (hb >> 8) & 0xffu);
                    put(sinkV, sh.bmV0, r.pa[3], (hb >> 16) & 0xffu);
                    put(sinkM, sh.bmM[0], r.pa[4], hb >> 24);
                    guard_flush(ex);
                });
                ex.stamp(14);
                if (nlevV > 1 && !(K2R_DIAG_SKIP & 32)) bitmap_write_zero<C>(ex, sinkV.n1, io + ex.uni(DV.bm_off[1]));
                if (nlevM > 1 && !(K2R_DIAG_SKIP & 32)) bitmap_write_zero<C>(ex, sinkM.n1, io + ex.uni(DM.bm_off[1]));
                // (no barrier: nothing the next instant's first phase writes is read above)
            }
            ex.stamp(9);
        } else {
            if (as_snapshot && s_cmp) compact_pass(inst, head && inst == 0);  // leave the compact copy of this snapshot instant for the logs that follow
            if (head && inst == 0) {  // ... and for the parts that continue this chunk
                ex.publish(ta.shared_flag, s_cmp ? PART_CMP : PART_NOCMP, (uint32_t)s_base);
                published = true;
            }
            passA(EmTag<EM_LIST>{});
            ex.barrier();  // pass B consumes the work list pass A built
            ex.stamp(10);  // emission pass A (own/top nodes, height-2 groups, work list)
            // 5b'. one work item per internal height-2 node (dense, level order): its four height-1 children
            ex.par([&](int tid, EncRegs&) {
                int32_t lerr = 0;  // loads were validated in phase 1
                for (uint32_t k = (uint32_t)tid; k < nI2; k += NT) {
                    const uint32_t ent = sh.L2()[k];
                    const uint32_t blk = (ent >> 2) & (C::NBLK - 1), j = ent & 3u, pre = ent >> 12;
                    uint32_t r0, c0;
                    blk_origin((int)blk, r0, c0);
                    const uint32_t rj = r0 + 4 * (j >> 1), cj = c0 + 4 * (j & 1);
                    int32_t t16[16];
                    load_sub16<PADDED, VEC>(ta, inst, r0, c0, (int)j, t16, lerr);
                    int32_t mn1[4], mx1[4];
                    bool inv1[4];
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) {
                        mn1[qq] = min4(t16[4 * qq], t16[4 * qq + 1], t16[4 * qq + 2], t16[4 * qq + 3]);
                        mx1[qq] = max4(t16[4 * qq], t16[4 * qq + 1], t16[4 * qq + 2], t16[4 * qq + 3]);
                        inv1[qq] = inval(rj + 2 * (qq >> 1), cj + 2 * (qq & 1));
                    }
                    uint32_t z1v[4], zm1[4], tb1 = 0, erun = 0, elen = 0;
                    bool P1[4];
                    if (as_snapshot) {
                        const int32_t mn2 = min4(mn1[0], mn1[1], mn1[2], mn1[3]);
                        const int32_t mx2 = max4(mx1[0], mx1[1], mx1[2], mx1[3]);
#pragma unroll
                        for (int qq = 0; qq < 4; qq++) {
                            P1[qq] = !inv1[qq] && mn1[qq] != mx1[qq];
                            z1v[qq] = zz32(inv1[qq] ? mx2 : mx2 - mx1[qq]);
                            zm1[qq] = zz32(mn1[qq] - mn2);
                            tb1 = (tb1 << 1) | (P1[qq] ? 1u : 0u);
                        }
                    } else {
                        int32_t s16[16];
                        load_sub16<PADDED, VEC>(ta, s_idx, r0, c0, (int)j, s16, lerr);  // L2 / Infinity Cache hit
#pragma unroll
                        for (int qq = 0; qq < 4; qq++) {
                            const uint32_t rq = rj + 2 * (qq >> 1), cq = cj + 2 * (qq & 1);
                            int32_t d[4];
#pragma unroll
                            for (int i = 0; i < 4; i++)
                                d[i] = inval(rq + (i >> 1), cq + (i & 1)) ? 0 : t16[4 * qq + i] - s16[4 * qq + i];
                            const int32_t smn1 = min4(s16[4 * qq], s16[4 * qq + 1], s16[4 * qq + 2], s16[4 * qq + 3]);
                            const int32_t smx1 = max4(s16[4 * qq], s16[4 * qq + 1], s16[4 * qq + 2], s16[4 * qq + 3]);
                            const bool eq1 = d[0] == d[1] && d[0] == d[2] && d[0] == d[3];
                            const bool unif1 = inv1[qq] || mn1[qq] == mx1[qq];
                            P1[qq] = !unif1 && !eq1;
                            z1v[qq] = zz32(inv1[qq] ? 0 : mx1[qq] - smx1);
                            zm1[qq] = zz32(mn1[qq] - smn1);
                            tb1 = (tb1 << 1) | (P1[qq] ? 1u : 0u);
                            if (!P1[qq]) {
                                erun = (erun << 1) | (unif1 ? 0u : 1u);
                                elen++;
                            }
                        }
                    }
                    const uint32_t p1 = TT.offV[1] + 4 * k;  // level order: four children per internal parent
                    emit4<0>(ex, sinkV, p1, z1v[0], z1v[1], z1v[2], z1v[3], tid);
                    bm_or_run(ex, sh.bmT, guard_pos(ex, p1, 4, TT.LT, kGuardTRun1), 4, tb1);
                    if (!as_snapshot) bm_or_run(ex, sh.bmE, guard_pos<false>(ex, TT.offZ[1] + 4 * k - pre, elen, TT.LT - TT.M0, kGuardE1), elen, erun);
                    uint32_t n1 = 0;
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) {
                        if (P1[qq]) {
                            emit_val<1>(ex, sinkM, TT.offI[1] + pre + n1, zm1[qq], tid);
                            sh.L1()[guard_pos(ex, pre + n1, 1, 16 * C::NBLK, kGuardList1)] = (uint16_t)((blk << 4) | (j << 2) | (uint32_t)qq);
                            n1++;
                        }
                    }
                }
                guard_flush(ex);
            });

            ex.stamp(11);  // emission pass B (height-1 groups)
            // 5b''. one work item per internal quad: its four cells
            ex.par([&](int tid, EncRegs&) {
                int32_t lerr = 0;
                // four quads per trip, their loads issued together (one load in flight per thread = a trip per memory latency)
                constexpr uint32_t U = 4;
                for (uint32_t m0 = (uint32_t)tid; m0 < nI1; m0 += U * NT) {
                    int32_t t4[U][4], s4[U][4];
                    uint32_t rq[U], cq[U];
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) {
                        const uint32_t m = m0 + u * NT;
                        if (m >= nI1) continue;
                        const uint32_t key = sh.L1()[m];
                        const uint32_t blk = key >> 4, j = (key >> 2) & 3u, qq = key & 3u;
                        uint32_t r0, c0;
                        blk_origin((int)blk, r0, c0);
                        rq[u] = r0 + 4 * (j >> 1) + 2 * (qq >> 1);
                        cq[u] = c0 + 4 * (j & 1) + 2 * (qq & 1);
                        load_quad<PADDED, VEC>(ta, inst, rq[u], cq[u], t4[u], lerr);
                        if (!as_snapshot) load_quad<PADDED, VEC>(ta, s_idx, rq[u], cq[u], s4[u], lerr);
                    }
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) {
                        const uint32_t m = m0 + u * NT;
                        if (m >= nI1) continue;
                        uint32_t z[4];
                        if (as_snapshot) {
                            const int32_t mx1 = max4(t4[u][0], t4[u][1], t4[u][2], t4[u][3]);
#pragma unroll
                            for (int i = 0; i < 4; i++) z[i] = zz32(inval(rq[u] + (i >> 1), cq[u] + (i & 1)) ? mx1 : mx1 - t4[u][i]);
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; i++) z[i] = zz32(inval(rq[u] + (i >> 1), cq[u] + (i & 1)) ? 0 : t4[u][i] - s4[u][i]);
                        }
                        emit4<0>(ex, sinkV, TT.offV[0] + 4 * m, z[0], z[1], z[2], z[3], tid);
                    }
                }
                guard_flush(ex);
            });
            ex.stamp(as_snapshot ? 5 : 6);  // plane-0 emission (5: snapshot, 6: log)
            // 5c. bitmaps + higher planes
            bitmap_finish_write<C>(ex, sh.bmT, ex.uni(TT.LT), sh.prefM, io + 13);
            if (!as_snapshot) bitmap_finish_write<C>(ex, sh.bmE, ex.uni(TT.LT) - ex.uni(TT.M0), sh.prefM, io + log_eq_off);
            ex.stamp(7);  // T / eqB bitmaps
            auto uni_layout = [&](const DacLayout& L) {  // wave-uniform copy (dac_finish branches on it around barriers)
                DacLayout U;
#pragma unroll
                for (int i = 0; i < 5; i++) U.n[i] = ex.uni(L.n[i]);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    U.bm_off[i] = ex.uni(L.bm_off[i]);
                    U.by_off[i] = ex.uni(L.by_off[i]);
                }
                U.nlev = ex.uni(L.nlev);
                U.end = ex.uni(L.end);
                return U;
            };
            dac_finish<C>(ex, uni_layout(DV), io, sh.bmV0, sh.bmV1(), sh.prefV(), listV, &sh.nlistV);
            ex.stamp(8);  // Lmax Dac: bitmaps + planes >= 1
            dac_finish<C>(ex, uni_layout(DM), io, sh.bmM[0], sh.bmM[1], sh.prefM, listM, &sh.nlistM);
            ex.stamp(9);  // Lmin Dac: planes >= 1
        }

        off += isize;
        blk_count++;
    }

    if (head && !published) ex.publish(ta.shared_flag, PART_FAILED, 0u);  // (an error before instant 0 was emitted)

    // ---- close the last block, chunk header (chunk.rs:76-78, 238) ----
    ex.par([&](int tid, EncRegs&) {
        if (tid == 0) {
            if (status == ST_OK) {
                if (blk_hdr == NO_HDR) carry = blk_count;  // the inherited block never closed
                else out[blk_hdr] = (uint8_t)blk_count;
                if (!cont) store_be32(out + 2, n_blocks + 1);
            }
            res->carry_count = carry;
            const bool faulted = sh.fault[0] != 0;
            res->status = faulted ? (int32_t)ST_INTERNAL : status;
            res->snapshots = n_snap;
            res->logs = n_log;
            res->stash_logs = n_stash;
            res->len = (status == ST_OK && !faulted) ? off : 0;
            for (int i = 0; i < 6; i++) res->dbg[i] = sh.fault[i];
            for (int i = 0; i < NPROF; i++) res->prof[i] = sh.prof[i];
        }
#ifdef K2R_PROFILE
        if (!EX::kSim && (tid & 63) == 0)
            for (int k = 0; k < NPROF; k++) {
                res->pw[tid >> 6][k][0] = sh.pw[tid >> 6][k][0];
                res->pw[tid >> 6][k][1] = sh.pw[tid >> 6][k][1];
            }
#endif
    });
}

}  // namespace k2r
