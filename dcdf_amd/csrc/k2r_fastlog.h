// k2r_fastlog.h -- level-order streaming emission of a Log (included by k2r_encode.h).
//
// The general path of k2r_encode.h gives every thread an 8x8 block and learns stream positions from a workgroup scan,
// so everything a Log needs below height 2 has to be parked (the "stash") until the scan is done, and is then decoded
// and scattered by three more passes.  This path removes that round trip for the common instant -- a Log whose values
// fit 16 bits, taken against a snapshot whose compact copy exists -- by choosing the thread mapping so that a wave
// PRODUCES ITS VALUES IN LEVEL ORDER:
//
//   * lane = one height-2 node (4x4 cells); a wave-iteration covers 64 Morton-consecutive height-2 nodes (32x32 cells,
//     one height-5 node), a wave makes 4 iterations (one height-6 node), the waves of the workgroup follow each other
//     in Morton order.  Level order (snapshot.rs:126-146 BFS with child order i*k+j) restricted to one level is then
//     exactly (wave, iteration, lane, local order).
//   * heights 3, 4, 5 are groups of 4 / 16 / 64 lanes: their min / max / "equal" come from DPP group reductions and one
//     ballot, inside the iteration, so a node's "visited" and "internal" flags (log.rs:137-152) are known while its
//     values are still in registers.
//   * every value of levels 0..4 is appended, as a 16-bit zig-zag code, to the wave's piece of an LDS staging pool at
//     [running count of the wave] + [rank inside the iteration] (ballot + mbcnt; one DPP scan for the quads).  No
//     owner tags, no per-thread prefix tables, no atomics on the data path (one pool allocation per wave-iteration).
//   * after ONE barrier a single wave finishes heights 5..H (85 nodes at sidelen 256), turns the 16 waves' counts into
//     level offsets, sizes both candidates (A.8) and applies chunk.rs:62.  When the Log wins, each
//     wave copies its own pieces out: plane-0 bytes, continuation bits by ballot, second bytes by rank (dac.rs:96-132),
//     T / eqB bit runs by funnel shift (bitmap.rs:44-62).
//
// Anything else -- the instant becomes a Snapshot, a value needs more than two bytes, the cells do not fit the compact
// 16-bit window of the block's snapshot, the pool overflows, a conversion error -- returns false and the general path
// encodes the instant (bit-identical by construction: same decision function, same formats).
#pragma once

namespace k2r {

template <class C>
struct FastCfg {
    static constexpr int H = C::H;
    static constexpr int NW = C::NW;
    static constexpr int N5 = NW * 4;  // height-5 nodes == wave-iterations
    // (used for sidelen >= 64 only: N5 == 4^(H-5))
    enum : int {  // per-wave totals handed to the planner
        W_Q1 = 0, W_I2, W_I3, W_I4, W_I5, W_LV0, W_LV1, W_LV2, W_LV3, W_LV4, W_LM1, W_LM2, W_LM3, W_LM4, W_S1, W_S2, W_S3, W_S4, W_S5, W_N  // (19; tables are padded to 20)
    };
};

template <class C, int STAGE_WORDS>
struct FastShared {
    using F = FastCfg<C>;
    uint32_t stage[STAGE_WORDS];           // 16-bit zig-zag codes, pieces of 8-byte granularity
    uint32_t piece[F::N5][8];              // per wave-iteration: base (u16 index), q1, i2, i3, i4, i5
    int32_t t5[F::N5][6];                  // height-5 summaries: min_t, max_t, min_s, max_s, diff, equal
    uint32_t wt[F::NW][F::W_N + 1];        // per-wave totals
    // the plan (written by wave 0)
    uint32_t ok, isize, lt, m0, n0, n1v, n1m, eq_off;
    uint32_t vbm0, vby0, vbm1, vby1, vnlev, mbm0, mby0, mbm1, mby1, mnlev;
    uint32_t wex[F::NW][20];               // per wave: exclusive prefix over the waves of every wt field
    uint32_t offV[5], offI[5], offZ[5];    // level offsets of heights 0..4 (Lmax/T, Lmin, eqB index of the first node)
    uint32_t lvb[5], lmb[5];               // second bytes of the Lmax / Lmin Dac that come before height h's
};

// ---- packed 16-bit helpers (two cells per register) --------------------------------------------------------
K2R_HD uint32_t pk_sub16(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    const s16x2 r = __builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b);
    return __builtin_bit_cast(uint32_t, r);
#else
    return ((a - b) & 0xffffu) | (((a >> 16) - (b >> 16)) << 16);
#endif
}
// zig-zag of two int16 lanes (dac.rs:134-137 on values that fit 16 bits)
K2R_HD uint32_t pk_zz16(uint32_t d) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    const s16x2 v = __builtin_bit_cast(s16x2, d);
    const s16x2 r = (v << (short)1) ^ (v >> (short)15);
    return __builtin_bit_cast(uint32_t, r);
#else
    const int32_t lo = (int16_t)(d & 0xffffu), hi = (int16_t)(d >> 16);
    return (uint32_t)(((lo << 1) ^ (lo >> 15)) & 0xffff) | ((uint32_t)(((hi << 1) ^ (hi >> 15)) & 0xffff) << 16);
#endif
}
K2R_HD uint32_t rot16(uint32_t x) { return (x >> 16) | (x << 16); }
// number of 16-bit halves of x above 0xff (values needing a second byte), as 0..2
K2R_HD uint32_t pk_long_count(uint32_t x) {
    const uint32_t t = (x >> 8) & 0x00ff00ffu;
    const uint32_t nz = ((t + 0x00ff00ffu) >> 8) & 0x00010001u;
    return (nz + (nz >> 16)) & 3u;
}
K2R_HD int32_t sext16(uint32_t x) { return (int32_t)(int16_t)(x & 0xffffu); }
K2R_HD uint32_t brev32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bitreverse32(x);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
    return __builtin_bswap32(x);
#endif
}

// Loads the 8 words (16 cells as uint16 offsets, local Morton order) of height-2 node m2 of the compact snapshot copy.
K2R_HD void load_compact_node(const uint32_t* scmp, uint32_t m2, uint32_t (&w)[8]) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(1))) const char* gptr;
    const uint32_t ob = m2 * 32u;
    const uint4 a = *(__attribute__((address_space(1))) const uint4*)((gptr)scmp + ob);
    const uint4 b = *(__attribute__((address_space(1))) const uint4*)((gptr)scmp + ob + 16);
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
#else
    for (int i = 0; i < 8; i++) w[i] = scmp[(size_t)m2 * 8 + i];
#endif
}


// Height-2 node (rows rj.., cols cj..) of instant `inst`, for the 4-byte element types with 16-byte rows (VEC 1, 2): the four
// raw 16-byte rows, requested here and consumed one iteration later (software prefetch) ...
template <int VEC>
K2R_HD void load_rows_raw(const TileArgs& ta, uint32_t inst, uint32_t rj, uint32_t cj, int32_t (&raw)[16]) {
    const int32_t* ib = (const int32_t*)ta.base + (int64_t)inst * ta.st;
    const uint32_t o0 = rj * (uint32_t)ta.sr + cj;
#pragma unroll
    for (int dr = 0; dr < 4; dr++) {
        const uint32_t o = o0 + (uint32_t)dr * (uint32_t)ta.sr;
#if defined(__HIP_DEVICE_COMPILE__)
        typedef __attribute__((address_space(1))) const char* gptr;
        const int4 a = *(__attribute__((address_space(1))) const int4*)((gptr)ib + (o << 2));
        raw[4 * dr] = a.x; raw[4 * dr + 1] = a.y; raw[4 * dr + 2] = a.z; raw[4 * dr + 3] = a.w;
#else
        for (int i = 0; i < 4; i++) raw[4 * dr + i] = ib[o + i];
#endif
    }
}
// ... and their conversion to stored values in local Morton order (cell_m), as load_sub16 delivers them
template <int VEC>
K2R_HD void rows_to_stored(const TileArgs& ta, const int32_t (&raw)[16], int32_t (&dst)[16], int32_t& err) {
#pragma unroll
    for (int dr = 0; dr < 4; dr++) {
        int32_t v[4] = {raw[4 * dr], raw[4 * dr + 1], raw[4 * dr + 2], raw[4 * dr + 3]};
        if (VEC == 2) {
            const float f[4] = {as_f32(v[0]), as_f32(v[1]), as_f32(v[2]), as_f32(v[3])};
            fixed4_f32(f, ta, (float)((int64_t)1 << ta.fbits), v, err);
        }
#pragma unroll
        for (int dc = 0; dc < 4; dc++) dst[cell_m(dr, dc)] = v[dc];
    }
}

// ORs a run of `len` (1..32) bits, first bit most significant of the len right-aligned bits, at bit position p of an
// MSB-first LDS bit array (same convention as bm_or_run, without the workgroup-exec indirection)
template <class EX>
K2R_HD void te_or_run(EX& ex, uint32_t* bm, uint32_t p, uint32_t len, uint32_t bits) {
    if (len == 0 || bits == 0) return;
    const uint64_t v = (uint64_t)bits << (64 - len);
    const uint64_t w = v >> (p & 31);
    const uint32_t hi = (uint32_t)(w >> 32), lo = (uint32_t)w;
    if (hi) ex.lds_or_nr(&bm[p >> 5], hi);
    if (lo) ex.lds_or_nr(&bm[(p >> 5) + 1], lo);
}

// Tries to encode instant `inst` as a Log against the block's snapshot (compact copy `scmp`, base `s_base`).
// Returns true when the Log was chosen by chunk.rs:62 and written at `io` (isize = its serialized size).
// False: nothing was written that matters; the caller runs the general path for this instant.
template <class C, int VEC, class EX>
K2R_HD bool fast_log_instant(EX& ex, const TileArgs& ta, uint32_t inst, const uint32_t* scmp, int32_t s_base, uint8_t* io,
                             uint64_t room, bool cap254, uint32_t& isize_out) {
    using F = FastCfg<C>;
    constexpr int H = C::H;
    constexpr int NW = F::NW;
    constexpr int N5 = F::N5;
    auto& sh = ex.sh;
    auto& fs = sh.fast;
    // 16-bit codes the staging pool holds (ta.stash_words shrinks it: tests of the overflow fallback)
    const uint32_t STAGE16 = (ta.stash_words != 0 && 2u * ta.stash_words < (uint32_t)(sizeof(fs.stage) / 2)) ? 2u * ta.stash_words
                                                                                                         : (uint32_t)(sizeof(fs.stage) / 2);

    ex.simt([&](int tid) {
        const int lane = tid & 63;
        const int wave = ex.uni(tid >> 6);
        uint16_t* const st16 = (uint16_t*)fs.stage;
        // ---- clear the final bitmaps (read by the previous instant's serializer until its closing barrier) ----
        for (uint32_t w = (uint32_t)tid; w <= (uint32_t)C::WT; w += C::NT) {
            sh.bmT[w] = 0;
            sh.bmE[w] = 0;
            sh.bmM[0][w] = 0;
        }
        for (uint32_t w = (uint32_t)tid; w <= (uint32_t)C::WV; w += C::NT) sh.bmV0[w] = 0;

        // ================= phase 1: four iterations of 64 height-2 nodes each =================
        uint32_t r_q1 = 0, r_i2 = 0, r_i3 = 0, r_i4 = 0, r_i5 = 0;  // internal nodes of heights 1..5 so far (this wave)
        uint32_t LV2 = 0, LM2 = 0, LV3 = 0, LM3 = 0, LV4 = 0, LM4 = 0;
        uint32_t s1l = 0, S2 = 0, S3 = 0, S4 = 0, S5 = 0;           // snapshot candidate: internal nodes per height
        uint32_t bad = 0;
        int32_t err = 0;
        // Software prefetch (4-byte element types): the rows of iteration it + 1 are requested before iteration it is worked
        // on, so the wave computes while its next cells are in flight; the four iterations are unrolled so that the two
        // register sets alternate without copies.
        constexpr bool kPrefetch = VEC == 1 || VEC == 2;
        int32_t rawA[16], rawB[16];
        uint32_t spA[8], spB[8];
        auto request = [&](int it, int32_t (&raw)[16], uint32_t (&spn)[8]) {
            const uint32_t m2 = ((uint32_t)wave * 4u + (uint32_t)it) * 64u + (uint32_t)lane;
            uint32_t br, bc;
            morton_decode(m2 >> 2, br, bc);
            load_rows_raw<VEC>(ta, inst, br * 8 + 4 * ((m2 >> 1) & 1u), bc * 8 + 4 * (m2 & 1u), raw);
            load_compact_node(scmp, m2, spn);
        };
        if (kPrefetch) request(0, rawA, spA);
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const uint32_t n5 = (uint32_t)wave * 4u + (uint32_t)it;
            const uint32_t m2 = n5 * 64u + (uint32_t)lane;
            int32_t t16[16];
            uint32_t sp[8];
            if (kPrefetch) {
                if (it < 3) request(it + 1, (it & 1) ? rawA : rawB, (it & 1) ? spA : spB);
                rows_to_stored<VEC>(ta, (it & 1) ? rawB : rawA, t16, err);
#pragma unroll
                for (int i = 0; i < 8; i++) sp[i] = (it & 1) ? spB[i] : spA[i];
            } else {
                uint32_t br, bc;
                morton_decode(m2 >> 2, br, bc);
                load_sub16<false, VEC>(ta, inst, br * 8, bc * 8, (int)(m2 & 3u), t16, err);
                load_compact_node(scmp, m2, sp);
            }
            // ---- height 0 -> 1: four quads ----
            uint32_t zz01[4], zz23[4], zzx1p[2] = {0, 0}, zzn1[4];
            uint32_t tb1 = 0, f1w = 0, unifm = 0, dfirst[4];  // f1w: per quad 2 = internal, 1 = "equal" (not uniform), 0 = uniform
            bool eq1all = true;
            int32_t mn2 = 0, mx2 = 0, smn2 = 0, smx2 = 0;
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                const int32_t a0 = t16[4 * qq] - s_base, a1 = t16[4 * qq + 1] - s_base, a2 = t16[4 * qq + 2] - s_base,
                              a3 = t16[4 * qq + 3] - s_base;
                bad |= (uint32_t)(a0 | a1 | a2 | a3) >> 16;  // outside the 16-bit window of the compact copy
                const uint32_t tp0 = ((uint32_t)a0 & 0xffffu) | ((uint32_t)a1 << 16), tp1 = ((uint32_t)a2 & 0xffffu) | ((uint32_t)a3 << 16);
                const uint32_t sp0 = sp[2 * qq], sp1 = sp[2 * qq + 1];
                const uint32_t d01 = pk_sub16(tp0, sp0), d23 = pk_sub16(tp1, sp1);  // log.rs:751 (exact when the node is narrow)
                const bool eq1 = ((d01 ^ d23) | (d01 ^ rot16(d01))) == 0;           // log.rs:780,805
                const int32_t mn1 = min4(a0, a1, a2, a3), mx1 = max4(a0, a1, a2, a3);
                const int32_t s0 = (int32_t)(sp0 & 0xffffu), s1 = (int32_t)(sp0 >> 16), s2 = (int32_t)(sp1 & 0xffffu), s3 = (int32_t)(sp1 >> 16);
                const int32_t smn1 = min4(s0, s1, s2, s3), smx1 = max4(s0, s1, s2, s3);
                const bool unif1 = mn1 == mx1;
                const bool P1L = !unif1 && !eq1;  // log.rs:137-152
                zz01[qq] = pk_zz16(d01);
                zz23[qq] = pk_zz16(d23);
                const uint32_t zx = zz32(mx1 - smx1) & 0xffffu;  // log.rs:133
                zzn1[qq] = zz32(mn1 - smn1) & 0xffffu;            // log.rs:148
                zzx1p[qq >> 1] |= zx << (16 * (qq & 1));
                tb1 = (tb1 << 1) | (P1L ? 1u : 0u);
                f1w |= (P1L ? 2u : (unif1 ? 0u : 1u)) << (8 * qq);  // log.rs:137-152
                unifm |= (unif1 ? 1u : 0u) << qq;
                dfirst[qq] = d01 & 0xffffu;
                eq1all = eq1all && eq1;
                mn2 = qq == 0 ? mn1 : (mn1 < mn2 ? mn1 : mn2);
                mx2 = qq == 0 ? mx1 : (mx1 > mx2 ? mx1 : mx2);
                smn2 = qq == 0 ? smn1 : (smn1 < smn2 ? smn1 : smn2);
                smx2 = qq == 0 ? smx1 : (smx1 > smx2 ? smx1 : smx2);
            }
            // every value of this node lies in [mn2 - smx2, mx2 - smn2]; group values lie in the union of their lanes' ranges
            bad |= (mn2 - smx2 < -32768 || mx2 - smn2 > 32767) ? 1u : 0u;
            const bool eq2 = eq1all && dfirst[0] == dfirst[1] && dfirst[0] == dfirst[2] && dfirst[0] == dfirst[3];
            const int32_t df2 = sext16(dfirst[0]);
            const bool unif2 = mn2 == mx2;
            const bool P2L = !unif2 && !eq2;
            const uint32_t zzx2 = zz32(mx2 - smx2) & 0xffffu, zzn2 = zz32(mn2 - smn2) & 0xffffu;
            // ---- heights 3, 4, 5: groups of 4 / 16 / 64 lanes ----
            const int32_t mn3 = ex.template w_gmin<4>(tid, mn2), mx3 = ex.template w_gmax<4>(tid, mx2);
            const int32_t smn3 = ex.template w_gmin<4>(tid, smn2), smx3 = ex.template w_gmax<4>(tid, smx2);
            const int32_t mn4 = ex.w_gmin16_from4(tid, mn3), mx4 = ex.w_gmax16_from4(tid, mx3);
            const int32_t smn4 = ex.w_gmin16_from4(tid, smn3), smx4 = ex.w_gmax16_from4(tid, smx3);
            int32_t mn5 = ex.w_lane(tid, mn4, 0), mx5 = ex.w_lane(tid, mx4, 0), smn5 = ex.w_lane(tid, smn4, 0), smx5 = ex.w_lane(tid, smx4, 0);
#pragma unroll
            for (int r = 16; r < 64; r += 16) {
                const int32_t a = ex.w_lane(tid, mn4, r), b = ex.w_lane(tid, mx4, r), c = ex.w_lane(tid, smn4, r), d = ex.w_lane(tid, smx4, r);
                mn5 = a < mn5 ? a : mn5;
                mx5 = b > mx5 ? b : mx5;
                smn5 = c < smn5 ? c : smn5;
                smx5 = d > smx5 ? d : smx5;
            }
            // "equal" of a group: all members equal and all their first-cell diffs the same (log.rs:780-781,805)
            const uint64_t bne = ex.w_ballot(tid, df2 != ex.w_prev(tid, df2));
            const uint64_t beq = ex.w_ballot(tid, eq2);
            const uint64_t ok4 = beq & ~(bne & 0xEEEEEEEEEEEEEEEEull);
            uint64_t y4 = ok4 & (ok4 >> 1);
            y4 &= y4 >> 2;
            const uint64_t ok16 = beq & ~(bne & 0xFFFEFFFEFFFEFFFEull);
            uint64_t y16 = ok16 & (ok16 >> 1);
            y16 &= y16 >> 2;
            y16 &= y16 >> 4;
            y16 &= y16 >> 8;
            const int32_t dfa = ex.w_lane(tid, df2, 0), dfb = ex.w_lane(tid, df2, 16), dfc = ex.w_lane(tid, df2, 32), dfd = ex.w_lane(tid, df2, 48);
            const bool eq5 = ok16 == ~0ull && dfa == dfb && dfa == dfc && dfa == dfd;
            const bool eq3 = (y4 >> (lane & ~3)) & 1ull, eq4 = (y16 >> (lane & ~15)) & 1ull;
            const bool unif3 = mn3 == mx3, unif4 = mn4 == mx4, unif5 = mn5 == mx5;
            const bool P3L = !unif3 && !eq3, P4L = !unif4 && !eq4, P5L = !unif5 && !eq5;
            const bool lead4 = (lane & 3) == 0, lead16 = (lane & 15) == 0;
            const uint32_t zzx3 = zz32(mx3 - smx3) & 0xffffu, zzn3 = zz32(mn3 - smn3) & 0xffffu;
            const uint32_t zzx4 = zz32(mx4 - smx4) & 0xffffu, zzn4 = zz32(mn4 - smn4) & 0xffffu;
            // ---- counts and ranks ----
            const uint32_t nq = popc32(tb1);
            const uint32_t incl = ex.w_iscan(tid, nq);
            const uint32_t excl = incl - nq;
            const uint32_t q1 = (uint32_t)ex.w_lane(tid, (int32_t)incl, 63);
            const uint64_t b2 = ex.w_ballot(tid, P2L);
            const uint64_t b3a = ex.w_ballot(tid, P3L);            // lanes of internal height-3 nodes == visited height-2 nodes
            const uint64_t b3q = b3a & 0x1111111111111111ull;     // their leaders
            const uint64_t b4a = ex.w_ballot(tid, P4L);
            const uint64_t bl3 = b4a & 0x1111111111111111ull;     // leaders of visited height-3 nodes
            const uint64_t b4 = b4a & 0x0001000100010001ull;
            const uint32_t i2 = popc64(b2), i3 = popc64(b3q), i4 = popc64(b4), i5 = P5L ? 1u : 0u;
            const uint32_t rk2 = ex.w_mbcnt(tid, b2), rk3a = ex.w_mbcnt(tid, b3a), rk3q = ex.w_mbcnt(tid, b3q),
                           rkl3 = ex.w_mbcnt(tid, bl3), rk4 = ex.w_mbcnt(tid, b4);
            // ---- the wave's piece of the staging pool (units of 4 codes = 8 bytes): Lmax codes of levels 0..4, one flag
            //      byte per visited node of levels 1..4, Lmin codes of levels 1..4 ----
            const uint32_t oV1 = 4 * q1, oV2 = oV1 + 4 * i2, oV3 = oV2 + 4 * i3, oV4 = oV3 + 4 * i4, oF1 = oV4 + 4 * i5,
                           oF2 = oF1 + 2 * i2, oF3 = oF2 + 2 * i3, oF4 = oF3 + 2 * i4, oM1 = oF4 + 2 * i5,
                           oM2 = oM1 + q1, oM3 = oM2 + i2, oM4 = oM3 + i3, total = oM4 + i4;
            const uint32_t units = (total + 3) / 4;
            uint32_t base = 0;
            if (lane == 0) base = ex.lds_add(&sh.f_top, units);
            base = (uint32_t)ex.w_first(tid, (int32_t)base) * 4u;
            const bool fits = base + units * 4u <= STAGE16;
            if (!fits) bad |= 2u;
            if (lane == 0) {
                uint32_t* pc = fs.piece[n5];
                pc[0] = base; pc[1] = q1; pc[2] = i2; pc[3] = i3; pc[4] = i4; pc[5] = i5;
                int32_t* t = fs.t5[n5];
                t[0] = mn5; t[1] = mx5; t[2] = smn5; t[3] = smx5; t[4] = dfa; t[5] = eq5 ? 1 : 0;
            }
            if (fits) {
                uint16_t* const pb = st16 + base;
                // level 0 (cells of internal quads) and the Lmin of those quads
                uint32_t ord = excl;
#pragma unroll
                for (int qq = 0; qq < 4; qq++) {
                    if ((tb1 >> (3 - qq)) & 1u) {
                        uint32_t* p = (uint32_t*)(pb + 4 * ord);
                        p[0] = zz01[qq];
                        p[1] = zz23[qq];
                        pb[oM1 + ord] = (uint16_t)zzn1[qq];
                        ord++;
                    }
                }
                if (P2L) {  // level 1: the four quads of an internal height-2 node
                    uint32_t* p = (uint32_t*)(pb + oV1 + 4 * rk2);
                    p[0] = zzx1p[0];
                    p[1] = zzx1p[1];
                    ((uint32_t*)(pb + oF1))[rk2] = f1w;
                    pb[oM2 + rk2] = (uint16_t)zzn2;
                }
                if (P3L) {  // level 2: this lane's node is visited
                    pb[oV2 + rk3a] = (uint16_t)zzx2;
                    ((uint8_t*)(pb + oF2))[rk3a] = (uint8_t)(P2L ? 2u : (unif2 ? 0u : 1u));
                }
                if (lead4) {
                    if (P3L) pb[oM3 + rk3q] = (uint16_t)zzn3;
                    if (P4L) {  // level 3: visited height-3 node
                        pb[oV3 + rkl3] = (uint16_t)zzx3;
                        ((uint8_t*)(pb + oF3))[rkl3] = (uint8_t)(P3L ? 2u : (unif3 ? 0u : 1u));
                    }
                }
                if (lead16) {
                    if (P4L) pb[oM4 + rk4] = (uint16_t)zzn4;
                    if (P5L) {  // level 4: visited height-4 node
                        const uint32_t r = (uint32_t)lane >> 4;
                        pb[oV4 + r] = (uint16_t)zzx4;
                        ((uint8_t*)(pb + oF4))[r] = (uint8_t)(P4L ? 2u : (unif4 ? 0u : 1u));
                    }
                }
            }
            // long values of the group levels, snapshot candidate's internal nodes (snapshot.rs:133)
            LV2 += popc64(ex.w_ballot(tid, P3L && zzx2 > 0xffu));
            LM2 += popc64(ex.w_ballot(tid, P2L && zzn2 > 0xffu));
            LV3 += popc64(ex.w_ballot(tid, lead4 && P4L && zzx3 > 0xffu));
            LM3 += popc64(ex.w_ballot(tid, lead4 && P3L && zzn3 > 0xffu));
            LV4 += popc64(ex.w_ballot(tid, lead16 && P5L && zzx4 > 0xffu));
            LM4 += popc64(ex.w_ballot(tid, lead16 && P4L && zzn4 > 0xffu));
            s1l += 4u - popc32(unifm);  // (per lane; summed over the wave after the loop)
            S2 += popc64(ex.w_ballot(tid, !unif2));
            S3 += popc64(ex.w_ballot(tid, lead4 && !unif3));
            S4 += popc64(ex.w_ballot(tid, lead16 && !unif4));
            S5 += unif5 ? 0u : 1u;
            r_q1 += q1; r_i2 += i2; r_i3 += i3; r_i4 += i4; r_i5 += i5;
        }
        {   // values of levels 0 and 1 that need a second byte: counted over the wave's own staged codes (two per lane and step)
            ex.w_fence(tid);
            uint32_t LV0 = 0, LV1 = 0, LM1 = 0;
            auto count_long = [&](const uint16_t* src, uint32_t n, uint32_t& acc) {  // src 4-byte aligned
                const uint32_t* w32 = (const uint32_t*)src;
                for (uint32_t i0 = 0; i0 < n; i0 += 128) {
                    const uint32_t k = i0 + 2u * (uint32_t)lane;
                    const uint32_t x = k < n ? w32[k >> 1] : 0u;
                    acc += popc64(ex.w_ballot(tid, (x & 0xff00u) != 0)) + popc64(ex.w_ballot(tid, k + 1 < n && (x >> 24) != 0));
                }
            };
            const bool okw = ex.w_ballot(tid, (bad & 2u) != 0) == 0;  // every piece of this wave was allocated
            if (okw) {
#pragma unroll 1
                for (int it = 0; it < 4; it++) {
                    const uint32_t* pc = fs.piece[wave * 4 + it];
                    const uint32_t base = ex.uni(pc[0]), q1 = ex.uni(pc[1]), i2 = ex.uni(pc[2]), i3 = ex.uni(pc[3]), i4 = ex.uni(pc[4]), i5 = ex.uni(pc[5]);
                    count_long(st16 + base, 4 * q1, LV0);
                    count_long(st16 + base + 4 * q1, 4 * i2, LV1);
                    // (every V piece is a multiple of 4 codes, so the quads' Lmin codes start 4-byte aligned too)
                    count_long(st16 + base + 4 * (q1 + i2 + i3 + i4 + i5) + 2 * (i2 + i3 + i4 + i5), q1, LM1);
                }
            }
            const uint64_t anybad = ex.w_ballot(tid, bad != 0 || err != 0);
            const uint32_t S1 = (uint32_t)ex.w_lane(tid, (int32_t)ex.w_iscan(tid, s1l), 63);
            if (lane == 0) {
                uint32_t* w = fs.wt[wave];
                w[F::W_Q1] = r_q1; w[F::W_I2] = r_i2; w[F::W_I3] = r_i3; w[F::W_I4] = r_i4; w[F::W_I5] = r_i5;
                w[F::W_LV0] = LV0; w[F::W_LV1] = LV1; w[F::W_LM1] = LM1;
                w[F::W_LV2] = LV2; w[F::W_LV3] = LV3; w[F::W_LV4] = LV4; w[F::W_LM2] = LM2; w[F::W_LM3] = LM3; w[F::W_LM4] = LM4;
                w[F::W_S1] = S1; w[F::W_S2] = S2; w[F::W_S3] = S3; w[F::W_S4] = S4; w[F::W_S5] = S5;
                w[F::W_N] = 0;
                if (anybad) ex.lds_or_nr(&sh.f_flags, 1u);
            }
        }
        ex.stamp(15);  // fast: phase 1 (wave 0's view)
        ex.sync(tid);
        ex.stamp(16);  // fast: waiting for the other waves' phase 1

        // ================= top of the tree, sizes, heuristic: wave 0 =================
        if (wave == 0) {
            // lane l holds height-5 node l; heights 6, 7, 8 are its groups of 4, 16, 64 lanes
            const bool v5 = lane < N5;
            const int32_t* t = fs.t5[v5 ? lane : 0];
            const int32_t mn5 = t[0], mx5 = t[1], smn5 = t[2], smx5 = t[3], df5 = t[4];
            const bool eq5 = t[5] != 0;
            const int32_t mn6 = ex.template w_gmin<4>(tid, mn5), mx6 = ex.template w_gmax<4>(tid, mx5);
            const int32_t smn6 = ex.template w_gmin<4>(tid, smn5), smx6 = ex.template w_gmax<4>(tid, smx5);
            const int32_t mn7 = ex.template w_gmin<16>(tid, mn6), mx7 = ex.template w_gmax<16>(tid, mx6);
            const int32_t smn7 = ex.template w_gmin<16>(tid, smn6), smx7 = ex.template w_gmax<16>(tid, smx6);
            int32_t mn8 = ex.w_lane(tid, mn7, 0), mx8 = ex.w_lane(tid, mx7, 0), smn8 = ex.w_lane(tid, smn7, 0), smx8 = ex.w_lane(tid, smx7, 0);
            if (H == 8) {
#pragma unroll
                for (int r = 16; r < 64; r += 16) {
                    const int32_t a = ex.w_lane(tid, mn7, r), b = ex.w_lane(tid, mx7, r), c = ex.w_lane(tid, smn7, r), d = ex.w_lane(tid, smx7, r);
                    mn8 = a < mn8 ? a : mn8;
                    mx8 = b > mx8 ? b : mx8;
                    smn8 = c < smn8 ? c : smn8;
                    smx8 = d > smx8 ? d : smx8;
                }
            }
            const uint64_t bne = ex.w_ballot(tid, df5 != ex.w_prev(tid, df5));
            const uint64_t beq = ex.w_ballot(tid, eq5);
            const uint64_t ok4 = beq & ~(bne & 0xEEEEEEEEEEEEEEEEull);
            uint64_t y4 = ok4 & (ok4 >> 1);
            y4 &= y4 >> 2;
            const uint64_t ok16 = beq & ~(bne & 0xFFFEFFFEFFFEFFFEull);
            uint64_t y16 = ok16 & (ok16 >> 1);
            y16 &= y16 >> 2;
            y16 &= y16 >> 4;
            y16 &= y16 >> 8;
            const int32_t dfa = ex.w_lane(tid, df5, 0), dfb = ex.w_lane(tid, df5, 16), dfc = ex.w_lane(tid, df5, 32), dfd = ex.w_lane(tid, df5, 48);
            const bool eq8 = ok16 == ~0ull && dfa == dfb && dfa == dfc && dfa == dfd;
            const bool eq6 = (y4 >> (lane & ~3)) & 1ull, eq7 = (y16 >> (lane & ~15)) & 1ull;
            // per level h = 5..H: the node this lane represents (level 5: every lane; 6: quad leaders; 7: row leaders; 8: lane 0)
            const bool rep[4] = {v5, v5 && (lane & 3) == 0, v5 && (lane & 15) == 0, lane == 0};
            const int32_t tmn[4] = {mn5, mn6, mn7, mn8}, tmx[4] = {mx5, mx6, mx7, mx8}, tsn[4] = {smn5, smn6, smn7, smn8},
                          tsx[4] = {smx5, smx6, smx7, smx8};
            const bool teq[4] = {eq5, eq6, eq7, eq8};
            bool P[5], PS[5], unif[4];
            uint64_t bP[5], bS[5];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                unif[k] = tmn[k] == tmx[k];
                P[k] = 5 + k <= H && rep[k] && !unif[k] && !teq[k];
                PS[k] = 5 + k <= H && rep[k] && !unif[k];
                bP[k] = ex.w_ballot(tid, P[k]);
                bS[k] = ex.w_ballot(tid, PS[k]);
            }
            P[4] = false; PS[4] = false; bP[4] = 0; bS[4] = 0;
            // ---- waves' totals -> exclusive prefixes over the waves: lane = 16 g + w handles field 4 p + g of wave w in
            //      pass p, one row scan per pass; the prefixes go back to LDS for the waves to pick up ----
            uint32_t tot[20];
            {
                const int g = lane >> 4, w = lane & 15;
#pragma unroll
                for (int p = 0; p < 5; p++) {
                    const uint32_t x = w < NW ? fs.wt[w < NW ? w : 0][4 * p + g] : 0u;
                    const uint32_t inc = ex.w_rowscan(tid, x);
                    if (w < NW) fs.wex[w][4 * p + g] = inc - x;
#pragma unroll
                    for (int gg = 0; gg < 4; gg++) tot[4 * p + gg] = (uint32_t)ex.w_lane(tid, (int32_t)inc, 16 * gg + 15);
                }
            }
            // ---- level structure of both candidates ----
            Totals<C> TL;
            uint32_t NiL[H + 2];
            NiL[1] = tot[F::W_Q1]; NiL[2] = tot[F::W_I2]; NiL[3] = tot[F::W_I3]; NiL[4] = tot[F::W_I4]; NiL[5] = tot[F::W_I5];
            uint32_t sAll = tot[F::W_S1] + tot[F::W_S2] + tot[F::W_S3] + tot[F::W_S4] + tot[F::W_S5];  // snapshot: internal nodes
#pragma unroll
            for (int h = 6; h <= H; h++) {
                NiL[h] = popc64(bP[h - 5]);
                sAll += popc64(bS[h - 5]);
            }
            TL.from_counts(NiL);
            // snapshot candidate (snapshot.rs:126-146): every internal node has four visited children
            const uint32_t sN0 = 1 + 4 * sAll, sLT = 1 + 4 * (sAll - tot[F::W_S1]), sM0 = sAll;
            // ---- values of the top nodes; which of them are visited ----
            bool vis[4];
            uint32_t zx[4], zn[4], vidx[4], iidx[4], eidx[4];
            uint64_t bLV[4], bLM[4];
            bool widetop = false;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int h = 5 + k;
                if (h > H) {
                    vis[k] = false; zx[k] = zn[k] = vidx[k] = iidx[k] = eidx[k] = 0; bLV[k] = bLM[k] = 0;
                    continue;
                }
                // parent's flag: the group this lane belongs to one level up
                const uint32_t pl = k == 0 ? ((uint32_t)lane & ~3u) : (k == 1 ? ((uint32_t)lane & ~15u) : 0u);  // parent's representative lane
                const bool parentP = h == H ? true : ((bP[k + 1] >> pl) & 1ull) != 0;
                vis[k] = rep[k] && parentP;
                const int32_t vx = tmx[k] - tsx[k], vn = tmn[k] - tsn[k];
                widetop = widetop || (rep[k] && (vx < -32768 || vx > 32767 || vn < -32768 || vn > 32767));
                zx[k] = zz32(vx) & 0xffffu;
                zn[k] = zz32(vn) & 0xffffu;
                const uint32_t prank = h == H ? 0u : popc64(bP[k + 1] & (((uint64_t)1 << pl) - 1));  // internal nodes of height h+1 before the parent
                const uint32_t sib = k == 0 ? ((uint32_t)lane & 3u) : (k == 1 ? (((uint32_t)lane >> 2) & 3u) : (k == 2 ? (((uint32_t)lane >> 4) & 3u) : 0u));
                const uint32_t vrank = h == H ? 0u : 4 * prank + sib;
                const uint32_t irank = ex.w_mbcnt(tid, bP[k]);
                vidx[k] = TL.offV[h] + vrank;
                iidx[k] = TL.offI[h] + irank;
                eidx[k] = TL.offZ[h] + vrank - irank;
                bLV[k] = ex.w_ballot(tid, vis[k] && zx[k] > 0xffu);
                bLM[k] = ex.w_ballot(tid, P[k] && zn[k] > 0xffu);
            }
            const uint64_t anywide = ex.w_ballot(tid, widetop);
            uint32_t LtopV = 0, LtopM = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                LtopV += popc64(bLV[k]);
                LtopM += popc64(bLM[k]);
            }
            const uint32_t n1v = LtopV + tot[F::W_LV0] + tot[F::W_LV1] + tot[F::W_LV2] + tot[F::W_LV3] + tot[F::W_LV4];
            const uint32_t n1m = LtopM + tot[F::W_LM1] + tot[F::W_LM2] + tot[F::W_LM3] + tot[F::W_LM4];
            const uint32_t eo = 13 + bitmap_size(TL.LT);
            const DacLayout DV = dac_layout(eo + bitmap_size(TL.LT - TL.M0), TL.N0, n1v, 0, 0);
            const DacLayout DM = dac_layout(DV.end, TL.M0, n1m, 0, 0);
            const uint32_t log_size = DM.end;  // log.rs:95-97
            // snapshot.rs:87-92 with one byte per value: a lower bound of the Snapshot's size
            const uint32_t snap_lb = 13 + bitmap_size(sLT) + (1 + bitmap_size(sN0) + sN0) + (sM0 ? 1 + bitmap_size(sM0) + sM0 : 1u);
            const bool flags = ex.uni(sh.f_flags) != 0;
            const bool okk = !flags && anywide == 0 && !cap254 && snap_lb > log_size && (uint64_t)log_size <= room;
            if (lane == 0) {
                fs.ok = okk ? 1u : 0u;
                fs.isize = log_size;
                fs.lt = TL.LT; fs.m0 = TL.M0; fs.n0 = TL.N0; fs.n1v = n1v; fs.n1m = n1m; fs.eq_off = eo;
                fs.vbm0 = DV.bm_off[0]; fs.vby0 = DV.by_off[0]; fs.vbm1 = DV.bm_off[1]; fs.vby1 = DV.by_off[1]; fs.vnlev = DV.nlev;
                fs.mbm0 = DM.bm_off[0]; fs.mby0 = DM.by_off[0]; fs.mbm1 = DM.bm_off[1]; fs.mby1 = DM.by_off[1]; fs.mnlev = DM.nlev;
            }
            if (okk) {
                if (lane == 0) {  // what the waves need to place their pieces: level offsets, second bytes before each level
                    uint32_t lv = LtopV, lm = LtopM;
                    const uint32_t tLV[5] = {tot[F::W_LV0], tot[F::W_LV1], tot[F::W_LV2], tot[F::W_LV3], tot[F::W_LV4]};
                    const uint32_t tLM[5] = {0u, tot[F::W_LM1], tot[F::W_LM2], tot[F::W_LM3], tot[F::W_LM4]};
#pragma unroll
                    for (int h = 4; h >= 0; h--) {
                        fs.offV[h] = TL.offV[h];
                        fs.offI[h] = TL.offI[h];
                        fs.offZ[h] = TL.offZ[h];
                        fs.lvb[h] = lv;
                        fs.lmb[h] = lm;
                        lv += tLV[h];
                        lm += tLM[h];
                    }
                }
                // ---- header and the nodes of heights 5..H (level order: they come first in every stream) ----
                if (lane == 0) {
                    io[0] = 2;  // k (log.rs:54)
                    store_be32(io + 1, ta.rows);
                    store_be32(io + 5, ta.cols);
                    store_be32(io + 9, (uint32_t)C::S);
                    io[DV.bm_off[0] - 1] = (uint8_t)DV.nlev;  // dac.rs:38
                    io[DM.bm_off[0] - 1] = (uint8_t)DM.nlev;
                    if (ta.minmax) {
                        ta.minmax[2 * inst] = (int64_t)tmn[H - 5] + s_base;
                        ta.minmax[2 * inst + 1] = (int64_t)tmx[H - 5] + s_base;
                    }
                }
                uint32_t lvb = 0, lmb = 0;  // second bytes of the levels above
#pragma unroll
                for (int k = 3; k >= 0; k--) {
                    if (5 + k > H) continue;
                    if (vis[k]) {
                        gstore8(io + DV.by_off[0] + vidx[k], (uint8_t)zx[k]);
                        if (zx[k] > 0xffu) {
                            bm_set(ex, sh.bmV0, vidx[k]);
                            gstore8(io + DV.by_off[1] + lvb + ex.w_mbcnt(tid, bLV[k]), (uint8_t)(zx[k] >> 8));
                        }
                        if (P[k]) {
                            bm_set(ex, sh.bmT, vidx[k]);
                            gstore8(io + DM.by_off[0] + iidx[k], (uint8_t)zn[k]);
                            if (zn[k] > 0xffu) {
                                bm_set(ex, sh.bmM[0], iidx[k]);
                                gstore8(io + DM.by_off[1] + lmb + ex.w_mbcnt(tid, bLM[k]), (uint8_t)(zn[k] >> 8));
                            }
                        } else if (!unif[k]) {
                            bm_set(ex, sh.bmE, eidx[k]);
                        }
                    }
                    lvb += popc64(bLV[k]);
                    lmb += popc64(bLM[k]);
                }
            }
        }
        ex.stamp(17);  // fast: top of the tree + plan
        ex.sync(tid);
        const bool ok = ex.uni(fs.ok) != 0;
        if (!ok) return;

        // ================= copy-out: every wave moves its own pieces to their final places =================
        {
            uint8_t* const vp0 = io + ex.uni(fs.vby0);
            uint8_t* const vp1 = io + ex.uni(fs.vby1);
            uint8_t* const mp0 = io + ex.uni(fs.mby0);
            uint8_t* const mp1 = io + ex.uni(fs.mby1);
            // where this wave's values go: level offset + what the waves before it put there
            const uint32_t* wx = fs.wex[wave];
            const uint32_t eI[6] = {0u, ex.uni(wx[F::W_Q1]), ex.uni(wx[F::W_I2]), ex.uni(wx[F::W_I3]), ex.uni(wx[F::W_I4]), ex.uni(wx[F::W_I5])};
            const uint32_t eLV[5] = {ex.uni(wx[F::W_LV0]), ex.uni(wx[F::W_LV1]), ex.uni(wx[F::W_LV2]), ex.uni(wx[F::W_LV3]), ex.uni(wx[F::W_LV4])};
            const uint32_t eLM[5] = {0u, ex.uni(wx[F::W_LM1]), ex.uni(wx[F::W_LM2]), ex.uni(wx[F::W_LM3]), ex.uni(wx[F::W_LM4])};
            uint32_t dV[5], dM[5], dE[5], lV[5], lM[5];
#pragma unroll
            for (int h = 0; h < 5; h++) {
                dV[h] = ex.uni(fs.offV[h]) + 4 * eI[h + 1];
                lV[h] = ex.uni(fs.lvb[h]) + eLV[h];
                dM[h] = h ? ex.uni(fs.offI[h]) + eI[h] : 0u;
                dE[h] = h ? ex.uni(fs.offZ[h]) + 4 * eI[h + 1] - eI[h] : 0u;
                lM[h] = h ? ex.uni(fs.lmb[h]) + eLM[h] : 0u;
            }
            // (T bits share the index space of the Lmax values)
            // 64-bit lane mask (lane i = bit i) -> bits [P, P + 64) of an MSB-first LDS bit array
            auto or_mask = [&](uint32_t* bm, uint32_t P, uint64_t m) {
                const uint32_t hi = brev32((uint32_t)m), lo = brev32((uint32_t)(m >> 32));  // lanes 0..31 | 32..63, MSB first
                const uint32_t sft = P & 31u;
                uint32_t v = 0;
                if (lane == 0) v = hi >> sft;
                if (lane == 1) v = (sft ? hi << (32 - sft) : 0u) | (lo >> sft);
                if (lane == 2) v = sft ? lo << (32 - sft) : 0u;
                if (lane < 3 && v) ex.lds_or_nr(&bm[(P >> 5) + (uint32_t)lane], v);
            };
            // The wave's four pieces, one per iteration: a stream of the wave is the concatenation of its four parts, so every
            // step below works on full lanes (4 codes per lane for the Lmax streams, whose values come in groups of four
            // siblings; 1 per lane for the Lmin streams).
            uint32_t pb[4], pq1[4], pi2[4], pi3[4], pi4[4], pi5[4];
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const uint32_t* pc = fs.piece[wave * 4 + it];
                pb[it] = ex.uni(pc[0]); pq1[it] = ex.uni(pc[1]); pi2[it] = ex.uni(pc[2]); pi3[it] = ex.uni(pc[3]); pi4[it] = ex.uni(pc[4]); pi5[it] = ex.uni(pc[5]);
            }
            // part `it` of a stream: first code (u16 index into the pool), its flags (u32 index, groups of four), its length
            struct Parts {
                uint32_t src[4], fl[4], n[4];
            };
            auto parts = [&](int which) {  // 0..4: Lmax of level `which`; 5..8: Lmin of level which - 4
                Parts P;
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const uint32_t q1 = pq1[it], i2 = pi2[it], i3 = pi3[it], i4 = pi4[it], i5 = pi5[it];
                    const uint32_t oV1 = 4 * q1, oV2 = oV1 + 4 * i2, oV3 = oV2 + 4 * i3, oV4 = oV3 + 4 * i4, oF1 = oV4 + 4 * i5,
                                   oF2 = oF1 + 2 * i2, oF3 = oF2 + 2 * i3, oF4 = oF3 + 2 * i4, oM1 = oF4 + 2 * i5,
                                   oM2 = oM1 + q1, oM3 = oM2 + i2, oM4 = oM3 + i3;
                    const uint32_t o[9] = {0u, oV1, oV2, oV3, oV4, oM1, oM2, oM3, oM4};
                    const uint32_t f[9] = {0u, oF1, oF2, oF3, oF4, 0u, 0u, 0u, 0u};
                    const uint32_t n[9] = {4 * q1, 4 * i2, 4 * i3, 4 * i4, 4 * i5, q1, i2, i3, i4};
                    P.src[it] = pb[it] + o[which];
                    P.fl[it] = (pb[it] + f[which]) >> 1;
                    P.n[it] = n[which];
                }
                return P;
            };
            // Lmax stream of one level: four codes per lane and step -> one unaligned 4-byte store of the plane-0 bytes, the
            // continuation nibble, second bytes by rank (one DPP scan per step); with flags (levels >= 1) the T nibble and the
            // eqB run of the group (log.rs:137-152)
            auto stream4 = [&](const Parts& P, bool flags, uint32_t& d, uint32_t& l, uint32_t& e) {
                const uint32_t c1 = P.n[0], c2 = c1 + P.n[1], c3 = c2 + P.n[2], n = c3 + P.n[3];
                const uint32_t* const st32 = (const uint32_t*)st16;
                for (uint32_t i0 = 0; i0 < n; i0 += 256) {
                    const uint32_t k = i0 + 4u * (uint32_t)lane;
                    const bool valid = k < n;
                    // which part, and where in it
                    const uint32_t sidx = k < c2 ? (k < c1 ? P.src[0] + k : P.src[1] + (k - c1)) : (k < c3 ? P.src[2] + (k - c2) : P.src[3] + (k - c3));
                    const uint32_t a = valid ? st32[sidx >> 1] : 0u, b = valid ? st32[(sidx >> 1) + 1] : 0u;
                    if (valid) gstore32u(vp0 + d + k, (a & 0xffu) | ((a >> 8) & 0xff00u) | ((b & 0xffu) << 16) | ((b << 8) & 0xff000000u));
                    const uint32_t nib = ((a & 0xff00u) ? 8u : 0u) | ((a >> 24) ? 4u : 0u) | ((b & 0xff00u) ? 2u : 0u) | ((b >> 24) ? 1u : 0u);
                    const uint32_t c = popc32(nib);
                    uint32_t tn = 0, en = 0, ne = 0;  // T bits; eqB bits and their number
                    if (flags) {
                        const uint32_t fidx = k < c2 ? (k < c1 ? P.fl[0] + (k >> 2) : P.fl[1] + ((k - c1) >> 2))
                                                     : (k < c3 ? P.fl[2] + ((k - c2) >> 2) : P.fl[3] + ((k - c3) >> 2));
                        const uint32_t fw = valid ? st32[fidx] : 0x02020202u;
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const uint32_t f = (fw >> (8 * j)) & 3u;
                            tn = (tn << 1) | (f >> 1);
                            if (f != 2u) {
                                en = (en << 1) | (f & 1u);
                                ne++;
                            }
                        }
                    }
                    const uint32_t inc = ex.w_iscan(tid, c | (ne << 16));
                    if (nib) {
                        te_or_run(ex, sh.bmV0, d + k, 4, nib);
                        uint32_t q = l + ((inc & 0xffffu) - c);
                        if (nib & 8u) gstore8(vp1 + q++, (uint8_t)(a >> 8));
                        if (nib & 4u) gstore8(vp1 + q++, (uint8_t)(a >> 24));
                        if (nib & 2u) gstore8(vp1 + q++, (uint8_t)(b >> 8));
                        if (nib & 1u) gstore8(vp1 + q, (uint8_t)(b >> 24));
                    }
                    if (valid && tn) te_or_run(ex, sh.bmT, d + k, 4, tn);
                    if (en) te_or_run(ex, sh.bmE, e + ((inc >> 16) - ne), ne, en);
                    const uint32_t last = (uint32_t)ex.w_lane(tid, (int32_t)inc, 63);
                    l += last & 0xffffu;
                    e += last >> 16;
                }
            };
            // Lmin stream of one level: one code per lane and step; continuation bits by ballot, second bytes by rank
            auto stream1 = [&](const Parts& P, uint32_t d, uint32_t l) {
                const uint32_t c1 = P.n[0], c2 = c1 + P.n[1], c3 = c2 + P.n[2], n = c3 + P.n[3];
                for (uint32_t i0 = 0; i0 < n; i0 += 64) {
                    const uint32_t k = i0 + (uint32_t)lane;
                    const bool valid = k < n;
                    const uint32_t sidx = k < c2 ? (k < c1 ? P.src[0] + k : P.src[1] + (k - c1)) : (k < c3 ? P.src[2] + (k - c2) : P.src[3] + (k - c3));
                    const uint32_t z = valid ? st16[sidx] : 0u;
                    if (valid) gstore8(mp0 + d + k, (uint8_t)z);
                    const bool lng = z > 0xffu;
                    const uint64_t bl = ex.w_ballot(tid, lng);
                    if (lng) gstore8(mp1 + l + ex.w_mbcnt(tid, bl), (uint8_t)(z >> 8));
                    if (bl) or_mask(sh.bmM[0], d + i0, bl);
                    l += popc64(bl);
                }
            };
            uint32_t none = 0;
            stream4(parts(0), false, dV[0], lV[0], none);
#pragma unroll
            for (int h = 1; h <= 4; h++) stream4(parts(h), true, dV[h], lV[h], dE[h]);
#pragma unroll
            for (int h = 1; h <= 4; h++) stream1(parts(4 + h), dM[h], lM[h]);
        }
        ex.stamp(18);  // fast: copy-out
        ex.sync(tid);
    });

    const bool ok = ex.uni(fs.ok) != 0;
    const uint32_t isize = ex.uni(fs.isize);
    const uint32_t lt = ex.uni(fs.lt), m0 = ex.uni(fs.m0), n0 = ex.uni(fs.n0), n1v = ex.uni(fs.n1v), n1m = ex.uni(fs.n1m);
    const uint32_t eo = ex.uni(fs.eq_off), vbm0 = ex.uni(fs.vbm0), vbm1 = ex.uni(fs.vbm1), mbm0 = ex.uni(fs.mbm0), mbm1 = ex.uni(fs.mbm1);
    const uint32_t vnlev = ex.uni(fs.vnlev), mnlev = ex.uni(fs.mnlev);
    ex.par([&](int tid, EncRegs&) {  // (barrier: the plan has been read by everybody before the pool is reused)
        if (tid == 0) {
            sh.f_top = 0;
            sh.f_flags = 0;
        }
    });
    if (!ok) return false;
    // T, eqB and both continuation bitmaps serialized with their rank indexes (bitmap.rs:66-112,128-138)
    const BmJob jobs[4] = {{sh.bmT, lt, nullptr, 0, io + 13},
                           {sh.bmE, lt - m0, nullptr, 0, io + eo},
                           {sh.bmV0, n0, nullptr, 0, vnlev > 0 ? io + vbm0 : nullptr},
                           {sh.bmM[0], m0, nullptr, 0, mnlev > 0 ? io + mbm0 : nullptr}};
    bitmaps_finish<C, 4>(ex, jobs);
    if (vnlev > 1) bitmap_write_zero<C>(ex, n1v, io + vbm1);  // the last level of a Dac never continues
    if (mnlev > 1) bitmap_write_zero<C>(ex, n1m, io + mbm1);
    ex.stamp(19);  // fast: bitmap serialization
    isize_out = isize;
    return true;
}

}  // namespace k2r
