// k2r_decode.h -- random access into encoded chunks: rank / DAC get / tree descent.
//
// Device-side restatement of the reference's query path over the SERIALIZED bytes (no pointer-rich
// in-memory structures): BitMap::get/rank (bitmap.rs:176-217), Dac::get (dac.rs:80-93),
// Snapshot::get/_search_window (snapshot.rs:165-188, 347-421), Log::get/_search_window
// (log.rs:176-293, 553-702).  The host parses the layout once (chunk.rs:247-266 et al.) into
// InstDesc records holding byte offsets; kernels then index the raw big-endian stream directly.
// Search is not in this header: its result SET is the reference's because decoding is exact and the one shape where the
// reference's own search is not "the cells in range" (log.rs:527-548) is carried as data (k2r_query.hip, SearchExtra).
#pragma once
#include "k2r_common.h"

namespace k2r {

struct BmDesc {  // bitmap.rs:128-138 as laid out in the stream
    uint32_t len;        // bits
    uint32_t k;          // index stride in words (4)
    uint32_t idx_off;    // byte offset of index[0]
    uint32_t words_off;  // byte offset of words[0]
};
struct DacDesc {  // dac.rs:37-44
    uint32_t nlev;
    BmDesc bm[8];
    uint32_t bytes_off[8];
};
struct InstDesc {  // one Snapshot (snapshot.rs:48-58) or Log (log.rs:53-64)
    uint32_t is_log;
    uint32_t snap;  // index of the owning block's snapshot InstDesc
    uint32_t k, rows, cols, sidelen;
    BmDesc T, E;
    DacDesc mx, mn;
};

K2R_HD uint32_t bmd_words(const BmDesc& d) { return (d.len + 31) / 32; }
K2R_HD bool bmd_get(const uint8_t* b, const BmDesc& d, uint32_t i) {  // bitmap.rs:176-183
    const uint32_t w = i >> 5;
    if (w >= bmd_words(d)) return false;  // the reference would panic; malformed input must not fault the GPU
    return (load_be32(b + d.words_off + 4 * w) >> (31 - (i & 31))) & 1u;
}
K2R_HD uint32_t bmd_rank(const uint8_t* b, const BmDesc& d, uint32_t i) {  // bitmap.rs:186-212
    if (i > d.len) i = d.len;
    const uint32_t block = (i >> 5) / d.k;
    uint32_t count = block > 0 ? load_be32(b + d.idx_off + 4 * (block - 1)) : 0;
    const uint32_t end = i >> 5;
    for (uint32_t w = block * d.k; w < end; w++) count += popc32(load_be32(b + d.words_off + 4 * w));
    const uint32_t left = i & 31;
    if (left) count += popc32(load_be32(b + d.words_off + 4 * end) >> (32 - left));
    return count;
}
K2R_HD uint32_t bmd_rank0(const uint8_t* b, const BmDesc& d, uint32_t i) { return i - bmd_rank(b, d, i); }

K2R_HD int64_t dacd_get(const uint8_t* b, const DacDesc& d, uint32_t index) {  // dac.rs:80-93
    uint64_t n = 0;
    for (uint32_t l = 0; l < d.nlev; l++) {
        if (index >= d.bm[l].len) break;  // malformed input guard
        n |= (uint64_t)b[d.bytes_off[l] + index] << (8 * l);
        if (bmd_get(b, d.bm[l], index)) index = bmd_rank(b, d.bm[l], index);
        else break;
    }
    return (int64_t)((n >> 1) ^ (0 - (n & 1)));  // zigzag_decode, dac.rs:139-142
}

// ---- point access --------------------------------------------------------------------------------
K2R_HD int64_t snapshot_get(const uint8_t* b, const InstDesc& S, uint32_t row, uint32_t col) {  // snapshot.rs:165-188
    int64_t max_value = dacd_get(b, S.mx, 0);
    if (!bmd_get(b, S.T, 0)) return max_value;
    const uint32_t k = S.k;
    uint32_t sl = S.sidelen, index = 0;
    for (int guard = 0; guard < 40 && sl > 1; guard++) {
        sl /= k;
        if (sl == 0) break;
        index = 1 + bmd_rank(b, S.T, index) * k * k + (row / sl) * k + col / sl;
        max_value -= dacd_get(b, S.mx, index);
        if (index >= S.T.len || !bmd_get(b, S.T, index)) return max_value;
        row %= sl;
        col %= sl;
    }
    return max_value;
}

K2R_HD int64_t log_get(const uint8_t* b, const InstDesc& S, const InstDesc& L, uint32_t row, uint32_t col) {
    // log.rs:176-293
    int64_t max_t = dacd_get(b, L.mx, 0);
    int64_t max_s = dacd_get(b, S.mx, 0);
    const bool single_t = !bmd_get(b, L.T, 0);
    const bool single_s = !bmd_get(b, S.T, 0);
    if (single_t && single_s) return max_t + max_s;
    if (single_t && !bmd_get(b, L.E, 0)) return max_t + max_s;
    bool has_t = !single_t, has_s = !single_s;
    uint32_t it = 0, is = 0;
    const uint32_t k = L.k;
    uint32_t sl = L.sidelen;
    for (int guard = 0; guard < 40 && sl > 1; guard++) {
        sl /= k;
        if (sl == 0) break;
        if (has_s) {
            is = 1 + bmd_rank(b, S.T, is) * k * k + (row / sl) * k + col / sl;
            max_s -= dacd_get(b, S.mx, is);
        }
        if (has_t) {
            it = 1 + bmd_rank(b, L.T, it) * k * k + (row / sl) * k + col / sl;
            max_t = dacd_get(b, L.mx, it);
        }
        // log.rs:240,245 write `>`; `>=` is identical wherever the reference does not panic
        const bool leaf_t = has_t ? (it >= L.T.len || !bmd_get(b, L.T, it)) : true;
        const bool leaf_s = has_s ? (is >= S.T.len || !bmd_get(b, S.T, is)) : true;
        if (leaf_t && leaf_s) return max_t + max_s;
        if (leaf_s) {
            has_s = false;
        } else if (leaf_t) {
            if (has_t && it < L.T.len) {
                const bool eq = bmd_get(b, L.E, bmd_rank0(b, L.T, it + 1) - 1);
                if (!eq) return max_t + max_s;
            }
            has_t = false;
        }
        row %= sl;
        col %= sl;
    }
    return max_t + max_s;
}

// value of (instant-desc i) at (row,col)
K2R_HD int64_t inst_get(const uint8_t* b, const InstDesc* descs, uint32_t i, uint32_t row, uint32_t col) {
    const InstDesc& D = descs[i];
    if (!D.is_log) return snapshot_get(b, D, row, col);  // block.rs:42-47
    return log_get(b, descs[D.snap], D, row, col);
}

// MMBuffer3::set conversions (mmbuffer.rs:505,525,560,622; fixed.rs:81-86)
K2R_HD void store_typed(void* out, int64_t off, int32_t dtype, int64_t v, uint32_t fbits) {
    switch (dtype) {
        case ENC_I32: ((int32_t*)out)[off] = (int32_t)v; break;
        case ENC_I64: ((int64_t*)out)[off] = v; break;
        case ENC_F32: {
            float f;
            if (v == 0) f = __builtin_nanf("");
            else f = (float)(v - 1) / (float)((int64_t)1 << (fbits + 1));
            ((float*)out)[off] = f;
            break;
        }
        default: {
            double f;
            if (v == 0) f = __builtin_nan("");
            else f = (double)(v - 1) / (double)((int64_t)1 << (fbits + 1));
            ((double*)out)[off] = f;
            break;
        }
    }
}

}  // namespace k2r
