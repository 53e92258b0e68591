// k2r_decode.h -- random access into encoded chunks: rank / DAC get / tree descent.
//
// Device-side restatement of the reference's query path over the SERIALIZED bytes (no pointer-rich
// in-memory structures): BitMap::get/rank (bitmap.rs:176-217), Dac::get (dac.rs:80-93),
// Snapshot::get/_search_window (snapshot.rs:165-188, 347-421), Log::get/_search_window
// (log.rs:176-293, 553-702).  The host parses the layout once (chunk.rs:247-266 et al.) into
// InstDesc records holding byte offsets; kernels then index the raw big-endian stream directly.
// Search reproduces the reference's pruning decisions exactly (including snapshot.rs:392 and the
// root handling of single-node logs, log.rs:527-548) so the result SET is the reference's.
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

// ---- search (faithful DFS with the reference's pruning) ---------------------------------------------
// Results are marked in a bitmap over the window (bit = (row-top)*wcols + (col-left)), owned by one thread.
struct WinMark {
    uint32_t* bits;
    uint32_t top, left, wcols;
    K2R_HD void rect(uint32_t r0, uint32_t r1, uint32_t c0, uint32_t c1) const {  // inclusive bounds
        for (uint32_t r = r0; r <= r1; r++) {
            const uint32_t base = (r - top) * wcols - left;
            for (uint32_t c = c0; c <= c1; c++) {
                const uint32_t p = base + c;
                bits[p >> 5] |= 1u << (p & 31);
            }
        }
    }
};

constexpr int MAX_DEPTH = 20;

struct SFrame {  // one activation of _search_window
    uint32_t sl, base_t, base_s;  // child side length, first-child index in log / snapshot tree
    uint32_t top, bottom, left, right, toff, loff;
    uint32_t i, j, i_end, j0, j_end;
    int64_t min_t, min_s, max_t, max_s;
    uint8_t has_t, has_s;
};

// Snapshot::search_window, snapshot.rs:310-421
K2R_HD void snapshot_search(const uint8_t* b, const InstDesc& S, uint32_t top, uint32_t bottom, uint32_t left,
                            uint32_t right, int64_t lower, int64_t upper, const WinMark& wm) {
    // bounds are half-open on entry (geom::Rect); the recursion uses inclusive ones (snapshot.rs:329-332)
    if (!bmd_get(b, S.T, 0)) {
        const int64_t v = dacd_get(b, S.mx, 0);
        if (lower <= v && v <= upper) wm.rect(top, bottom - 1, left, right - 1);
        return;
    }
    const uint32_t k = S.k;
    SFrame st[MAX_DEPTH];
    int sp = 0;
    auto enter = [&](uint32_t sidelen, uint32_t t, uint32_t bo, uint32_t l, uint32_t r, uint32_t index, int64_t mn,
                     int64_t mx, uint32_t toff, uint32_t loff) {
        if (sidelen / k == 0) return;  // (malformed input only: dcdf_chunk_open rejects a sidelen that is no power of k)
        SFrame& f = st[sp++];
        f.sl = sidelen / k;
        f.base_s = 1 + bmd_rank(b, S.T, index) * k * k;
        f.top = t; f.bottom = bo; f.left = l; f.right = r; f.toff = toff; f.loff = loff;
        f.i = t / f.sl; f.i_end = bo / f.sl; f.j0 = l / f.sl; f.j_end = r / f.sl; f.j = f.j0;
        f.min_s = mn; f.max_s = mx;
    };
    enter(S.sidelen, top, bottom - 1, left, right - 1, 0, dacd_get(b, S.mn, 0), dacd_get(b, S.mx, 0), 0, 0);
    while (sp > 0) {
        SFrame& f = st[sp - 1];
        if (f.i > f.i_end) {
            sp--;
            continue;
        }
        const uint32_t i = f.i, j = f.j;
        if (++f.j > f.j_end) {
            f.j = f.j0;
            f.i++;
        }
        const uint32_t sl = f.sl;
        const uint32_t top_ = f.top > i * sl ? f.top - i * sl : 0;
        const uint32_t bottom_ = (f.bottom - i * sl) < (sl - 1) ? (f.bottom - i * sl) : (sl - 1);
        const uint32_t toff_ = f.toff + i * sl;
        const uint32_t left_ = f.left > j * sl ? f.left - j * sl : 0;
        const uint32_t right_ = (f.right - j * sl) < (sl - 1) ? (f.right - j * sl) : (sl - 1);
        const uint32_t loff_ = f.loff + j * sl;
        const uint32_t index_ = f.base_s + i * k + j;
        const int64_t max_value_ = f.max_s - dacd_get(b, S.mx, index_);
        if (index_ >= S.T.len || !bmd_get(b, S.T, index_)) {
            if (lower <= max_value_ && max_value_ <= upper) wm.rect(toff_ + top_, toff_ + bottom_, loff_ + left_, loff_ + right_);
        } else {
            const int64_t min_value_ = f.min_s + dacd_get(b, S.mn, bmd_rank(b, S.T, index_));
            if (lower <= f.min_s && max_value_ <= upper) {  // sic: the PARENT's min (snapshot.rs:392)
                wm.rect(toff_ + top_, toff_ + bottom_, loff_ + left_, loff_ + right_);
            } else if (upper >= min_value_ && lower <= max_value_) {
                if (sp < MAX_DEPTH && sl > 1) enter(sl, top_, bottom_, left_, right_, index_, min_value_, max_value_, toff_, loff_);
            }
        }
    }
}

// Log::search_window, log.rs:519-702
K2R_HD void log_search(const uint8_t* b, const InstDesc& S, const InstDesc& L, uint32_t top, uint32_t bottom,
                       uint32_t left, uint32_t right, int64_t lower, int64_t upper, const WinMark& wm) {
    const uint32_t k = L.k;
    const bool single_t = !bmd_get(b, L.T, 0);
    const bool single_s = !bmd_get(b, S.T, 0);
    SFrame st[MAX_DEPTH];
    int sp = 0;
    // returns true when the activation recursed (a frame was pushed)
    auto call = [&](uint32_t sidelen, uint32_t t, uint32_t bo, uint32_t l, uint32_t r, bool has_t, uint32_t index_t,
                    bool has_s, uint32_t index_s, int64_t min_t, int64_t min_s, int64_t max_t, int64_t max_s,
                    uint32_t toff, uint32_t loff) {
        const int64_t max_value = max_s + max_t, min_value = min_s + min_t;  // log.rs:573-574
        if (min_value >= lower && max_value <= upper) {
            wm.rect(toff + t, toff + bo, loff + l, loff + r);
            return;
        }
        if (min_value > upper || max_value < lower) return;
        const uint32_t sl = sidelen / k;
        if (sl == 0 || sp >= MAX_DEPTH) return;
        SFrame& f = st[sp++];
        f.sl = sl;
        f.has_t = has_t; f.has_s = has_s;
        f.base_t = has_t ? 1 + bmd_rank(b, L.T, index_t) * k * k : 0;
        f.base_s = has_s ? 1 + bmd_rank(b, S.T, index_s) * k * k : 0;
        f.top = t; f.bottom = bo; f.left = l; f.right = r; f.toff = toff; f.loff = loff;
        f.i = t / sl; f.i_end = bo / sl; f.j0 = l / sl; f.j_end = r / sl; f.j = f.j0;
        f.min_t = min_t; f.min_s = min_s; f.max_t = max_t; f.max_s = max_s;
    };
    call(L.sidelen, top, bottom - 1, left, right - 1, !single_t, 0, !single_s, 0, dacd_get(b, L.mn, 0),
         dacd_get(b, S.mn, 0), dacd_get(b, L.mx, 0), dacd_get(b, S.mx, 0), 0, 0);
    while (sp > 0) {
        SFrame& f = st[sp - 1];
        if (f.i > f.i_end) {
            sp--;
            continue;
        }
        const uint32_t i = f.i, j = f.j;
        if (++f.j > f.j_end) {
            f.j = f.j0;
            f.i++;
        }
        const uint32_t sl = f.sl;
        const uint32_t top_ = f.top > i * sl ? f.top - i * sl : 0;
        const uint32_t bottom_ = (f.bottom - i * sl) < (sl - 1) ? (f.bottom - i * sl) : (sl - 1);
        const uint32_t toff_ = f.toff + i * sl;
        const uint32_t left_ = f.left > j * sl ? f.left - j * sl : 0;
        const uint32_t right_ = (f.right - j * sl) < (sl - 1) ? (f.right - j * sl) : (sl - 1);
        const uint32_t loff_ = f.loff + j * sl;
        bool has_t = f.has_t, has_s = f.has_s;
        const uint32_t it = f.base_t + i * k + j, is = f.base_s + i * k + j;
        const int64_t max_t_ = has_t ? dacd_get(b, L.mx, it) : f.max_t;                 // log.rs:622-625
        const int64_t max_s_ = has_s ? f.max_s - dacd_get(b, S.mx, is) : f.max_s;       // log.rs:627-630
        const bool leaf_t = has_t ? (it >= L.T.len || !bmd_get(b, L.T, it)) : true;     // log.rs:632-635
        const bool leaf_s = has_s ? (is >= S.T.len || !bmd_get(b, S.T, is)) : true;
        int64_t min_t_ = has_t ? (leaf_t ? f.min_t : dacd_get(b, L.mn, bmd_rank(b, L.T, it))) : f.min_t;
        int64_t min_s_ = has_s ? (leaf_s ? f.min_s : f.min_s + dacd_get(b, S.mn, bmd_rank(b, S.T, is))) : f.min_s;
        if (leaf_s) {
            min_s_ = max_s_;
            has_s = false;
        }
        if (leaf_t) {
            min_t_ = max_t_;
            if (has_t) {
                if (it < L.T.len && !bmd_get(b, L.E, bmd_rank0(b, L.T, it + 1) - 1)) min_t_ = max_s_ + max_t_ - min_s_;
            }
            has_t = false;
        }
        // copy what the callee needs before `call` may push a frame that aliases nothing of f
        call(sl, top_, bottom_, left_, right_, has_t, it, has_s, is, min_t_, min_s_, max_t_, max_s_, toff_, loff_);
    }
}

K2R_HD void inst_search(const uint8_t* b, const InstDesc* descs, uint32_t i, uint32_t top, uint32_t bottom,
                        uint32_t left, uint32_t right, int64_t lower, int64_t upper, const WinMark& wm) {
    const InstDesc& D = descs[i];
    if (!D.is_log) snapshot_search(b, D, top, bottom, left, right, lower, upper, wm);  // block.rs:70-81
    else log_search(b, descs[D.snap], D, top, bottom, left, right, lower, upper, wm);
}

}  // namespace k2r
