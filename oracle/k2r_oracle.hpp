// k2r_oracle.hpp -- TEST INFRASTRUCTURE ONLY. NOT PART OF THE PRODUCT PATH.
//
// CPU restatement (C++17, host only) of the Heuristic K^2-Raster chunk path of
// Arbol-Project/dcdf v0.2.0.  Only tests/, __graft_entry__.smoke() and the
// `cpu_baseline` leg of bench.py may link or call this code, and only as the
// checker.  The shipped library (dcdf_amd/csrc -> libdcdf_k2r.so) never
// includes or links it.
//
// Parity status: PINNED by the reference's own known-answer tests (see
// tests/golden/reference_vectors.json and tests/test_oracle_golden.py):
// snapshot.rs:538-572, log.rs:901-955, bitmap.rs:261-284,336-346,
// dac.rs:163-179, fixed.rs:208-258,311-401.  The reference publishes no
// whole-chunk golden byte strings; whole-chunk bytes are pinned transitively
// (value streams + layout code + size()==len(bytes) identities).  The Rust
// crate itself cannot be built here (no cargo/rustc), so this is a "port".
//
// Structure deliberately follows the reference (recursive per-quad tree with a
// child vector per branch, VecDeque BFS, bit-at-a-time bitmap push, byte-plane
// DAC, two full tree builds per instant) because it doubles as the timed CPU
// baseline (`cpu_baseline.kind = "port"`).  Every function cites the reference
// file:line (relative to /root/reference/dcdf/src/) that it follows.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <limits>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

namespace orc {

// Reference panics become exceptions; the C ABI maps them to error codes.
struct Panic : std::runtime_error {
    int code;
    Panic(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
enum : int {
    ERR_OK = 0,
    ERR_BAD_ARG = -1,
    ERR_NONFINITE = -2,      // fixed.rs:39-41
    ERR_PRECISION = -3,      // fixed.rs:47-59
    ERR_OVERFLOW = -4,       // fixed.rs:65-70
    ERR_BOUNDS = -5,         // bitmap.rs:187-190 and slice indexing panics
    ERR_TOO_MANY_LOGS = -6,  // block.rs:27-32
    ERR_FORMAT = -7,         // short / malformed stream on read
};

// ---------------------------------------------------------------- byte sink / source
// extio.rs:196-233 (big-endian writers), extio.rs:55-111 (readers)
struct Writer {
    std::vector<uint8_t> buf;
    void write_byte(uint8_t b) { buf.push_back(b); }
    void write_u32(uint32_t w) {
        buf.push_back(uint8_t(w >> 24));
        buf.push_back(uint8_t(w >> 16));
        buf.push_back(uint8_t(w >> 8));
        buf.push_back(uint8_t(w));
    }
    void write_all(const std::vector<uint8_t>& b) { buf.insert(buf.end(), b.begin(), b.end()); }
};
struct Reader {
    const uint8_t* p;
    size_t n, pos = 0;
    Reader(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
    void need(size_t k) {
        if (pos + k > n) throw Panic(ERR_FORMAT, "unexpected end of stream");
    }
    uint8_t read_byte() {
        need(1);
        return p[pos++];
    }
    uint32_t read_u32() {
        need(4);
        uint32_t w = (uint32_t(p[pos]) << 24) | (uint32_t(p[pos + 1]) << 16) |
                     (uint32_t(p[pos + 2]) << 8) | uint32_t(p[pos + 3]);
        pos += 4;
        return w;
    }
    std::vector<uint8_t> read_bytes(size_t k) {
        need(k);
        std::vector<uint8_t> v(p + pos, p + pos + k);
        pos += k;
        return v;
    }
};

inline size_t div_ceil(size_t m, size_t n) { return m / n + (m % n > 0 ? 1 : 0); }  // bitmap.rs:221-231

// ---------------------------------------------------------------- BitMap
struct BitMap {  // bitmap.rs:117-122
    size_t length = 0;
    size_t k = 4;
    std::vector<uint32_t> index;
    std::vector<uint32_t> bitmap;

    bool get(size_t i) const {  // bitmap.rs:176-183
        size_t word_index = i / 32;
        size_t bit_index = i % 32;
        size_t shift = 31 - bit_index;
        if (word_index >= bitmap.size()) throw Panic(ERR_BOUNDS, "bitmap get out of bounds");
        uint32_t word = bitmap[word_index];
        return ((word >> shift) & 1) > 0;
    }
    size_t rank(size_t i) const {  // bitmap.rs:186-212
        if (i > length) throw Panic(ERR_BOUNDS, "index out of bounds (rank)");
        size_t block = i / 32 / k;
        uint32_t count = block > 0 ? index.at(block - 1) : 0;
        size_t start = block * k;
        size_t end = i / 32;
        for (size_t w = start; w < end; w++) count += (uint32_t)__builtin_popcount(bitmap.at(w));
        size_t leftover_bits = i - end * 32;
        if (leftover_bits > 0) {
            uint32_t word = bitmap.at(end);
            size_t shift = 32 - leftover_bits;
            count += (uint32_t)__builtin_popcount(word >> shift);
        }
        return count;
    }
    size_t rank0(size_t i) const { return i - rank(i); }  // bitmap.rs:215-217

    uint64_t size() const { return 4 + 4 + index.size() * 4 + bitmap.size() * 4; }  // bitmap.rs:169-171

    void write_to(Writer& w) const {  // bitmap.rs:128-138
        w.write_u32((uint32_t)length);
        w.write_u32((uint32_t)k);
        for (uint32_t b : index) w.write_u32(b);
        for (uint32_t b : bitmap) w.write_u32(b);
    }
    static BitMap read_from(Reader& r) {  // bitmap.rs:142-164
        BitMap bm;
        bm.length = r.read_u32();
        bm.k = r.read_u32();
        if (bm.k == 0) throw Panic(ERR_FORMAT, "bitmap k == 0");
        size_t blocks = bm.length / 32 / bm.k;
        for (size_t i = 0; i < blocks; i++) bm.index.push_back(r.read_u32());
        size_t words = div_ceil(bm.length, 32);
        for (size_t i = 0; i < words; i++) bm.bitmap.push_back(r.read_u32());
        return bm;
    }
};

struct BitMapBuilder {  // bitmap.rs:29-32
    size_t length = 0;
    std::vector<uint8_t> bitmap;

    void push(bool bit) {  // bitmap.rs:44-62
        size_t position = length % 8;
        size_t shift = 7 - position;
        if (position == 0) {
            bitmap.push_back(bit ? uint8_t(1u << shift) : 0);
        } else if (bit) {
            bitmap.back() = uint8_t(bitmap.back() + (1u << shift));
        }
        length += 1;
    }
    BitMap finish() const {  // bitmap.rs:66-112
        size_t k = 4;
        size_t blocks = length / 32 / k;
        std::vector<uint32_t> index;
        index.reserve(blocks);
        size_t words = div_ceil(length, 32);
        std::vector<uint32_t> bitmap32;
        bitmap32.reserve(words);
        if (words > 0) {
            int shift = 24;
            size_t word_index = 0;
            for (uint8_t byte : bitmap) {
                if (shift == 24) bitmap32.push_back(0);
                uint32_t word = byte;
                word <<= shift;
                bitmap32[word_index] |= word;
                if (shift == 0) {
                    word_index += 1;
                    shift = 24;
                } else {
                    shift -= 8;
                }
            }
        }
        uint32_t count = 0;
        for (size_t i = 0; i < blocks; i++) {
            for (size_t j = 0; j < k; j++) count += (uint32_t)__builtin_popcount(bitmap32.at(i * k + j));
            index.push_back(count);
        }
        BitMap bm;
        bm.length = length;
        bm.k = k;
        bm.index = std::move(index);
        bm.bitmap = std::move(bitmap32);
        return bm;
    }
    size_t naive_rank(size_t i) const {  // bitmap.rs:246-258 (test helper)
        size_t count = 0;
        for (size_t b = 0; b < i / 8; b++) count += (size_t)__builtin_popcount(bitmap[b]);
        size_t leftover = i % 8;
        if (leftover > 0) count += (size_t)__builtin_popcount((unsigned)(bitmap[i / 8] >> (8 - leftover)));
        return count;
    }
};

// ---------------------------------------------------------------- Dac
inline uint64_t zigzag_encode(int64_t n) {  // dac.rs:134-137
    return (uint64_t)(n >> 63) ^ ((uint64_t)n << 1);
}
inline int64_t zigzag_decode(uint64_t zz) {  // dac.rs:139-142
    return (int64_t)((zz >> 1) ^ ((zz & 1) == 1 ? 0xffffffffffffffffULL : 0ULL));
}

struct Dac {  // dac.rs:29-31
    std::vector<std::pair<BitMap, std::vector<uint8_t>>> levels;

    static Dac from(const std::vector<int64_t>& data) {  // dac.rs:101-131
        std::vector<std::pair<BitMapBuilder, std::vector<uint8_t>>> lv(8);
        for (int64_t d : data) {
            uint64_t datum = zigzag_encode(d);
            for (auto& [bitmap, bytes] : lv) {
                bytes.push_back(uint8_t(datum & 0xff));
                datum >>= 8;
                if (datum == 0) {
                    bitmap.push(false);
                    break;
                } else {
                    bitmap.push(true);
                }
            }
        }
        Dac dac;
        for (auto& [bitmap, bytes] : lv) {
            if (!(bitmap.length > 0)) break;  // take_while, dac.rs:126
            dac.levels.emplace_back(bitmap.finish(), std::move(bytes));
        }
        return dac;
    }
    int64_t get(size_t index) const {  // dac.rs:80-93
        uint64_t n = 0;
        for (size_t i = 0; i < levels.size(); i++) {
            const auto& [bitmap, bytes] = levels[i];
            if (index >= bytes.size()) throw Panic(ERR_BOUNDS, "dac get out of bounds");
            n |= (uint64_t)bytes[index] << (i * 8);
            if (bitmap.get(index)) {
                index = bitmap.rank(index);
            } else {
                break;
            }
        }
        return zigzag_decode(n);
    }
    size_t len() const { return levels.empty() ? 0 : levels[0].first.length; }  // dac.rs:153-155
    std::vector<int64_t> collect() const {                                       // dac.rs:158-160
        std::vector<int64_t> v;
        for (size_t i = 0; i < len(); i++) v.push_back(get(i));
        return v;
    }
    uint64_t size() const {  // dac.rs:68-74
        uint64_t s = 1;
        for (const auto& [bitmap, bytes] : levels) s += bitmap.size() + bytes.size();
        return s;
    }
    void write_to(Writer& w) const {  // dac.rs:37-44
        w.write_byte((uint8_t)levels.size());
        for (const auto& [bitmap, bytes] : levels) {
            bitmap.write_to(w);
            w.write_all(bytes);
        }
    }
    static Dac read_from(Reader& r) {  // dac.rs:48-63
        Dac d;
        size_t n_levels = r.read_byte();
        for (size_t i = 0; i < n_levels; i++) {
            BitMap bm = BitMap::read_from(r);
            std::vector<uint8_t> bytes = r.read_bytes(bm.length);
            d.levels.emplace_back(std::move(bm), std::move(bytes));
        }
        return d;
    }
};

// ---------------------------------------------------------------- fixed point
enum Encoding : int { ENC_I32 = 4, ENC_I64 = 8, ENC_F32 = 32, ENC_F64 = 64 };  // mmstruct.rs:36-43

// fixed.rs:31-71.  Arithmetic is carried out in the input float type F.
template <class F>
inline int64_t to_fixed(F n, size_t fractional_bits, bool round) {
    if (std::isnan(n)) return 0;
    if (!std::isfinite(n)) throw Panic(ERR_NONFINITE, "cannot convert non-finite to fixed");
    // F::from(1_i64 << fractional_bits): i64 -> F conversion (exact for powers of two)
    F shifted = n * (F)((int64_t)1 << fractional_bits);
    // fract() = self - self.trunc(); `> 0` is false for negatives (reference quirk)
    F fract = shifted - std::trunc(shifted);
    if (fract > (F)0) {
        if (round) {
            shifted = std::round(shifted);  // half away from zero, like Rust's round()
        } else {
            throw Panic(ERR_PRECISION, "loss of precision converting to fixed");
        }
    }
    shifted = shifted * (F)2;
    // to_i64(): truncating; None when out of range (num-traits checks against i64 bounds)
    // (num-traits 0.2 float->i64: Some iff  MIN as F <= x < MAX as F (== 2^63))
    if (!(shifted >= (F)-9223372036854775808.0 && shifted < (F)9223372036854775808.0))
        throw Panic(ERR_OVERFLOW, "overflow converting to fixed");
    int64_t number = (int64_t)shifted;
    return number + 1;
}
// fixed.rs:81-86
template <class F>
inline F from_fixed(int64_t n, size_t fractional_bits) {
    if (n == 0) return std::numeric_limits<F>::quiet_NaN();
    return (F)(n - 1) / (F)((int64_t)1 << (fractional_bits + 1));
}

struct Fraction {
    bool round;   // false = Precise, true = Round  (fixed.rs:88-92)
    size_t bits;
};
// fixed.rs:96-159; `data` is iterated in logical (row-major) order
template <class F, class Iter>
inline Fraction suggest_fraction(Iter begin, Iter end) {
    const size_t TOTAL_BITS = 62;
    if (begin == end) throw Panic(ERR_BAD_ARG, "suggest_fraction on empty data");
    Iter it = begin;
    F max_value = *it;
    ++it;
    while (std::isnan(max_value)) {
        if (it != end) {
            max_value = *it;
            ++it;
        } else {
            break;
        }
    }
    for (; it != end; ++it) {
        F n = *it;
        if (!std::isnan(n) && n > max_value) max_value = n;
    }
    if (std::isnan(max_value)) return Fraction{false, 0};
    // 1 + log2(max).floor() as usize  (saturating float->usize cast: negative -> 0, -inf -> 0)
    double lg = std::floor(std::log2((double)max_value));
    size_t whole_bits = 1 + (lg > 0 ? (size_t)lg : 0);
    if (std::isnan(lg)) whole_bits = 1;  // log2 of a negative max: NaN as usize == 0
    if (whole_bits > TOTAL_BITS) throw Panic(ERR_OVERFLOW, "value too large for fixed point");
    size_t max_fraction_bits = TOTAL_BITS - whole_bits;
    size_t fraction_bits = 0;
    for (it = begin; it != end; ++it) {
        double n = (double)*it;
        if (std::isnan(n)) continue;
        double shifted = n * (double)((int64_t)1 << max_fraction_bits);
        if (shifted - std::trunc(shifted) != 0.0) return Fraction{true, max_fraction_bits};
        // `shifted as i64` saturates in Rust
        int64_t si;
        if (shifted >= 9223372036854775808.0) si = INT64_MAX;
        else if (shifted <= -9223372036854775808.0) si = INT64_MIN;
        else si = (int64_t)shifted;
        size_t tz = si == 0 ? 64 : (size_t)__builtin_ctzll((uint64_t)si);
        size_t these_bits = max_fraction_bits > tz ? max_fraction_bits - tz : 0;  // saturating_sub
        if (these_bits > fraction_bits) fraction_bits = these_bits;
    }
    return Fraction{false, fraction_bits};
}

// ---------------------------------------------------------------- MMBuffer3 (typed strided view <-> i64)
struct MMBuffer3 {  // mmbuffer.rs:255-260 (enum of typed ndarray views)
    void* base = nullptr;
    int encoding = ENC_I64;
    int64_t stride[3] = {0, 0, 0};  // in elements
    size_t shape[3] = {0, 0, 0};
    size_t fractional_bits = 0;
    bool round = false;

    inline int64_t off(size_t i, size_t r, size_t c) const {
        if (i >= shape[0] || r >= shape[1] || c >= shape[2]) throw Panic(ERR_BOUNDS, "buffer index out of bounds");
        return (int64_t)i * stride[0] + (int64_t)r * stride[1] + (int64_t)c * stride[2];
    }
    int64_t get(size_t i, size_t r, size_t c) const {  // mmbuffer.rs:301-308,565-571,627-633
        int64_t o = off(i, r, c);
        switch (encoding) {
            case ENC_I32: return (int64_t)((const int32_t*)base)[o];
            case ENC_I64: return ((const int64_t*)base)[o];
            case ENC_F32: return to_fixed<float>(((const float*)base)[o], fractional_bits, round);
            case ENC_F64: return to_fixed<double>(((const double*)base)[o], fractional_bits, round);
        }
        throw Panic(ERR_BAD_ARG, "bad encoding");
    }
    void set(size_t i, size_t r, size_t c, int64_t value) {  // mmbuffer.rs:292-299,505,525,560,622
        int64_t o = off(i, r, c);
        switch (encoding) {
            case ENC_I32: ((int32_t*)base)[o] = (int32_t)value; return;  // wrapping `as i32`
            case ENC_I64: ((int64_t*)base)[o] = value; return;
            case ENC_F32: ((float*)base)[o] = from_fixed<float>(value, fractional_bits); return;
            case ENC_F64: ((double*)base)[o] = from_fixed<double>(value, fractional_bits); return;
        }
        throw Panic(ERR_BAD_ARG, "bad encoding");
    }
};

// ---------------------------------------------------------------- geom
struct Rect {  // geom.rs:4-23 (bounds auto-reordered)
    size_t top, bottom, left, right;
    Rect(size_t t, size_t b, size_t l, size_t r) {
        if (t > b) std::swap(t, b);
        if (l > r) std::swap(l, r);
        top = t; bottom = b; left = l; right = r;
    }
    size_t rows() const { return bottom - top; }
    size_t cols() const { return right - left; }
};
struct Cube {  // geom.rs:71-120
    size_t start, end, top, bottom, left, right;
    Cube(size_t s, size_t e, size_t t, size_t b, size_t l, size_t r) {
        if (s > e) std::swap(s, e);
        if (t > b) std::swap(t, b);
        if (l > r) std::swap(l, r);
        start = s; end = e; top = t; bottom = b; left = l; right = r;
    }
    Rect rect() const { return Rect(top, bottom, left, right); }
};

// snapshot.rs:118-119 / log.rs:124-125: k.pow(ceil(ln(max)/ln(k)) as u32)
inline size_t compute_sidelen(size_t rows, size_t cols, int k) {
    double s = (double)std::max(rows, cols);
    double e = std::ceil(std::log(s) / std::log((double)k));
    uint32_t exp = (e > 0 && std::isfinite(e)) ? (uint32_t)e : 0;  // saturating `as u32`
    int64_t v = 1;
    for (uint32_t i = 0; i < exp; i++) v *= k;  // i32 pow in the reference; fine at our sizes
    return (size_t)v;
}

using OptI64 = std::optional<int64_t>;

// ---------------------------------------------------------------- Snapshot
struct K2TreeNode {  // snapshot.rs:425-429
    OptI64 max, min;
    std::vector<K2TreeNode> children;

    template <class G>
    static K2TreeNode build(const G& get, const size_t shape[2], size_t k, size_t sidelen, size_t row, size_t col) {
        // snapshot.rs:439-500
        if (sidelen == 1) {
            OptI64 value;
            if (row < shape[0] && col < shape[1]) value = get(row, col);
            K2TreeNode n;
            n.max = value;
            n.min = value;
            return n;
        }
        std::vector<K2TreeNode> children;
        sidelen = sidelen / k;
        for (size_t i = 0; i < k; i++) {
            size_t row_ = row + i * sidelen;
            for (size_t j = 0; j < k; j++) {
                size_t col_ = col + j * sidelen;
                children.push_back(K2TreeNode::build(get, shape, k, sidelen, row_, col_));
            }
        }
        OptI64 max = children[0].max;
        OptI64 min = children[0].min;
        for (size_t c = 1; c < children.size(); c++) {
            const K2TreeNode& child = children[c];
            if (child.max) {
                if (max) {
                    if (*child.max > *max) max = child.max;
                } else {
                    max = child.max;
                }
            }
            if (child.min) {
                if (min) {
                    if (*child.min < *min) min = child.min;
                } else {
                    min = child.min;
                }
            }
        }
        K2TreeNode n;
        n.min = min;
        n.max = max;
        n.children = std::move(children);
        return n;
    }
};

struct Snapshot {  // snapshot.rs:20-42
    BitMap nodemap;
    Dac max, min;
    int k = 2;
    size_t shape[2] = {0, 0};
    size_t sidelen = 0;

    template <class G>
    static Snapshot build(const G& get, const size_t shape2[2], int k) {  // snapshot.rs:108-156
        BitMapBuilder nodemap;
        std::vector<int64_t> max, min;
        size_t sidelen = compute_sidelen(shape2[0], shape2[1], k);
        K2TreeNode root = K2TreeNode::build(get, shape2, (size_t)k, sidelen, 0, 0);
        std::deque<std::tuple<int64_t, int64_t, const K2TreeNode*>> to_traverse;
        to_traverse.emplace_back(root.max.value_or(0), root.min.value_or(0), &root);
        while (!to_traverse.empty()) {
            auto [diff_max, diff_min, child] = to_traverse.front();
            to_traverse.pop_front();
            int64_t child_max = child->max.value_or(0);
            int64_t child_min = child->min.value_or(0);
            max.push_back(diff_max);
            if (!child->children.empty()) {
                bool elide = child_min == child_max;
                nodemap.push(!elide);
                if (!elide) {
                    min.push_back(diff_min);
                    for (const K2TreeNode& d : child->children) {
                        to_traverse.emplace_back(child_max - d.max.value_or(0), d.min.value_or(0) - child_min, &d);
                    }
                }
            }
        }
        Snapshot s;
        s.nodemap = nodemap.finish();
        s.max = Dac::from(max);
        s.min = Dac::from(min);
        s.k = k;
        s.shape[0] = shape2[0];
        s.shape[1] = shape2[1];
        s.sidelen = sidelen;
        return s;
    }

    uint64_t size() const { return 1 + 4 + 4 + 4 + nodemap.size() + max.size() + min.size(); }  // snapshot.rs:87-92
    void write_to(Writer& w) const {                                                           // snapshot.rs:48-58
        w.write_byte((uint8_t)k);
        w.write_u32((uint32_t)shape[0]);
        w.write_u32((uint32_t)shape[1]);
        w.write_u32((uint32_t)sidelen);
        nodemap.write_to(w);
        max.write_to(w);
        min.write_to(w);
    }
    static Snapshot read_from(Reader& r) {  // snapshot.rs:62-81
        Snapshot s;
        s.k = r.read_byte();
        s.shape[0] = r.read_u32();
        s.shape[1] = r.read_u32();
        s.sidelen = r.read_u32();
        s.nodemap = BitMap::read_from(r);
        s.max = Dac::read_from(r);
        s.min = Dac::read_from(r);
        return s;
    }

    int64_t get(size_t row, size_t col) const {  // snapshot.rs:165-172
        if (!nodemap.get(0)) return max.get(0);
        return _get(sidelen, row, col, 0, max.get(0));
    }
    int64_t _get(size_t sl, size_t row, size_t col, size_t index, int64_t max_value) const {  // snapshot.rs:174-188
        size_t kk = (size_t)k;
        sl = sl / kk;
        index = 1 + nodemap.rank(index) * kk * kk;
        index = index + row / sl * kk + col / sl;
        max_value = max_value - max.get(index);
        if (index >= nodemap.length || !nodemap.get(index)) return max_value;
        return _get(sl, row % sl, col % sl, index, max_value);
    }

    template <class S>
    void fill_window(S&& set, const Rect& b) const {  // snapshot.rs:204-235
        if (!nodemap.get(0)) {
            int64_t value = max.get(0);
            for (size_t row = 0; row < b.rows(); row++)
                for (size_t col = 0; col < b.cols(); col++) set(row, col, value);
        } else {
            _fill_window(set, sidelen, b.top, b.bottom - 1, b.left, b.right - 1, 0, max.get(0), b.top, b.left, 0, 0);
        }
    }
    template <class S>
    void _fill_window(S& set, size_t sl, size_t top, size_t bottom, size_t left, size_t right, size_t index,
                      int64_t max_value, size_t window_top, size_t window_left, size_t top_offset,
                      size_t left_offset) const {  // snapshot.rs:237-301
        size_t kk = (size_t)k;
        sl = sl / kk;
        index = 1 + nodemap.rank(index) * kk * kk;
        for (size_t i = top / sl; i <= bottom / sl; i++) {
            size_t top_ = top > i * sl ? top - i * sl : 0;
            size_t bottom_ = std::min(sl - 1, bottom - i * sl);
            size_t top_offset_ = top_offset + i * sl;
            for (size_t j = left / sl; j <= right / sl; j++) {
                size_t left_ = left > j * sl ? left - j * sl : 0;
                size_t right_ = std::min(sl - 1, right - j * sl);
                size_t left_offset_ = left_offset + j * sl;
                size_t index_ = index + i * kk + j;
                int64_t max_value_ = max_value - max.get(index_);
                if (index_ >= nodemap.length || !nodemap.get(index_)) {
                    for (size_t row = top_; row <= bottom_; row++)
                        for (size_t col = left_; col <= right_; col++)
                            set(top_offset_ + row - window_top, left_offset_ + col - window_left, max_value_);
                } else {
                    _fill_window(set, sl, top_, bottom_, left_, right_, index_, max_value_, window_top, window_left,
                                 top_offset_, left_offset_);
                }
            }
        }
    }

    std::vector<std::pair<size_t, size_t>> search_window(const Rect& b, int64_t lower, int64_t upper) const {
        // snapshot.rs:310-345
        std::vector<std::pair<size_t, size_t>> cells;
        if (!nodemap.get(0)) {
            int64_t value = max.get(0);
            if (lower <= value && value <= upper) {
                for (size_t row = b.top; row < b.bottom; row++)
                    for (size_t col = b.left; col < b.right; col++) cells.emplace_back(row, col);
            }
        } else {
            _search_window(sidelen, b.top, b.bottom - 1, b.left, b.right - 1, lower, upper, 0, min.get(0), max.get(0),
                           cells, 0, 0);
        }
        return cells;
    }
    void _search_window(size_t sl, size_t top, size_t bottom, size_t left, size_t right, int64_t lower, int64_t upper,
                        size_t index, int64_t min_value, int64_t max_value,
                        std::vector<std::pair<size_t, size_t>>& cells, size_t top_offset, size_t left_offset) const {
        // snapshot.rs:347-421
        size_t kk = (size_t)k;
        sl = sl / kk;
        index = 1 + nodemap.rank(index) * kk * kk;
        for (size_t i = top / sl; i <= bottom / sl; i++) {
            size_t top_ = top > i * sl ? top - i * sl : 0;
            size_t bottom_ = std::min(sl - 1, bottom - i * sl);
            size_t top_offset_ = top_offset + i * sl;
            for (size_t j = left / sl; j <= right / sl; j++) {
                size_t left_ = left > j * sl ? left - j * sl : 0;
                size_t right_ = std::min(sl - 1, right - j * sl);
                size_t left_offset_ = left_offset + j * sl;
                size_t index_ = index + i * kk + j;
                int64_t max_value_ = max_value - max.get(index_);
                if (index_ >= nodemap.length || !nodemap.get(index_)) {
                    if (lower <= max_value_ && max_value_ <= upper) {
                        for (size_t row = top_; row <= bottom_; row++)
                            for (size_t col = left_; col <= right_; col++)
                                cells.emplace_back(top_offset_ + row, left_offset_ + col);
                    }
                } else {
                    int64_t min_value_ = min_value + min.get(nodemap.rank(index_));
                    if (lower <= min_value && max_value_ <= upper) {  // sic: parent min_value, snapshot.rs:392
                        for (size_t row = top_; row <= bottom_; row++)
                            for (size_t col = left_; col <= right_; col++)
                                cells.emplace_back(top_offset_ + row, left_offset_ + col);
                    } else if (upper >= min_value_ && lower <= max_value_) {
                        _search_window(sl, top_, bottom_, left_, right_, lower, upper, index_, min_value_, max_value_,
                                       cells, top_offset_, left_offset_);
                    }
                }
            }
        }
    }
};

// ---------------------------------------------------------------- Log
struct K2PTreeNode {  // log.rs:706-714
    OptI64 max_t, min_t, max_s, min_s;
    int64_t diff = 0;
    bool equal = true;
    std::vector<K2PTreeNode> children;

    static bool is_lt(const OptI64& l, const OptI64& r) {  // log.rs:783-790
        if (l && r) return *l < *r;
        return false;
    }
    template <class GS, class GT>
    static K2PTreeNode build(const GS& get_s, const GT& get_t, const size_t shape[2], size_t k, size_t sidelen,
                             size_t row, size_t col) {  // log.rs:725-817
        if (sidelen == 1) {
            OptI64 value_s, value_t;
            if (row < shape[0] && col < shape[1]) value_s = get_s(row, col);
            if (row < shape[0] && col < shape[1]) value_t = get_t(row, col);
            K2PTreeNode n;
            n.diff = value_t.value_or(0) - value_s.value_or(0);
            n.max_t = value_t;
            n.min_t = value_t;
            n.max_s = value_s;
            n.min_s = value_s;
            n.equal = true;
            return n;
        }
        std::vector<K2PTreeNode> children;
        sidelen = sidelen / k;
        for (size_t i = 0; i < k; i++) {
            size_t row_ = row + i * sidelen;
            for (size_t j = 0; j < k; j++) {
                size_t col_ = col + j * sidelen;
                children.push_back(K2PTreeNode::build(get_s, get_t, shape, k, sidelen, row_, col_));
            }
        }
        OptI64 max_t = children[0].max_t, min_t = children[0].min_t;
        OptI64 max_s = children[0].max_s, min_s = children[0].min_s;
        bool equal = true;
        for (const auto& c : children) equal = equal && c.equal;
        int64_t diff = children[0].diff;
        for (size_t c = 1; c < children.size(); c++) {
            const K2PTreeNode& child = children[c];
            if (is_lt(max_t, child.max_t)) max_t = child.max_t;
            if (is_lt(child.min_t, min_t)) min_t = child.min_t;
            if (is_lt(max_s, child.max_s)) max_s = child.max_s;
            if (is_lt(child.min_s, min_s)) min_s = child.min_s;
            equal = equal && child.diff == diff;
        }
        K2PTreeNode n;
        n.min_t = min_t; n.max_t = max_t; n.min_s = min_s; n.max_s = max_s;
        n.diff = diff;
        n.equal = equal;
        n.children = std::move(children);
        return n;
    }
};

struct Log {  // log.rs:21-47
    BitMap nodemap, equal;
    Dac max, min;
    int k = 2;
    size_t shape[2] = {0, 0};
    size_t sidelen = 0;

    template <class GS, class GT>
    static Log build(const GS& get_s, const GT& get_t, const size_t shape2[2], int k) {  // log.rs:112-165
        BitMapBuilder nodemap, equal;
        std::vector<int64_t> max, min;
        size_t sidelen = compute_sidelen(shape2[0], shape2[1], k);
        K2PTreeNode root = K2PTreeNode::build(get_s, get_t, shape2, (size_t)k, sidelen, 0, 0);
        std::deque<const K2PTreeNode*> to_traverse;
        to_traverse.push_back(&root);
        while (!to_traverse.empty()) {
            const K2PTreeNode* node = to_traverse.front();
            to_traverse.pop_front();
            max.push_back(node->max_t.value_or(0) - node->max_s.value_or(0));
            if (!node->children.empty()) {
                if (node->min_t == node->max_t) {  // Option equality: None == None
                    nodemap.push(false);
                    equal.push(false);
                } else if (node->equal) {
                    nodemap.push(false);
                    equal.push(true);
                } else {
                    nodemap.push(true);
                    min.push_back(node->min_t.value() - node->min_s.value());
                    for (const auto& child : node->children) to_traverse.push_back(&child);
                }
            }
        }
        Log l;
        l.nodemap = nodemap.finish();
        l.equal = equal.finish();
        l.max = Dac::from(max);
        l.min = Dac::from(min);
        l.k = k;
        l.shape[0] = shape2[0];
        l.shape[1] = shape2[1];
        l.sidelen = sidelen;
        return l;
    }

    uint64_t size() const {  // log.rs:95-97
        return 1 + 4 + 4 + 4 + nodemap.size() + equal.size() + max.size() + min.size();
    }
    void write_to(Writer& w) const {  // log.rs:53-64
        w.write_byte((uint8_t)k);
        w.write_u32((uint32_t)shape[0]);
        w.write_u32((uint32_t)shape[1]);
        w.write_u32((uint32_t)sidelen);
        nodemap.write_to(w);
        equal.write_to(w);
        max.write_to(w);
        min.write_to(w);
    }
    static Log read_from(Reader& r) {  // log.rs:68-89
        Log l;
        l.k = r.read_byte();
        l.shape[0] = r.read_u32();
        l.shape[1] = r.read_u32();
        l.sidelen = r.read_u32();
        l.nodemap = BitMap::read_from(r);
        l.equal = BitMap::read_from(r);
        l.max = Dac::read_from(r);
        l.min = Dac::read_from(r);
        return l;
    }

    using OptIdx = std::optional<size_t>;

    int64_t get(const Snapshot& snapshot, size_t row, size_t col) const {  // log.rs:176-201
        int64_t max_t = max.get(0);
        int64_t max_s = snapshot.max.get(0);
        bool single_t = !nodemap.get(0);
        bool single_s = !snapshot.nodemap.get(0);
        if (single_t && single_s) return max_t + max_s;
        if (single_t && !equal.get(0)) return max_t + max_s;
        OptIdx index_t = single_t ? OptIdx() : OptIdx(0);
        OptIdx index_s = single_s ? OptIdx() : OptIdx(0);
        return _get(snapshot, sidelen, row, col, index_t, index_s, max_t, max_s);
    }
    int64_t _get(const Snapshot& snapshot, size_t sl, size_t row, size_t col, OptIdx index_t, OptIdx index_s,
                 int64_t max_t, int64_t max_s) const {  // log.rs:203-293
        size_t kk = (size_t)k;
        sl = sl / kk;
        if (index_s) {
            size_t index = 1 + snapshot.nodemap.rank(*index_s) * kk * kk;
            index = index + row / sl * kk + col / sl;
            max_s = max_s - snapshot.max.get(index);
            index_s = index;
        }
        if (index_t) {
            size_t index = 1 + nodemap.rank(*index_t) * kk * kk;
            index = index + row / sl * kk + col / sl;
            max_t = max.get(index);
            index_t = index;
        }
        // log.rs:240,245 use `>`; SURVEY section 8(a14): differs from `>=` only when index == length,
        // where get() reads zero padding (or panics when length % 32 == 0).  Implemented as `>=`.
        bool leaf_t = index_t ? (*index_t >= nodemap.length || !nodemap.get(*index_t)) : true;
        bool leaf_s = index_s ? (*index_s >= snapshot.nodemap.length || !snapshot.nodemap.get(*index_s)) : true;
        if (leaf_t && leaf_s) {
            return max_t + max_s;
        } else if (leaf_s) {
            return _get(snapshot, sl, row % sl, col % sl, index_t, OptIdx(), max_t, max_s);
        } else if (leaf_t) {
            if (index_t) {
                if (*index_t < nodemap.length) {
                    bool eq = equal.get(nodemap.rank0(*index_t + 1) - 1);
                    if (!eq) return max_t + max_s;
                }
            }
            return _get(snapshot, sl, row % sl, col % sl, OptIdx(), index_s, max_t, max_s);
        } else {
            return _get(snapshot, sl, row % sl, col % sl, index_t, index_s, max_t, max_s);
        }
    }

    template <class S>
    void fill_window(S&& set, const Snapshot& snapshot, const Rect& b) const {  // log.rs:311-347
        bool single_t = !nodemap.get(0);
        bool single_s = !snapshot.nodemap.get(0);
        if (single_t && (single_s || !equal.get(0))) {
            int64_t max_t = max.get(0);
            int64_t max_s = snapshot.max.get(0);
            for (size_t row = 0; row < b.rows(); row++)
                for (size_t col = 0; col < b.cols(); col++) set(row, col, max_t + max_s);
        } else {
            _fill_window(set, snapshot, sidelen, b.top, b.bottom - 1, b.left, b.right - 1,
                         single_t ? OptIdx() : OptIdx(0), single_s ? OptIdx() : OptIdx(0), max.get(0),
                         snapshot.max.get(0), b.top, b.left, 0, 0);
        }
    }
    template <class S>
    void _fill_window(S& set, const Snapshot& snapshot, size_t sl, size_t top, size_t bottom, size_t left, size_t right,
                      OptIdx index_t, OptIdx index_s, int64_t max_t, int64_t max_s, size_t window_top,
                      size_t window_left, size_t top_offset, size_t left_offset) const {  // log.rs:349-508
        size_t kk = (size_t)k;
        sl = sl / kk;
        if (index_t) index_t = 1 + nodemap.rank(*index_t) * kk * kk;
        if (index_s) index_s = 1 + snapshot.nodemap.rank(*index_s) * kk * kk;
        for (size_t i = top / sl; i <= bottom / sl; i++) {
            size_t top_ = top > i * sl ? top - i * sl : 0;
            size_t bottom_ = std::min(sl - 1, bottom - i * sl);
            size_t top_offset_ = top_offset + i * sl;
            for (size_t j = left / sl; j <= right / sl; j++) {
                size_t left_ = left > j * sl ? left - j * sl : 0;
                size_t right_ = std::min(sl - 1, right - j * sl);
                size_t left_offset_ = left_offset + j * sl;
                OptIdx index_t_ = index_t ? OptIdx(*index_t + i * kk + j) : OptIdx();
                int64_t max_t_ = index_t_ ? max.get(*index_t_) : max_t;
                bool leaf_t = index_t_ ? (*index_t_ >= nodemap.length || !nodemap.get(*index_t_)) : true;
                OptIdx index_s_ = index_s ? OptIdx(*index_s + i * kk + j) : OptIdx();
                int64_t max_s_ = index_s_ ? max_s - snapshot.max.get(*index_s_) : max_s;
                bool leaf_s =
                    index_s_ ? (*index_s_ >= snapshot.nodemap.length || !snapshot.nodemap.get(*index_s_)) : true;
                auto fill = [&](int64_t value) {
                    for (size_t row = top_; row <= bottom_; row++)
                        for (size_t col = left_; col <= right_; col++)
                            set(top_offset_ + row - window_top, left_offset_ + col - window_left, value);
                };
                if (leaf_t && leaf_s) {
                    fill(max_t_ + max_s_);
                } else if (leaf_s) {
                    _fill_window(set, snapshot, sl, top_, bottom_, left_, right_, index_t_, OptIdx(), max_t_, max_s_,
                                 window_top, window_left, top_offset_, left_offset_);
                } else if (leaf_t) {
                    if (index_t_) {
                        // log.rs:453 tests !nodemap.get(index) without a length check; identical to the
                        // `index < length` guard of log.rs:264/672 wherever get() does not panic.
                        if (*index_t_ < nodemap.length && !nodemap.get(*index_t_)) {
                            bool eq = equal.get(nodemap.rank0(*index_t_ + 1) - 1);
                            if (!eq) {
                                fill(max_t_ + max_s_);
                                continue;
                            }
                        }
                    }
                    _fill_window(set, snapshot, sl, top_, bottom_, left_, right_, OptIdx(), index_s_, max_t_, max_s_,
                                 window_top, window_left, top_offset_, left_offset_);
                } else {
                    _fill_window(set, snapshot, sl, top_, bottom_, left_, right_, index_t_, index_s_, max_t_, max_s_,
                                 window_top, window_left, top_offset_, left_offset_);
                }
            }
        }
    }

    std::vector<std::pair<size_t, size_t>> search_window(const Snapshot& snapshot, const Rect& b, int64_t lower,
                                                         int64_t upper) const {  // log.rs:519-551
        std::vector<std::pair<size_t, size_t>> cells;
        bool single_t = !nodemap.get(0);
        bool single_s = !snapshot.nodemap.get(0);
        // Reference calls self.min.get(0) / snapshot.min.get(0) unconditionally (log.rs:541-542); for a
        // single-node tree the min Dac has no levels and Dac::get returns zigzag_decode(0) = 0.
        _search_window(snapshot, sidelen, b.top, b.bottom - 1, b.left, b.right - 1, lower, upper,
                       single_t ? OptIdx() : OptIdx(0), single_s ? OptIdx() : OptIdx(0), min.get(0),
                       snapshot.min.get(0), max.get(0), snapshot.max.get(0), cells, 0, 0);
        return cells;
    }
    void _search_window(const Snapshot& snapshot, size_t sl, size_t top, size_t bottom, size_t left, size_t right,
                        int64_t lower, int64_t upper, OptIdx index_t, OptIdx index_s, int64_t min_t, int64_t min_s,
                        int64_t max_t, int64_t max_s, std::vector<std::pair<size_t, size_t>>& cells,
                        size_t top_offset, size_t left_offset) const {  // log.rs:553-702
        int64_t max_value = max_s + max_t;
        int64_t min_value = min_s + min_t;
        if (min_value >= lower && max_value <= upper) {
            for (size_t row = top; row <= bottom; row++)
                for (size_t col = left; col <= right; col++) cells.emplace_back(top_offset + row, left_offset + col);
            return;
        } else if (min_value > upper || max_value < lower) {
            return;
        }
        size_t kk = (size_t)k;
        sl = sl / kk;
        if (sl == 0) return;  // unreachable for consistent structures (a cell always has min == max)
        if (index_t) index_t = 1 + nodemap.rank(*index_t) * kk * kk;
        if (index_s) index_s = 1 + snapshot.nodemap.rank(*index_s) * kk * kk;
        for (size_t i = top / sl; i <= bottom / sl; i++) {
            size_t top_ = top > i * sl ? top - i * sl : 0;
            size_t bottom_ = std::min(sl - 1, bottom - i * sl);
            size_t top_offset_ = top_offset + i * sl;
            for (size_t j = left / sl; j <= right / sl; j++) {
                size_t left_ = left > j * sl ? left - j * sl : 0;
                size_t right_ = std::min(sl - 1, right - j * sl);
                size_t left_offset_ = left_offset + j * sl;
                OptIdx index_t_ = index_t ? OptIdx(*index_t + i * kk + j) : OptIdx();
                OptIdx index_s_ = index_s ? OptIdx(*index_s + i * kk + j) : OptIdx();
                int64_t max_t_ = index_t_ ? max.get(*index_t_) : max_t;
                int64_t max_s_ = index_s_ ? max_s - snapshot.max.get(*index_s_) : max_s;
                bool leaf_t = index_t_ ? (*index_t_ >= nodemap.length || !nodemap.get(*index_t_)) : true;
                bool leaf_s =
                    index_s_ ? (*index_s_ >= snapshot.nodemap.length || !snapshot.nodemap.get(*index_s_)) : true;
                int64_t min_t_ = index_t_ ? (leaf_t ? min_t : min.get(nodemap.rank(*index_t_))) : min_t;
                int64_t min_s_ =
                    index_s_ ? (leaf_s ? min_s : min_s + snapshot.min.get(snapshot.nodemap.rank(*index_s_))) : min_s;
                if (leaf_s) {
                    min_s_ = max_s_;
                    index_s_ = OptIdx();
                }
                if (leaf_t) {
                    min_t_ = max_t_;
                    if (index_t_) {
                        if (*index_t_ < nodemap.length && !equal.get(nodemap.rank0(*index_t_ + 1) - 1)) {
                            min_t_ = max_s_ + max_t_ - min_s_;
                        }
                    }
                    index_t_ = OptIdx();
                }
                _search_window(snapshot, sl, top_, bottom_, left_, right_, lower, upper, index_t_, index_s_, min_t_,
                               min_s_, max_t_, max_s_, cells, top_offset_, left_offset_);
            }
        }
    }
};

// ---------------------------------------------------------------- Block
struct Block {  // block.rs:15-21
    Snapshot snapshot;
    std::vector<Log> logs;

    Block() = default;
    Block(Snapshot s, std::vector<Log> l) : snapshot(std::move(s)), logs(std::move(l)) {  // block.rs:26-38
        if (logs.size() > 254) throw Panic(ERR_TOO_MANY_LOGS, "too many logs in one block");
    }
    int64_t get(size_t instant, size_t row, size_t col) const {  // block.rs:42-47
        if (instant == 0) return snapshot.get(row, col);
        return logs.at(instant - 1).get(snapshot, row, col);
    }
    template <class S>
    void fill_window(S&& set, size_t instant, const Rect& b) const {  // block.rs:56-64
        if (instant == 0) snapshot.fill_window(set, b);
        else logs.at(instant - 1).fill_window(set, snapshot, b);
    }
    std::vector<std::pair<size_t, size_t>> search_window(size_t instant, const Rect& b, int64_t lower,
                                                         int64_t upper) const {  // block.rs:70-81
        if (instant == 0) return snapshot.search_window(b, lower, upper);
        return logs.at(instant - 1).search_window(snapshot, b, lower, upper);
    }
    uint64_t size() const {  // block.rs:114-118
        uint64_t s = 1 + snapshot.size();
        for (const Log& l : logs) s += l.size();
        return s;
    }
    void write_to(Writer& w) const {  // block.rs:88-95
        w.write_byte((uint8_t)(logs.size() + 1));
        snapshot.write_to(w);
        for (const Log& l : logs) l.write_to(w);
    }
    static Block read_from(Reader& r) {  // block.rs:99-109
        Block b;
        size_t n_instants = r.read_byte();
        if (n_instants == 0) throw Panic(ERR_FORMAT, "block with zero instants");
        b.snapshot = Snapshot::read_from(r);
        for (size_t i = 0; i + 1 < n_instants; i++) b.logs.push_back(Log::read_from(r));
        return b;
    }
};

// ---------------------------------------------------------------- Chunk
struct ChunkBuild;

struct Chunk {  // chunk.rs:24-39
    std::vector<Block> blocks;
    std::vector<size_t> index;
    int encoding = ENC_I64;
    size_t fractional_bits = 0;

    Chunk() = default;
    Chunk(std::vector<Block> b, int enc, size_t fb) : blocks(std::move(b)), encoding(enc), fractional_bits(fb) {
        // chunk.rs:100-114
        size_t count = 0;
        for (const Block& blk : blocks) {
            count += blk.logs.size() + 1;
            index.push_back(count);
        }
    }

    static ChunkBuild build(const MMBuffer3& buffer, const size_t shape[3], int k);

    void shape(size_t out[3]) const {  // chunk.rs:119-123
        out[1] = blocks.at(0).snapshot.shape[0];
        out[2] = blocks.at(0).snapshot.shape[1];
        out[0] = 0;
        for (const Block& b : blocks) out[0] += 1 + b.logs.size();
    }

    std::pair<size_t, size_t> find_block(size_t instant) const {  // chunk.rs:164-191
        if (index.empty()) throw Panic(ERR_BOUNDS, "empty chunk");
        if (instant < index[0]) return {0, instant};
        size_t lower = 0, upper = blocks.size();
        size_t idx = upper / 2;
        for (;;) {
            if (idx >= index.size()) throw Panic(ERR_BOUNDS, "instant out of bounds");
            size_t here = index[idx];
            if (here == instant) {
                idx += 1;
                break;
            } else if (here < instant) {
                if (lower == idx) throw Panic(ERR_BOUNDS, "instant out of bounds");  // reference would spin
                lower = idx;
            } else {
                if (index.at(idx - 1) <= instant) break;
                upper = idx;
            }
            idx = (lower + upper) / 2;
        }
        if (idx >= blocks.size()) throw Panic(ERR_BOUNDS, "instant out of bounds");
        return {idx, instant - index[idx - 1]};
    }

    // ChunkIter, chunk.rs:284-313
    struct Iter {
        const Chunk* chunk;
        size_t block, instant, remaining;
        bool next(size_t& b, size_t& i) {
            if (remaining == 0) return false;
            b = block;
            i = instant;
            const Block& blk = chunk->blocks.at(block);
            if (instant == blk.logs.size()) {
                instant = 0;
                block += 1;
            } else {
                instant += 1;
            }
            remaining -= 1;
            return true;
        }
    };
    Iter iter(size_t start, size_t end) const {  // chunk.rs:197-206
        if (end == start) return Iter{this, 0, 0, 0};
        auto [block, instant] = find_block(start);
        return Iter{this, block, instant, end - start};
    }

    int64_t get(size_t instant, size_t row, size_t col) const {  // chunk.rs:127-131
        auto [block, inst] = find_block(instant);
        return blocks[block].get(inst, row, col);
    }
    void fill_cell(size_t start, size_t end, size_t row, size_t col, std::vector<int64_t>& out) const {
        // chunk.rs:135-148
        Iter it = iter(start, end);
        size_t b, i;
        while (it.next(b, i)) out.push_back(blocks[b].get(i, row, col));
    }
    template <class S3>
    void fill_window(const Cube& bounds, S3&& set3) const {  // chunk.rs:152-158
        Iter it = iter(bounds.start, bounds.end);
        size_t b, i, n = 0;
        Rect rect = bounds.rect();
        while (it.next(b, i)) {
            auto set2d = [&](size_t row, size_t col, int64_t value) { set3(n, row, col, value); };
            blocks[b].fill_window(set2d, i, rect);
            n++;
        }
    }
    // iter_search + SearchIter, chunk.rs:213-229,336-383: yields (instant,row,col) in emission order
    std::vector<std::tuple<size_t, size_t, size_t>> search(const Cube& bounds, int64_t lower, int64_t upper) const {
        if (lower > upper) std::swap(lower, upper);  // helpers.rs:7-16 via chunk.rs:214
        std::vector<std::tuple<size_t, size_t, size_t>> out;
        Iter it = iter(bounds.start, bounds.end);
        size_t b, i;
        size_t instant = bounds.start;
        Rect rect = bounds.rect();
        while (it.next(b, i)) {
            auto cells = blocks[b].search_window(i, rect, lower, upper);
            for (auto& [row, col] : cells) out.emplace_back(instant, row, col);
            instant++;
        }
        return out;
    }

    uint64_t size() const {  // chunk.rs:272-277
        uint64_t s = 1 + 1 + 4;
        for (const Block& b : blocks) s += b.size();
        return s;
    }
    void write_to(Writer& w) const {  // chunk.rs:235-243
        w.write_byte((uint8_t)encoding);
        w.write_byte((uint8_t)fractional_bits);
        w.write_u32((uint32_t)blocks.size());
        for (const Block& b : blocks) b.write_to(w);
    }
    static Chunk read_from(Reader& r) {  // chunk.rs:247-266
        Chunk c;
        int enc = r.read_byte();
        if (enc != ENC_I32 && enc != ENC_I64 && enc != ENC_F32 && enc != ENC_F64)
            throw Panic(ERR_FORMAT, "bad encoding byte");  // mmstruct.rs:49-57
        c.encoding = enc;
        c.fractional_bits = r.read_byte();
        size_t n_blocks = r.read_u32();
        size_t count = 0;
        for (size_t i = 0; i < n_blocks; i++) {
            Block b = Block::read_from(r);
            count += b.logs.size() + 1;
            c.blocks.push_back(std::move(b));
            c.index.push_back(count);
        }
        return c;
    }
};

struct ChunkBuild {  // mmstruct.rs:24-34 (MMStruct3Build, Subchunk arm)
    Chunk data;
    uint64_t size = 0;
    size_t snapshots = 0, logs = 0;
    // instants that start a block (not in the reference; exposed for parity diagnostics)
    std::vector<uint32_t> snapshot_instants;
};

inline ChunkBuild Chunk::build(const MMBuffer3& buffer, const size_t shape[3], int k) {  // chunk.rs:42-96
    size_t count_snapshots = 0, count_logs = 0;
    size_t instants = shape[0];
    size_t shape2[2] = {shape[1], shape[2]};
    std::vector<Block> blocks;
    std::vector<uint32_t> snap_instants;

    auto first_get = [&](size_t row, size_t col) { return buffer.get(0, row, col); };
    Snapshot snapshot = Snapshot::build(first_get, shape2, k);
    size_t snapshot_index = 0;
    std::vector<Log> logs;
    snap_instants.push_back(0);

    for (size_t i = 1; i < instants; i++) {
        auto get_t = [&](size_t row, size_t col) { return buffer.get(i, row, col); };
        Snapshot new_snapshot = Snapshot::build(get_t, shape2, k);
        auto get_s = [&](size_t row, size_t col) { return buffer.get(snapshot_index, row, col); };
        Log new_log = Log::build(get_s, get_t, shape2, k);
        if (logs.size() == 254 || new_snapshot.size() <= new_log.size()) {
            count_snapshots += 1;
            count_logs += logs.size();
            Snapshot block_snapshot = std::move(snapshot);
            snapshot = std::move(new_snapshot);
            std::vector<Log> block_logs = std::move(logs);
            logs.clear();
            snapshot_index = i;
            snap_instants.push_back((uint32_t)i);
            blocks.emplace_back(std::move(block_snapshot), std::move(block_logs));
        } else {
            logs.push_back(std::move(new_log));
        }
    }
    count_snapshots += 1;
    count_logs += logs.size();
    blocks.emplace_back(std::move(snapshot), std::move(logs));

    ChunkBuild out;
    out.data = Chunk(std::move(blocks), buffer.encoding, buffer.fractional_bits);
    out.size = out.data.size();
    out.logs = count_logs;
    out.snapshots = count_snapshots;
    out.snapshot_instants = std::move(snap_instants);
    return out;
}

}  // namespace orc
