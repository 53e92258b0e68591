// k2r_oracle_capi.cpp -- TEST INFRASTRUCTURE ONLY (see k2r_oracle.hpp header).
// C ABI over the CPU restatement so that tests (ctypes) and bench.py's
// cpu_baseline leg can call it.  Never linked into libdcdf_k2r.so.
#include <chrono>
#include <thread>
#include <cstdlib>

#include "k2r_oracle.hpp"

using namespace orc;

#define ORC_TRY try {
#define ORC_CATCH                                 \
    }                                             \
    catch (const Panic& p) { return p.code; }     \
    catch (const std::out_of_range&) { return ERR_BOUNDS; } \
    catch (const std::bad_optional_access&) { return ERR_BOUNDS; } \
    catch (const std::bad_alloc&) { return -100; } \
    catch (...) { return -101; }

static MMBuffer3 make_buffer(const void* base, int encoding, int64_t st, int64_t sr, int64_t sc, uint32_t instants,
                             uint32_t rows, uint32_t cols, int fractional_bits, int round) {
    MMBuffer3 b;
    b.base = const_cast<void*>(base);
    b.encoding = encoding;
    b.stride[0] = st; b.stride[1] = sr; b.stride[2] = sc;
    b.shape[0] = instants; b.shape[1] = rows; b.shape[2] = cols;
    b.fractional_bits = (encoding == ENC_F32 || encoding == ENC_F64) ? (size_t)fractional_bits : 0;
    b.round = round != 0;
    return b;
}

struct OrcChunk {
    Chunk chunk;
};

extern "C" {

void orc_free(void* p) { std::free(p); }

// Chunk::build + Chunk::write_to  (chunk.rs:42-96, 235-243)
int orc_chunk_build(const void* base, int encoding, int64_t st, int64_t sr, int64_t sc, uint32_t instants,
                    uint32_t rows, uint32_t cols, int k, int fractional_bits, int round, uint8_t** out_bytes,
                    size_t* out_len, uint32_t* snapshots, uint32_t* logs, uint32_t* snapshot_instants /*cap=instants, may be null*/) {
    ORC_TRY
    if (instants == 0 || k < 2) return ERR_BAD_ARG;
    if (encoding != ENC_I32 && encoding != ENC_I64 && encoding != ENC_F32 && encoding != ENC_F64) return ERR_BAD_ARG;
    MMBuffer3 buf = make_buffer(base, encoding, st, sr, sc, instants, rows, cols, fractional_bits, round);
    size_t shape[3] = {instants, rows, cols};
    ChunkBuild b = Chunk::build(buf, shape, k);
    Writer w;
    b.data.write_to(w);
    if (w.buf.size() != b.size) return -102;  // size() identity (chunk.rs:572-574)
    uint8_t* o = (uint8_t*)std::malloc(w.buf.size() ? w.buf.size() : 1);
    if (!o) return -100;
    std::memcpy(o, w.buf.data(), w.buf.size());
    *out_bytes = o;
    *out_len = w.buf.size();
    if (snapshots) *snapshots = (uint32_t)b.snapshots;
    if (logs) *logs = (uint32_t)b.logs;
    if (snapshot_instants)
        for (size_t i = 0; i < b.snapshot_instants.size(); i++) snapshot_instants[i] = b.snapshot_instants[i];
    return 0;
    ORC_CATCH
}

// Test helper mirroring chunk.rs:397-417: fixed-length blocks (snapshot + block_len-1 logs), i64 data.
// A trailing partial block is kept (the reference helper drops it; callers pass multiples).
int orc_chunk_build_forced(const int64_t* data, uint32_t instants, uint32_t rows, uint32_t cols, int k,
                           uint32_t block_len, uint8_t** out_bytes, size_t* out_len) {
    ORC_TRY
    if (instants == 0 || block_len == 0 || block_len > 255) return ERR_BAD_ARG;
    size_t shape2[2] = {rows, cols};
    std::vector<Block> blocks;
    for (uint32_t b0 = 0; b0 < instants; b0 += block_len) {
        const int64_t* s = data + (size_t)b0 * rows * cols;
        auto get_s = [&](size_t r, size_t c) { return s[r * cols + c]; };
        Snapshot snap = Snapshot::build(get_s, shape2, k);
        std::vector<Log> logs;
        for (uint32_t i = b0 + 1; i < std::min(instants, b0 + block_len); i++) {
            const int64_t* t = data + (size_t)i * rows * cols;
            auto get_t = [&](size_t r, size_t c) { return t[r * cols + c]; };
            logs.push_back(Log::build(get_s, get_t, shape2, k));
        }
        blocks.emplace_back(std::move(snap), std::move(logs));
    }
    Chunk chunk(std::move(blocks), ENC_I64, 0);
    Writer w;
    chunk.write_to(w);
    if (w.buf.size() != chunk.size()) return -102;
    uint8_t* o = (uint8_t*)std::malloc(w.buf.size());
    if (!o) return -100;
    std::memcpy(o, w.buf.data(), w.buf.size());
    *out_bytes = o;
    *out_len = w.buf.size();
    return 0;
    ORC_CATCH
}

// Chunk::read_from (chunk.rs:247-266)
int orc_chunk_open(const uint8_t* bytes, size_t len, void** handle) {
    ORC_TRY
    Reader r(bytes, len);
    OrcChunk* c = new OrcChunk();
    try {
        c->chunk = Chunk::read_from(r);
    } catch (...) {
        delete c;
        throw;
    }
    *handle = c;
    return 0;
    ORC_CATCH
}
void orc_chunk_close(void* handle) { delete (OrcChunk*)handle; }

// re-serialize (round-trip check of read_from/write_to) -> malloc'ed bytes
int orc_chunk_serialize(void* handle, uint8_t** out_bytes, size_t* out_len) {
    ORC_TRY
    Writer w;
    ((OrcChunk*)handle)->chunk.write_to(w);
    uint8_t* o = (uint8_t*)std::malloc(w.buf.size() ? w.buf.size() : 1);
    std::memcpy(o, w.buf.data(), w.buf.size());
    *out_bytes = o;
    *out_len = w.buf.size();
    return 0;
    ORC_CATCH
}

// info: [instants, rows, cols, encoding, fractional_bits, n_blocks, size()]
int orc_chunk_info(void* handle, uint64_t out[7]) {
    ORC_TRY
    const Chunk& c = ((OrcChunk*)handle)->chunk;
    size_t shp[3];
    c.shape(shp);
    out[0] = shp[0]; out[1] = shp[1]; out[2] = shp[2];
    out[3] = (uint64_t)c.encoding;
    out[4] = c.fractional_bits;
    out[5] = c.blocks.size();
    out[6] = c.size();
    return 0;
    ORC_CATCH
}

// block lengths (instants per block), cap = n_blocks
int orc_chunk_block_lengths(void* handle, uint32_t* out) {
    ORC_TRY
    const Chunk& c = ((OrcChunk*)handle)->chunk;
    for (size_t i = 0; i < c.blocks.size(); i++) out[i] = (uint32_t)(c.blocks[i].logs.size() + 1);
    return 0;
    ORC_CATCH
}

// Chunk::get (chunk.rs:127-131) -- raw stored i64
int orc_chunk_get(void* handle, uint32_t instant, uint32_t row, uint32_t col, int64_t* out) {
    ORC_TRY
    *out = ((OrcChunk*)handle)->chunk.get(instant, row, col);
    return 0;
    ORC_CATCH
}
// Chunk::fill_cell (chunk.rs:135-148) -- raw stored i64, out has end-start entries
int orc_chunk_fill_cell(void* handle, uint32_t start, uint32_t end, uint32_t row, uint32_t col, int64_t* out) {
    ORC_TRY
    std::vector<int64_t> v;
    ((OrcChunk*)handle)->chunk.fill_cell(start, end, row, col, v);
    for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
    return 0;
    ORC_CATCH
}
// Chunk::fill_window (chunk.rs:152-158) into a caller-preallocated typed strided array using
// MMBuffer3::set conversion (mmbuffer.rs:292-299): out_encoding in {4,8,32,64}; strides in elements.
int orc_chunk_fill_window(void* handle, uint32_t start, uint32_t end, uint32_t top, uint32_t bottom, uint32_t left,
                          uint32_t right, void* out, int out_encoding, int64_t st, int64_t sr, int64_t sc) {
    ORC_TRY
    const Chunk& c = ((OrcChunk*)handle)->chunk;
    Cube cube(start, end, top, bottom, left, right);
    MMBuffer3 buf = make_buffer(out, out_encoding, st, sr, sc, (uint32_t)(cube.end - cube.start),
                                (uint32_t)(cube.bottom - cube.top), (uint32_t)(cube.right - cube.left),
                                (int)c.fractional_bits, 0);
    c.fill_window(cube, [&](size_t i, size_t r, size_t cc, int64_t v) { buf.set(i, r, cc, v); });
    return 0;
    ORC_CATCH
}
// Chunk::iter_search (chunk.rs:213-229): malloc'ed (instant,row,col) u32 triples in emission order
int orc_chunk_search(void* handle, uint32_t start, uint32_t end, uint32_t top, uint32_t bottom, uint32_t left,
                     uint32_t right, int64_t lower, int64_t upper, uint32_t** triples, size_t* n) {
    ORC_TRY
    const Chunk& c = ((OrcChunk*)handle)->chunk;
    Cube cube(start, end, top, bottom, left, right);
    auto res = c.search(cube, lower, upper);
    uint32_t* o = (uint32_t*)std::malloc(res.size() * 12 + 4);
    if (!o) return -100;
    for (size_t i = 0; i < res.size(); i++) {
        o[3 * i] = (uint32_t)std::get<0>(res[i]);
        o[3 * i + 1] = (uint32_t)std::get<1>(res[i]);
        o[3 * i + 2] = (uint32_t)std::get<2>(res[i]);
    }
    *triples = o;
    *n = res.size();
    return 0;
    ORC_CATCH
}

// ---- component dumps for the golden-vector tests -------------------------------------------
static void push_bitmap(std::vector<int64_t>& o, const BitMap& b) {
    o.push_back((int64_t)b.length);
    o.push_back((int64_t)b.bitmap.size());
    for (uint32_t w : b.bitmap) o.push_back((int64_t)w);
    o.push_back((int64_t)b.index.size());
    for (uint32_t w : b.index) o.push_back((int64_t)w);
}
static void push_vec(std::vector<int64_t>& o, const std::vector<int64_t>& v) {
    o.push_back((int64_t)v.size());
    for (int64_t x : v) o.push_back(x);
}
static int give(const std::vector<int64_t>& o, int64_t** out, size_t* n) {
    int64_t* p = (int64_t*)std::malloc(o.size() * 8 + 8);
    if (!p) return -100;
    std::memcpy(p, o.data(), o.size() * 8);
    *out = p;
    *n = o.size();
    return 0;
}

// Snapshot::build on a dense row-major i64 array (testing.rs:332-340 `from_array`).
// out: T{length,nwords,words..,nindex,index..}, max{n,vals..}, min{n,vals..}, size(), serialized_len, sidelen
int orc_snapshot_dump(const int64_t* data, uint32_t rows, uint32_t cols, int k, int64_t** out, size_t* n) {
    ORC_TRY
    size_t shape2[2] = {rows, cols};
    auto get = [&](size_t r, size_t c) { return data[r * cols + c]; };
    Snapshot s = Snapshot::build(get, shape2, k);
    std::vector<int64_t> o;
    push_bitmap(o, s.nodemap);
    push_vec(o, s.max.collect());
    push_vec(o, s.min.collect());
    Writer w;
    s.write_to(w);
    o.push_back((int64_t)s.size());
    o.push_back((int64_t)w.buf.size());
    o.push_back((int64_t)s.sidelen);
    return give(o, out, n);
    ORC_CATCH
}
// Log::build (testing.rs:357-366 `from_arrays`).
// out: T{..}, eq{..}, max{..}, min{..}, size(), serialized_len, sidelen
int orc_log_dump(const int64_t* s_data, const int64_t* t_data, uint32_t rows, uint32_t cols, int k, int64_t** out,
                 size_t* n) {
    ORC_TRY
    size_t shape2[2] = {rows, cols};
    auto get_s = [&](size_t r, size_t c) { return s_data[r * cols + c]; };
    auto get_t = [&](size_t r, size_t c) { return t_data[r * cols + c]; };
    Log l = Log::build(get_s, get_t, shape2, k);
    std::vector<int64_t> o;
    push_bitmap(o, l.nodemap);
    push_bitmap(o, l.equal);
    push_vec(o, l.max.collect());
    push_vec(o, l.min.collect());
    Writer w;
    l.write_to(w);
    o.push_back((int64_t)l.size());
    o.push_back((int64_t)w.buf.size());
    o.push_back((int64_t)l.sidelen);
    return give(o, out, n);
    ORC_CATCH
}
// Snapshot/Log point + window + search on dense arrays (for the exhaustive sweeps of
// snapshot.rs:574-870, log.rs:957-1616 incl. k=3).  which=0 snapshot(s), which=1 log(s->t).
int orc_sl_get(const int64_t* s_data, const int64_t* t_data, uint32_t rows, uint32_t cols, int k, int which,
               uint32_t row, uint32_t col, int64_t* out) {
    ORC_TRY
    size_t shape2[2] = {rows, cols};
    auto get_s = [&](size_t r, size_t c) { return s_data[r * cols + c]; };
    Snapshot s = Snapshot::build(get_s, shape2, k);
    if (which == 0) {
        *out = s.get(row, col);
    } else {
        auto get_t = [&](size_t r, size_t c) { return t_data[r * cols + c]; };
        Log l = Log::build(get_s, get_t, shape2, k);
        *out = l.get(s, row, col);
    }
    return 0;
    ORC_CATCH
}
// all windows at once would be slow through ctypes; this does the full exhaustive sweep natively and
// returns the number of mismatching cells vs. the dense arrays (window) and mismatching result sets (search).
int orc_sl_exhaustive(const int64_t* s_data, const int64_t* t_data, uint32_t rows, uint32_t cols, int k, int which,
                      int64_t search_lo, int64_t search_hi, uint64_t* n_windows, uint64_t* bad_window_cells,
                      uint64_t* bad_searches, uint64_t* bad_gets) {
    ORC_TRY
    size_t shape2[2] = {rows, cols};
    auto get_s = [&](size_t r, size_t c) { return s_data[r * cols + c]; };
    auto get_t = [&](size_t r, size_t c) { return t_data[r * cols + c]; };
    Snapshot s = Snapshot::build(get_s, shape2, k);
    Log l = Log::build(get_s, get_t, shape2, k);
    const int64_t* truth = which == 0 ? s_data : t_data;
    *n_windows = *bad_window_cells = *bad_searches = *bad_gets = 0;
    for (uint32_t r = 0; r < rows; r++)
        for (uint32_t c = 0; c < cols; c++) {
            int64_t v = which == 0 ? s.get(r, c) : l.get(s, r, c);
            if (v != truth[r * cols + c]) (*bad_gets)++;
        }
    for (uint32_t top = 0; top < rows; top++)
        for (uint32_t bottom = top + 1; bottom <= rows; bottom++)
            for (uint32_t left = 0; left < cols; left++)
                for (uint32_t right = left + 1; right <= cols; right++) {
                    Rect b(top, bottom, left, right);
                    std::vector<int64_t> win(b.rows() * b.cols(), INT64_MIN);
                    auto set = [&](size_t rr, size_t cc, int64_t v) { win.at(rr * b.cols() + cc) = v; };
                    if (which == 0) s.fill_window(set, b);
                    else l.fill_window(set, s, b);
                    (*n_windows)++;
                    for (size_t rr = 0; rr < b.rows(); rr++)
                        for (size_t cc = 0; cc < b.cols(); cc++)
                            if (win[rr * b.cols() + cc] != truth[(top + rr) * cols + left + cc]) (*bad_window_cells)++;
                    for (int64_t lo = search_lo; lo <= search_hi; lo++)
                        for (int64_t hi = lo; hi <= search_hi; hi++) {
                            auto res = which == 0 ? s.search_window(b, lo, hi) : l.search_window(s, b, lo, hi);
                            std::vector<uint8_t> mark(rows * cols, 0);
                            bool bad = false;
                            for (auto& [rr, cc] : res) {
                                if (rr >= rows || cc >= cols || mark[rr * cols + cc]) { bad = true; break; }
                                mark[rr * cols + cc] = 1;
                            }
                            for (uint32_t rr = 0; rr < rows && !bad; rr++)
                                for (uint32_t cc = 0; cc < cols; cc++) {
                                    bool in = rr >= top && rr < bottom && cc >= left && cc < right &&
                                              truth[rr * cols + cc] >= lo && truth[rr * cols + cc] <= hi;
                                    if (in != (mark[rr * cols + cc] != 0)) { bad = true; break; }
                                }
                            if (bad) (*bad_searches)++;
                        }
                }
    return 0;
    ORC_CATCH
}

// BitMapBuilder{length, bytes}.finish()  (bitmap.rs:261-284 style construction)
int orc_bitmap_dump(uint64_t length, const uint8_t* bytes, size_t nbytes, int64_t** out, size_t* n) {
    ORC_TRY
    BitMapBuilder b;
    b.length = length;
    b.bitmap.assign(bytes, bytes + nbytes);
    BitMap bm = b.finish();
    std::vector<int64_t> o;
    push_bitmap(o, bm);
    o.push_back((int64_t)bm.size());
    return give(o, out, n);
    ORC_CATCH
}
// rank via index vs naive rank (bitmap.rs:286-317); also get(i)
int orc_bitmap_rank(uint64_t length, const uint8_t* bytes, size_t nbytes, uint64_t i, uint64_t* rank,
                    uint64_t* naive, int* bit) {
    ORC_TRY
    BitMapBuilder b;
    b.length = length;
    b.bitmap.assign(bytes, bytes + nbytes);
    BitMap bm = b.finish();
    *rank = bm.rank(i);
    *naive = b.naive_rank(i);
    *bit = i < length ? (bm.get(i) ? 1 : 0) : -1;
    return 0;
    ORC_CATCH
}
// bit-at-a-time push path (bitmap.rs:44-62)
int orc_bitmap_push_dump(const uint8_t* bits, size_t nbits, int64_t** out, size_t* n) {
    ORC_TRY
    BitMapBuilder b;
    for (size_t i = 0; i < nbits; i++) b.push(bits[i] != 0);
    BitMap bm = b.finish();
    std::vector<int64_t> o;
    push_bitmap(o, bm);
    o.push_back((int64_t)bm.size());
    return give(o, out, n);
    ORC_CATCH
}
// Dac::from(values) -> out: n_levels, per level {bitmap{..}, nbytes, bytes..}, collect{n, vals..}, size(), serialized_len
int orc_dac_dump(const int64_t* values, size_t nvalues, int64_t** out, size_t* n) {
    ORC_TRY
    Dac d = Dac::from(std::vector<int64_t>(values, values + nvalues));
    std::vector<int64_t> o;
    o.push_back((int64_t)d.levels.size());
    for (auto& [bm, bytes] : d.levels) {
        push_bitmap(o, bm);
        o.push_back((int64_t)bytes.size());
        for (uint8_t b : bytes) o.push_back(b);
    }
    push_vec(o, d.collect());
    Writer w;
    d.write_to(w);
    o.push_back((int64_t)d.size());
    o.push_back((int64_t)w.buf.size());
    // read back and collect again
    Reader r(w.buf.data(), w.buf.size());
    Dac d2 = Dac::read_from(r);
    push_vec(o, d2.collect());
    return give(o, out, n);
    ORC_CATCH
}

// Dac::from + Dac::write_to (dac.rs:37-44,101-131): the serialized bytes
int orc_dac_serialize(const int64_t* values, size_t nvalues, uint8_t** out_bytes, size_t* out_len) {
    ORC_TRY
    Dac d = Dac::from(std::vector<int64_t>(values, values + nvalues));
    Writer w;
    d.write_to(w);
    uint8_t* p = (uint8_t*)std::malloc(w.buf.size() ? w.buf.size() : 1);
    std::memcpy(p, w.buf.data(), w.buf.size());
    *out_bytes = p;
    *out_len = w.buf.size();
    return 0;
    ORC_CATCH
}

int orc_to_fixed_f32(float v, int bits, int round, int64_t* out) {
    ORC_TRY
    *out = to_fixed<float>(v, (size_t)bits, round != 0);
    return 0;
    ORC_CATCH
}
int orc_to_fixed_f64(double v, int bits, int round, int64_t* out) {
    ORC_TRY
    *out = to_fixed<double>(v, (size_t)bits, round != 0);
    return 0;
    ORC_CATCH
}
float orc_from_fixed_f32(int64_t v, int bits) { return from_fixed<float>(v, (size_t)bits); }
double orc_from_fixed_f64(int64_t v, int bits) { return from_fixed<double>(v, (size_t)bits); }
int orc_suggest_fraction_f32(const float* data, size_t n, int* round, int* bits) {
    ORC_TRY
    Fraction f = suggest_fraction<float>(data, data + n);
    *round = f.round ? 1 : 0;
    *bits = (int)f.bits;
    return 0;
    ORC_CATCH
}
int orc_suggest_fraction_f64(const double* data, size_t n, int* round, int* bits) {
    ORC_TRY
    Fraction f = suggest_fraction<double>(data, data + n);
    *round = f.round ? 1 : 0;
    *bits = (int)f.bits;
    return 0;
    ORC_CATCH
}
uint64_t orc_sidelen(uint32_t rows, uint32_t cols, int k) { return compute_sidelen(rows, cols, k); }

// ---- timed batch build for bench.py's cpu_baseline leg (kind = "port") ----------------------------
// Builds `n_chunks` chunks (each [instants, rows, cols], dense, consecutive in `base`) serially on the
// calling thread, exactly as the reference does (superchunk.rs:166-188), and returns elapsed seconds
// for build + serialize.  Optionally returns a 64-bit FNV-1a over all serialized bytes.
// Timed batches of chunk-level queries for the decode-path CPU baseline (tools/bench_query.py): query q =
// Chunk::fill_window (chunk.rs:152-158, into an int32 array as MMBuffer3::set does for I32 chunks) or Chunk::iter_search
// (chunk.rs:213-229) of chunks[q] on cubes[6 q ..]; `threads` workers take the queries in strides.  The clock runs inside.
int orc_bench_queries(void* const* chunks, const uint32_t* cubes, const int64_t* lower, const int64_t* upper, size_t n, int search,
                      int threads, double* seconds, uint64_t* work) {
    ORC_TRY
    if (threads < 1) threads = 1;
    std::vector<uint64_t> acc((size_t)threads, 0);
    std::vector<int> failed((size_t)threads, 0);
    auto body = [&](int t) {
        try {
            std::vector<int32_t> out;
            for (size_t q = (size_t)t; q < n; q += (size_t)threads) {
                const Chunk& c = ((OrcChunk*)chunks[q])->chunk;
                const uint32_t* u = cubes + 6 * q;
                Cube cube(u[0], u[1], u[2], u[3], u[4], u[5]);
                if (search) {
                    acc[t] += c.search(cube, lower[q], upper[q]).size();
                } else {
                    const size_t wr = cube.bottom - cube.top, wc = cube.right - cube.left, wt = cube.end - cube.start;
                    out.resize(wt * wr * wc);
                    c.fill_window(cube, [&](size_t i, size_t r, size_t cc, int64_t v) { out[(i * wr + r) * wc + cc] = (int32_t)v; });
                    acc[t] += out.size();
                }
            }
        } catch (...) {
            failed[t] = 1;
        }
    };
    auto t0 = std::chrono::steady_clock::now();
    if (threads == 1) body(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++) th.emplace_back(body, t);
        for (auto& x : th) x.join();
    }
    auto t1 = std::chrono::steady_clock::now();
    uint64_t tot = 0;
    for (int t = 0; t < threads; t++) {
        if (failed[t]) return -101;
        tot += acc[t];
    }
    *seconds = std::chrono::duration<double>(t1 - t0).count();
    *work = tot;
    return 0;
    ORC_CATCH
}

int orc_bench_build(const void* base, int encoding, uint32_t n_chunks, uint32_t instants, uint32_t rows, uint32_t cols,
                    int k, double* seconds, uint64_t* total_bytes, uint64_t* fnv) {
    ORC_TRY
    size_t esz = (encoding == ENC_I32 || encoding == ENC_F32) ? 4 : 8;
    uint64_t h = 1469598103934665603ULL, tb = 0;
    double acc = 0.0;
    for (uint32_t c = 0; c < n_chunks; c++) {
        const uint8_t* p = (const uint8_t*)base + (size_t)c * instants * rows * cols * esz;
        MMBuffer3 buf = make_buffer(p, encoding, (int64_t)rows * cols, cols, 1, instants, rows, cols, 0, 0);
        size_t shape[3] = {instants, rows, cols};
        auto t0 = std::chrono::steady_clock::now();
        ChunkBuild b = Chunk::build(buf, shape, k);
        Writer w;
        b.data.write_to(w);
        auto t1 = std::chrono::steady_clock::now();
        acc += std::chrono::duration<double>(t1 - t0).count();
        tb += w.buf.size();
        for (uint8_t x : w.buf) { h ^= x; h *= 1099511628211ULL; }  // checksum is outside the timed region
    }
    *seconds = acc;
    if (total_bytes) *total_bytes = tb;
    if (fnv) *fnv = h;
    return 0;
    ORC_CATCH
}

}  // extern "C"
