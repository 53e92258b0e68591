// k2r_query.hip -- C ABI, query side: open (parse + upload), get / fill_cell / fill_window / search,
// single-chunk and batched.  Kernels walk the serialized big-endian bytes in HBM via k2r_decode.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "k2r_decode.h"
#include "k2r_runtime.h"

using namespace k2r;

struct dcdf_chunk {
    std::vector<InstDesc> descs;  // one per instant, stream order == instant order
    uint32_t instants = 0, rows = 0, cols = 0, n_blocks = 0;
    int32_t encoding = 0;
    uint32_t fbits = 0;
    size_t len = 0;
    DevBuf d_bytes, d_descs;
};

namespace k2r {

struct EventPair {  // destroyed on every exit path
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t create() {
        hipError_t r = hipEventCreate(&e0);
        return r != hipSuccess ? r : hipEventCreate(&e1);
    }
    ~EventPair() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
};

struct ChunkRef {  // device-visible handle of an opened chunk
    const uint8_t* bytes;
    const InstDesc* descs;
    uint32_t instants, rows, cols, fbits;
};

// ---- host-side parser (chunk.rs:247-266, block.rs:99-109, snapshot.rs:62-81, log.rs:68-89,
//      bitmap.rs:142-164, dac.rs:48-63): records byte offsets instead of materialising vectors ----
struct Cursor {
    const uint8_t* p;
    size_t n, pos = 0;
    bool ok = true;
    bool need(size_t k) {
        if (!ok || pos + k > n || pos + k < pos) ok = false;
        return ok;
    }
    uint8_t u8() { return need(1) ? p[pos++] : 0; }
    uint32_t u32() {
        if (!need(4)) return 0;
        uint32_t v = load_be32(p + pos);
        pos += 4;
        return v;
    }
    void skip(size_t k) {
        if (need(k)) pos += k;
    }
};
static void parse_bitmap(Cursor& c, BmDesc& d) {
    d.len = c.u32();
    d.k = c.u32();
    if (d.k == 0) c.ok = false;
    if (!c.ok) return;
    d.idx_off = (uint32_t)c.pos;
    c.skip(4ull * (d.len / 32 / d.k));
    d.words_off = (uint32_t)c.pos;
    c.skip(4ull * ((d.len + 31) / 32));
}
static void parse_dac(Cursor& c, DacDesc& d) {
    std::memset(&d, 0, sizeof(d));
    d.nlev = c.u8();
    if (d.nlev > 8) c.ok = false;
    for (uint32_t l = 0; l < d.nlev && c.ok; l++) {
        parse_bitmap(c, d.bm[l]);
        d.bytes_off[l] = (uint32_t)c.pos;
        c.skip(d.bm[l].len);
    }
}
static void parse_inst(Cursor& c, InstDesc& d, bool is_log, uint32_t snap) {
    std::memset(&d, 0, sizeof(d));
    d.is_log = is_log ? 1u : 0u;
    d.snap = snap;
    d.k = c.u8();
    d.rows = c.u32();
    d.cols = c.u32();
    d.sidelen = c.u32();
    if (d.k < 2 || d.sidelen == 0) c.ok = false;
    parse_bitmap(c, d.T);
    if (is_log) parse_bitmap(c, d.E);
    parse_dac(c, d.mx);
    parse_dac(c, d.mn);
}

// ---- kernels ---------------------------------------------------------------------------------------
struct WinQuery {  // one fill_window / search request against one chunk
    uint32_t chunk;
    uint32_t start, end, top, bottom, left, right;
    uint32_t _pad;
    int64_t lower, upper;
    uint64_t out_off;  // fill_window: first output element
};

// fill_window: one workgroup per query, threads stride over the window's cells; every cell is an
// independent root-to-leaf descent (block.rs:42-47), so writes are coalesced along columns.
__global__ void __launch_bounds__(256)
k_fill_window(const ChunkRef* __restrict__ chunks, const WinQuery* __restrict__ qs, uint32_t nq, void* out,
              int32_t out_dtype, int64_t st, int64_t sr, int64_t sc, int strided_single) {
    for (uint32_t q = blockIdx.x; q < nq; q += gridDim.x) {
        const WinQuery Q = qs[q];
        const ChunkRef C = chunks[Q.chunk];
        const uint32_t wr = Q.bottom - Q.top, wc = Q.right - Q.left, wt = Q.end - Q.start;
        const uint64_t cells = (uint64_t)wt * wr * wc;
        for (uint64_t e = threadIdx.x; e < cells; e += blockDim.x) {
            const uint32_t c = (uint32_t)(e % wc), r = (uint32_t)((e / wc) % wr), t = (uint32_t)(e / ((uint64_t)wc * wr));
            const int64_t v = inst_get(C.bytes, C.descs, Q.start + t, Q.top + r, Q.left + c);
            if (strided_single) store_typed(out, (int64_t)t * st + (int64_t)r * sr + (int64_t)c * sc, out_dtype, v, C.fbits);
            else ((int64_t*)out)[Q.out_off + e] = v;
        }
    }
}

// get / fill_cell: one thread per (query) point
struct PointQuery {
    uint32_t chunk, instant, row, col;
};
__global__ void __launch_bounds__(256)
k_get(const ChunkRef* __restrict__ chunks, const PointQuery* __restrict__ qs, uint32_t nq, int64_t* __restrict__ out) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const PointQuery Q = qs[q];
    const ChunkRef C = chunks[Q.chunk];
    out[q] = inst_get(C.bytes, C.descs, Q.instant, Q.row, Q.col);
}

// search pass 1: one thread per (query, instant) item runs the reference's pruned DFS and marks matches
// in its private window bitmap.
struct SearchItem {
    uint32_t query, instant;  // instant is absolute within the chunk
    uint64_t bits_off;        // u32 words
};
__global__ void __launch_bounds__(64)
k_search_mark(const ChunkRef* __restrict__ chunks, const WinQuery* __restrict__ qs, const SearchItem* __restrict__ items,
              uint32_t n_items, uint32_t* __restrict__ bits, uint32_t* __restrict__ counts) {
    const uint32_t it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= n_items) return;
    const SearchItem I = items[it];
    const WinQuery Q = qs[I.query];
    const ChunkRef C = chunks[Q.chunk];
    WinMark wm{bits + I.bits_off, Q.top, Q.left, Q.right - Q.left};
    inst_search(C.bytes, C.descs, I.instant, Q.top, Q.bottom, Q.left, Q.right, Q.lower, Q.upper, wm);
    const uint32_t nbits = (Q.bottom - Q.top) * (Q.right - Q.left);
    uint32_t cnt = 0;
    for (uint32_t w = 0; w < (nbits + 31) / 32; w++) cnt += popc32(wm.bits[w]);
    counts[it] = cnt;
}
// search pass 2: expand the bitmaps into sorted (instant,row,col) triples
__global__ void __launch_bounds__(64)
k_search_emit(const WinQuery* __restrict__ qs, const SearchItem* __restrict__ items, uint32_t n_items,
              const uint32_t* __restrict__ bits, const uint64_t* __restrict__ offs, uint32_t* __restrict__ out) {
    const uint32_t it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= n_items) return;
    const SearchItem I = items[it];
    const WinQuery Q = qs[I.query];
    const uint32_t wc = Q.right - Q.left, nbits = (Q.bottom - Q.top) * wc;
    const uint32_t* bw = bits + I.bits_off;
    uint32_t* o = out + 3 * offs[it];
    for (uint32_t w = 0; w < (nbits + 31) / 32; w++) {
        uint32_t x = bw[w];
        while (x) {
            const uint32_t p = 32 * w + (uint32_t)__builtin_ctz(x);
            x &= x - 1;
            o[0] = I.instant;
            o[1] = Q.top + p / wc;
            o[2] = Q.left + p % wc;
            o += 3;
        }
    }
}

}  // namespace k2r

// ---- open / close / info -----------------------------------------------------------------------------
extern "C" int dcdf_chunk_open(const uint8_t* bytes, size_t len, dcdf_chunk** h) {
    if (!bytes || !h || len < 6 || len > 0xfffffff0ull) return DCDF_ERR_BAD_ARG;
    if (!Runtime::get().ok) return DCDF_ERR_NO_DEVICE;
    std::unique_ptr<dcdf_chunk> c(new (std::nothrow) dcdf_chunk());
    if (!c) return DCDF_ERR_NOMEM;
    Cursor cur{bytes, len};
    c->encoding = cur.u8();
    if (c->encoding != DCDF_I32 && c->encoding != DCDF_I64 && c->encoding != DCDF_F32 && c->encoding != DCDF_F64)
        return DCDF_ERR_FORMAT;  // mmstruct.rs:49-57
    c->fbits = cur.u8();
    c->n_blocks = cur.u32();
    for (uint32_t b = 0; b < c->n_blocks && cur.ok; b++) {
        const uint32_t n_inst = cur.u8();  // block.rs:100
        if (n_inst == 0) cur.ok = false;
        const uint32_t snap = (uint32_t)c->descs.size();
        for (uint32_t i = 0; i < n_inst && cur.ok; i++) {
            InstDesc d;
            parse_inst(cur, d, i > 0, snap);
            c->descs.push_back(d);
        }
    }
    if (!cur.ok || c->descs.empty() || cur.pos != len) return DCDF_ERR_FORMAT;
    c->instants = (uint32_t)c->descs.size();
    c->rows = c->descs[0].rows;  // chunk.rs:119-123
    c->cols = c->descs[0].cols;
    for (const InstDesc& d : c->descs)
        if (d.rows != c->rows || d.cols != c->cols || d.k != c->descs[0].k || d.sidelen != c->descs[0].sidelen ||
            d.sidelen < std::max(d.rows, d.cols))
            return DCDF_ERR_FORMAT;
    c->len = len;
    K2R_HIP(c->d_bytes.alloc(len + 8));
    K2R_HIP(hipMemcpy(c->d_bytes.p, bytes, len, hipMemcpyHostToDevice));
    K2R_HIP(c->d_descs.alloc(c->descs.size() * sizeof(InstDesc)));
    K2R_HIP(hipMemcpy(c->d_descs.p, c->descs.data(), c->descs.size() * sizeof(InstDesc), hipMemcpyHostToDevice));
    *h = c.release();
    return DCDF_OK;
}
extern "C" void dcdf_chunk_close(dcdf_chunk* h) { delete h; }
extern "C" int dcdf_chunk_info(const dcdf_chunk* h, uint32_t shape[3], int32_t* encoding, uint32_t* fractional_bits,
                               uint32_t* n_blocks) {
    if (!h) return DCDF_ERR_BAD_ARG;
    if (shape) {
        shape[0] = h->instants;
        shape[1] = h->rows;
        shape[2] = h->cols;
    }
    if (encoding) *encoding = h->encoding;
    if (fractional_bits) *fractional_bits = h->fbits;
    if (n_blocks) *n_blocks = h->n_blocks;
    return DCDF_OK;
}

static ChunkRef make_ref(const dcdf_chunk* h) {
    return ChunkRef{h->d_bytes.as<uint8_t>(), h->d_descs.as<InstDesc>(), h->instants, h->rows, h->cols, h->fbits};
}
// geom::Cube::new reorders reversed bounds (geom.rs:83-103)
static dcdf_cube norm_cube(const dcdf_cube& c) {
    dcdf_cube o = c;
    if (o.start > o.end) std::swap(o.start, o.end);
    if (o.top > o.bottom) std::swap(o.top, o.bottom);
    if (o.left > o.right) std::swap(o.left, o.right);
    return o;
}
static bool cube_in(const dcdf_chunk* h, const dcdf_cube& c) {  // mmarray.rs:218-229
    return c.end <= h->instants && c.bottom <= h->rows && c.right <= h->cols;
}

// ---- point queries --------------------------------------------------------------------------------------
static int run_points(const dcdf_chunk* h, const std::vector<PointQuery>& pq, int64_t* out) {
    const ChunkRef ref = make_ref(h);
    DevBuf d_ref, d_q, d_o;
    K2R_HIP(d_ref.alloc(sizeof(ref)));
    K2R_HIP(hipMemcpy(d_ref.p, &ref, sizeof(ref), hipMemcpyHostToDevice));
    K2R_HIP(d_q.alloc(pq.size() * sizeof(PointQuery)));
    K2R_HIP(hipMemcpy(d_q.p, pq.data(), pq.size() * sizeof(PointQuery), hipMemcpyHostToDevice));
    K2R_HIP(d_o.alloc(pq.size() * 8));
    const uint32_t n = (uint32_t)pq.size();
    hipLaunchKernelGGL(k_get, dim3((n + 255) / 256), dim3(256), 0, 0, d_ref.as<ChunkRef>(), d_q.as<PointQuery>(), n,
                       d_o.as<int64_t>());
    K2R_HIP(hipGetLastError());
    K2R_HIP(hipMemcpy(out, d_o.p, pq.size() * 8, hipMemcpyDeviceToHost));
    return DCDF_OK;
}
extern "C" int dcdf_chunk_get(const dcdf_chunk* h, uint32_t instant, uint32_t row, uint32_t col, int64_t* out) {
    if (!h || !out) return DCDF_ERR_BAD_ARG;
    if (instant >= h->instants || row >= h->rows || col >= h->cols) return DCDF_ERR_BOUNDS;
    std::vector<PointQuery> pq{PointQuery{0, instant, row, col}};
    return run_points(h, pq, out);
}
extern "C" int dcdf_chunk_fill_cell(const dcdf_chunk* h, uint32_t start, uint32_t end, uint32_t row, uint32_t col,
                                    int64_t* out) {
    if (!h || !out) return DCDF_ERR_BAD_ARG;
    if (start > end) std::swap(start, end);
    if (end > h->instants || row >= h->rows || col >= h->cols) return DCDF_ERR_BOUNDS;
    if (end == start) return DCDF_OK;
    std::vector<PointQuery> pq;
    for (uint32_t i = start; i < end; i++) pq.push_back(PointQuery{0, i, row, col});
    return run_points(h, pq, out);
}

// ---- windows ----------------------------------------------------------------------------------------------
extern "C" int dcdf_chunk_fill_window(const dcdf_chunk* h, const dcdf_cube* cube, void* out, int32_t out_dtype,
                                      int64_t stride_t, int64_t stride_r, int64_t stride_c) {
    if (!h || !cube || !out) return DCDF_ERR_BAD_ARG;
    if (out_dtype != DCDF_I32 && out_dtype != DCDF_I64 && out_dtype != DCDF_F32 && out_dtype != DCDF_F64)
        return DCDF_ERR_BAD_ARG;
    const dcdf_cube c = norm_cube(*cube);
    if (!cube_in(h, c)) return DCDF_ERR_BOUNDS;
    const uint64_t wt = c.end - c.start, wr = c.bottom - c.top, wc = c.right - c.left;
    if (wt * wr * wc == 0) return DCDF_OK;
    const size_t es = (out_dtype == DCDF_I32 || out_dtype == DCDF_F32) ? 4 : 8;
    // decode into a dense device array of the requested type, then scatter into the caller's strides
    DevBuf d_ref, d_q, d_o;
    const ChunkRef ref = make_ref(h);
    WinQuery q{};
    q.chunk = 0; q.start = c.start; q.end = c.end; q.top = c.top; q.bottom = c.bottom; q.left = c.left; q.right = c.right;
    K2R_HIP(d_ref.alloc(sizeof(ref)));
    K2R_HIP(hipMemcpy(d_ref.p, &ref, sizeof(ref), hipMemcpyHostToDevice));
    K2R_HIP(d_q.alloc(sizeof(q)));
    K2R_HIP(hipMemcpy(d_q.p, &q, sizeof(q), hipMemcpyHostToDevice));
    K2R_HIP(d_o.alloc(wt * wr * wc * es));
    hipLaunchKernelGGL(k_fill_window, dim3(1), dim3(256), 0, 0, d_ref.as<ChunkRef>(), d_q.as<WinQuery>(), 1u, d_o.p,
                       out_dtype, (int64_t)(wr * wc), (int64_t)wc, (int64_t)1, 1);
    K2R_HIP(hipGetLastError());
    std::vector<uint8_t> dense(wt * wr * wc * es);
    K2R_HIP(hipMemcpy(dense.data(), d_o.p, dense.size(), hipMemcpyDeviceToHost));
    const bool contiguous = stride_c == 1 && stride_r == (int64_t)wc && stride_t == (int64_t)(wr * wc);
    if (contiguous) {
        std::memcpy(out, dense.data(), dense.size());
    } else {
        const uint8_t* s = dense.data();
        for (uint64_t t = 0; t < wt; t++)
            for (uint64_t r = 0; r < wr; r++)
                for (uint64_t cc = 0; cc < wc; cc++, s += es)
                    std::memcpy((uint8_t*)out + ((int64_t)t * stride_t + (int64_t)r * stride_r + (int64_t)cc * stride_c) * (int64_t)es, s, es);
    }
    return DCDF_OK;
}

// ---- batched machinery shared by search (single + batch) and fill_window batch ------------------------------
static int upload_refs(dcdf_chunk* const* chunks, const std::vector<uint32_t>& uniq_of_query, size_t nuniq,
                       const std::vector<const dcdf_chunk*>& uniq, DevBuf& d_refs) {
    std::vector<ChunkRef> refs(nuniq);
    for (size_t i = 0; i < nuniq; i++) refs[i] = make_ref(uniq[i]);
    K2R_HIP(d_refs.alloc(nuniq * sizeof(ChunkRef)));
    K2R_HIP(hipMemcpy(d_refs.p, refs.data(), nuniq * sizeof(ChunkRef), hipMemcpyHostToDevice));
    (void)chunks;
    (void)uniq_of_query;
    return DCDF_OK;
}
static void dedup_chunks(dcdf_chunk* const* chunks, size_t nq, std::vector<uint32_t>& idx,
                         std::vector<const dcdf_chunk*>& uniq) {
    // queries usually hit few distinct chunks repeatedly; map pointer -> dense index (sorted unique)
    std::vector<const dcdf_chunk*> sorted(chunks, chunks + nq);
    std::sort(sorted.begin(), sorted.end());
    sorted.erase(std::unique(sorted.begin(), sorted.end()), sorted.end());
    uniq = sorted;
    idx.resize(nq);
    for (size_t q = 0; q < nq; q++)
        idx[q] = (uint32_t)(std::lower_bound(uniq.begin(), uniq.end(), (const dcdf_chunk*)chunks[q]) - uniq.begin());
}

static int search_impl(dcdf_chunk* const* chunks, const dcdf_cube* cubes, const int64_t* lower, const int64_t* upper,
                       size_t nq, uint32_t* out, size_t cap, uint64_t* counts, uint64_t* offsets, size_t* total_out,
                       float* kernel_ms) {
    std::vector<uint32_t> cidx;
    std::vector<const dcdf_chunk*> uniq;
    dedup_chunks(chunks, nq, cidx, uniq);
    std::vector<WinQuery> qs(nq);
    std::vector<SearchItem> items;
    uint64_t bits_words = 0;
    for (size_t q = 0; q < nq; q++) {
        if (!chunks[q]) return DCDF_ERR_BAD_ARG;
        const dcdf_cube c = norm_cube(cubes[q]);
        if (!cube_in(chunks[q], c)) return DCDF_ERR_BOUNDS;
        WinQuery& Q = qs[q];
        Q = WinQuery{};
        Q.chunk = cidx[q];
        Q.start = c.start; Q.end = c.end; Q.top = c.top; Q.bottom = c.bottom; Q.left = c.left; Q.right = c.right;
        Q.lower = std::min(lower[q], upper[q]);  // helpers.rs:7-16 via chunk.rs:214
        Q.upper = std::max(lower[q], upper[q]);
        const uint64_t cells = (uint64_t)(c.bottom - c.top) * (c.right - c.left);
        if (cells == 0) continue;
        for (uint32_t i = c.start; i < c.end; i++) {
            items.push_back(SearchItem{(uint32_t)q, i, bits_words});
            bits_words += (cells + 31) / 32;
        }
    }
    for (size_t q = 0; q < nq; q++) counts[q] = 0;
    float ms_total = 0.f;
    std::vector<uint32_t> item_counts(items.size());
    DevBuf d_refs, d_qs, d_items, d_bits, d_counts, d_offs, d_out;
    if (!items.empty()) {
        int rc = upload_refs(chunks, cidx, uniq.size(), uniq, d_refs);
        if (rc != DCDF_OK) return rc;
        K2R_HIP(d_qs.alloc(nq * sizeof(WinQuery)));
        K2R_HIP(hipMemcpy(d_qs.p, qs.data(), nq * sizeof(WinQuery), hipMemcpyHostToDevice));
        K2R_HIP(d_items.alloc(items.size() * sizeof(SearchItem)));
        K2R_HIP(hipMemcpy(d_items.p, items.data(), items.size() * sizeof(SearchItem), hipMemcpyHostToDevice));
        K2R_HIP(d_bits.alloc(bits_words * 4));
        K2R_HIP(hipMemset(d_bits.p, 0, bits_words * 4));
        K2R_HIP(d_counts.alloc(items.size() * 4));
        EventPair ev;
        K2R_HIP(ev.create());
        const hipEvent_t e0 = ev.e0, e1 = ev.e1;
        const uint32_t ni = (uint32_t)items.size();
        K2R_HIP(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_search_mark, dim3((ni + 63) / 64), dim3(64), 0, 0, d_refs.as<ChunkRef>(), d_qs.as<WinQuery>(),
                           d_items.as<SearchItem>(), ni, d_bits.as<uint32_t>(), d_counts.as<uint32_t>());
        K2R_HIP(hipEventRecord(e1, 0));
        K2R_HIP(hipGetLastError());
        K2R_HIP(hipMemcpy(item_counts.data(), d_counts.p, items.size() * 4, hipMemcpyDeviceToHost));
        float ms = 0.f;
        K2R_HIP(hipEventElapsedTime(&ms, e0, e1));
        ms_total += ms;
        std::vector<uint64_t> item_offs(items.size());
        uint64_t run = 0;
        for (size_t i = 0; i < items.size(); i++) {
            item_offs[i] = run;
            run += item_counts[i];
            counts[items[i].query] += item_counts[i];
        }
        *total_out = run;
        uint64_t acc = 0;
        for (size_t q = 0; q < nq; q++) {
            offsets[q] = acc;
            acc += counts[q];
        }
        if (run > cap) return DCDF_ERR_CAPACITY;
        if (run > 0) {
            K2R_HIP(d_offs.alloc(items.size() * 8));
            K2R_HIP(hipMemcpy(d_offs.p, item_offs.data(), items.size() * 8, hipMemcpyHostToDevice));
            K2R_HIP(d_out.alloc(run * 12));
            K2R_HIP(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_search_emit, dim3((ni + 63) / 64), dim3(64), 0, 0, d_qs.as<WinQuery>(),
                               d_items.as<SearchItem>(), ni, d_bits.as<uint32_t>(), d_offs.as<uint64_t>(),
                               d_out.as<uint32_t>());
            K2R_HIP(hipEventRecord(e1, 0));
            K2R_HIP(hipGetLastError());
            K2R_HIP(hipMemcpy(out, d_out.p, run * 12, hipMemcpyDeviceToHost));
            K2R_HIP(hipEventElapsedTime(&ms, e0, e1));
            ms_total += ms;
        }
    } else {
        *total_out = 0;
        for (size_t q = 0; q < nq; q++) offsets[q] = 0;
    }
    if (kernel_ms) *kernel_ms = ms_total;
    return DCDF_OK;
}

extern "C" int dcdf_chunk_search(const dcdf_chunk* h, const dcdf_cube* cube, int64_t lower, int64_t upper,
                                 uint32_t* out, size_t cap, size_t* n) {
    if (!h || !cube || !n || (!out && cap)) return DCDF_ERR_BAD_ARG;
    if (!Runtime::get().ok) return DCDF_ERR_NO_DEVICE;
    dcdf_chunk* hp = const_cast<dcdf_chunk*>(h);
    uint64_t cnt = 0, off = 0;
    size_t total = 0;
    const int rc = search_impl(&hp, cube, &lower, &upper, 1, out, cap, &cnt, &off, &total, nullptr);
    *n = total;
    return rc;
}

extern "C" int dcdf_query_search_batch(dcdf_chunk* const* chunks, const dcdf_cube* cubes, const int64_t* lower,
                                       const int64_t* upper, size_t nq, uint32_t* out, size_t cap, uint64_t* counts,
                                       uint64_t* offsets, float* kernel_ms) {
    if (!chunks || !cubes || !lower || !upper || !counts || !offsets || nq == 0 || (!out && cap)) return DCDF_ERR_BAD_ARG;
    if (!Runtime::get().ok) return DCDF_ERR_NO_DEVICE;
    size_t total = 0;
    return search_impl(chunks, cubes, lower, upper, nq, out, cap, counts, offsets, &total, kernel_ms);
}

extern "C" int dcdf_query_fill_window_batch(dcdf_chunk* const* chunks, const dcdf_cube* cubes, size_t nq, int64_t* out,
                                            const uint64_t* out_offset, float* kernel_ms) {
    if (!chunks || !cubes || !out || !out_offset || nq == 0) return DCDF_ERR_BAD_ARG;
    if (!Runtime::get().ok) return DCDF_ERR_NO_DEVICE;
    std::vector<uint32_t> cidx;
    std::vector<const dcdf_chunk*> uniq;
    dedup_chunks(chunks, nq, cidx, uniq);
    std::vector<WinQuery> qs(nq);
    // windows are decoded into a DENSE device buffer (internal offsets) and only the windows themselves are written to
    // the caller's array: query q touches out[out_offset[q] .. + its cell count) and nothing else
    uint64_t total = 0;
    bool dense = true;
    for (size_t q = 0; q < nq; q++) {
        if (!chunks[q]) return DCDF_ERR_BAD_ARG;
        const dcdf_cube c = norm_cube(cubes[q]);
        if (!cube_in(chunks[q], c)) return DCDF_ERR_BOUNDS;
        WinQuery& Q = qs[q];
        Q = WinQuery{};
        Q.chunk = cidx[q];
        Q.start = c.start; Q.end = c.end; Q.top = c.top; Q.bottom = c.bottom; Q.left = c.left; Q.right = c.right;
        Q.out_off = total;
        dense = dense && out_offset[q] == total + out_offset[0];
        total += (uint64_t)(c.end - c.start) * (c.bottom - c.top) * (c.right - c.left);
    }
    if (total == 0) return DCDF_OK;
    DevBuf d_refs, d_qs, d_o;
    int rc = upload_refs(chunks, cidx, uniq.size(), uniq, d_refs);
    if (rc != DCDF_OK) return rc;
    K2R_HIP(d_qs.alloc(nq * sizeof(WinQuery)));
    K2R_HIP(hipMemcpy(d_qs.p, qs.data(), nq * sizeof(WinQuery), hipMemcpyHostToDevice));
    K2R_HIP(d_o.alloc(total * 8));
    EventPair ev;
    K2R_HIP(ev.create());
    const uint32_t grid = (uint32_t)std::min<size_t>(nq, 1u << 20);
    K2R_HIP(hipEventRecord(ev.e0, 0));
    hipLaunchKernelGGL(k_fill_window, dim3(grid), dim3(256), 0, 0, d_refs.as<ChunkRef>(), d_qs.as<WinQuery>(),
                       (uint32_t)nq, d_o.p, (int32_t)DCDF_I64, (int64_t)0, (int64_t)0, (int64_t)0, 0);
    K2R_HIP(hipEventRecord(ev.e1, 0));
    K2R_HIP(hipGetLastError());
    if (dense) {  // the usual case: windows back to back in query order -> one copy straight into the caller's array
        K2R_HIP(hipMemcpy(out + out_offset[0], d_o.p, total * 8, hipMemcpyDeviceToHost));
    } else {
        std::vector<int64_t> tmp(total);
        K2R_HIP(hipMemcpy(tmp.data(), d_o.p, total * 8, hipMemcpyDeviceToHost));
        for (size_t q = 0; q < nq; q++) {
            const uint64_t cells = (uint64_t)(qs[q].end - qs[q].start) * (qs[q].bottom - qs[q].top) * (qs[q].right - qs[q].left);
            if (cells) std::memcpy(out + out_offset[q], tmp.data() + qs[q].out_off, cells * 8);
        }
    }
    float ms = 0.f;
    K2R_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
    if (kernel_ms) *kernel_ms = ms;
    return DCDF_OK;
}
