// k2r_superchunk.hip -- Superchunk assembly around the chunk encoder: the direct caller of the hot path.
//
// Replaces `Superchunk::build(resolver, buffer, shape, levels, k)` (superchunk.rs:88-270) up to, but not including, the
// store: it returns every object the reference would hand to `resolver.save` -- the framed sub-chunks (resolver.rs:126-138,
// mmstruct.rs:215-218), nested Superchunk nodes, the `Links` node (links.rs:65-76) -- each with its CID (testing.rs:172-183:
// CIDv1, codec 0x12, sha2-256), de-duplicated exactly as `external_references` does (superchunk.rs:199-235), and last the
// Superchunk node itself (superchunk.rs:672-706), byte for byte.
//
// Where the work happens:
//   * per-tile per-instant (min, max) -- MMBuffer3::min_max, mmbuffer.rs:366-499, the float variant's NaN behaviour
//     included -- one workgroup per (tile, instant) on the device: `k_tile_minmax`;
//   * uniform-tile elision, the reference table and the recursion into nested superchunks: host control flow on those
//     numbers (superchunk.rs:127-181,207-236);
//   * per-tile fractional bits (superchunk.rs:167): dcdf_suggest_fraction (device reductions, k2r_suggest.hip);
//   * Chunk::build of every non-elided bottom tile: ONE batched launch of the encoder (dcdf_chunk_build_batch);
//   * the instant-major min / max Dacs (superchunk.rs:190-198,246-247): `k_dac_pack` (k2r_generic.hip);
//   * SHA-256 of every stored object: on the host (x86 SHA extensions, k2r_sha256_host.h) by the download's worker threads as each
//     object lands in host memory -- a hash is one serial chain per object, 139 ms on one GPU lane for the 1.4 MB sub-chunks of a
//     4096^2 level against 15 ms overlapped with the copy; the device kernels (`k_object_sha256`, `k_sha256_buffers`, k2r_cid.hip)
//     remain for objects that stay in HBM (dcdf_encoder_object_sha256).
#include <hip/hip_runtime.h>

#include <atomic>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "k2r_encode.h"
#include "k2r_runtime.h"
#include "k2r_sha256_host.h"

namespace k2r {
hipError_t launch_dac_pack(const int64_t* values, uint64_t n, uint8_t* out, uint8_t* tmp, uint64_t* out_len, hipStream_t stream);
hipError_t launch_sha256_buffers(const uint8_t* data, const uint64_t* offs, const uint64_t* lens, uint32_t n, uint8_t* digests, hipStream_t stream);

struct MinMaxTile {  // one tile of the superchunk's grid (device view)
    const void* base;
    int64_t st, sr, sc;
    uint32_t rows, cols;
};

// (min, max) of one instant of one tile as STORED values (mmbuffer.rs:366-395).  Floats follow min_max_float
// (mmbuffer.rs:466-499): leading NaNs are skipped, a NaN met later makes the minimum NaN for good; the result goes through
// to_fixed with the buffer's own fractional bits (NaN -> 0).
__global__ void __launch_bounds__(256) k_tile_minmax(const MinMaxTile* __restrict__ tiles, uint32_t instants, int32_t dtype, uint32_t fbits,
                                                     uint32_t round, int64_t* __restrict__ out, int32_t* __restrict__ status) {
    const uint32_t tile = blockIdx.x / instants, inst = blockIdx.x % instants;
    const MinMaxTile T = tiles[tile];
    const uint64_t n = (uint64_t)T.rows * T.cols;
    __shared__ int64_t smn[256], smx[256];
    __shared__ unsigned long long sfirst[256], slastnan[256];
    const int tid = (int)threadIdx.x;
    const bool is_float = dtype == ENC_F32 || dtype == ENC_F64;
    int64_t mn = INT64_MAX, mx = INT64_MIN;
    double fmn = INFINITY, fmx = -INFINITY;
    unsigned long long first = ~0ull, lastnan = 0;  // first non-NaN position; 1 + last NaN position
    for (uint64_t i = tid; i < n; i += 256) {
        const int64_t off = (int64_t)inst * T.st + (int64_t)(i / T.cols) * T.sr + (int64_t)(i % T.cols) * T.sc;
        if (!is_float) {
            const int64_t v = dtype == ENC_I32 ? (int64_t)((const int32_t*)T.base)[off] : ((const int64_t*)T.base)[off];
            mn = v < mn ? v : mn;
            mx = v > mx ? v : mx;
        } else {
            const double v = dtype == ENC_F32 ? (double)((const float*)T.base)[off] : ((const double*)T.base)[off];
            if (v != v) {
                lastnan = i + 1 > lastnan ? i + 1 : lastnan;
            } else {
                first = i < first ? i : first;
                fmn = v < fmn ? v : fmn;
                fmx = v > fmx ? v : fmx;
            }
        }
    }
    if (is_float) {
        mn = (int64_t)__double_as_longlong(fmn);
        mx = (int64_t)__double_as_longlong(fmx);
    }
    smn[tid] = mn; smx[tid] = mx; sfirst[tid] = first; slastnan[tid] = lastnan;
    __syncthreads();
    if (tid == 0) {
        int32_t err = 0;
        if (!is_float) {
            for (int i = 1; i < 256; i++) {
                mn = smn[i] < mn ? smn[i] : mn;
                mx = smx[i] > mx ? smx[i] : mx;
            }
        } else {
            for (int i = 1; i < 256; i++) {
                const double a = __longlong_as_double(smn[i]), b = __longlong_as_double(smx[i]);
                fmn = a < fmn ? a : fmn;
                fmx = b > fmx ? b : fmx;
                first = sfirst[i] < first ? sfirst[i] : first;
                lastnan = slastnan[i] > lastnan ? slastnan[i] : lastnan;
            }
            const bool all_nan = first == ~0ull;
            const bool nan_after = !all_nan && lastnan > first + 1;  // a NaN behind the first number (lastnan = its position + 1)
            const double qn = __longlong_as_double(0x7ff8000000000000ll);
            const double vmin = (all_nan || nan_after) ? qn : fmn, vmax = all_nan ? qn : fmx;
            if (dtype == ENC_F32) {
                mn = to_fixed_dev<float>((float)vmin, fbits, round != 0, err);
                mx = to_fixed_dev<float>((float)vmax, fbits, round != 0, err);
            } else {
                mn = to_fixed_dev<double>(vmin, fbits, round != 0, err);
                mx = to_fixed_dev<double>(vmax, fbits, round != 0, err);
            }
        }
        out[2ull * blockIdx.x] = mn;
        out[2ull * blockIdx.x + 1] = mx;
        if (err) atomicMin(status, err);
    }
}

}  // namespace k2r

using namespace k2r;

namespace {

struct Obj {
    std::string bytes;  // the stored object (header included)
};
struct Level {
    std::string node;  // header + NODE_SUPERCHUNK + body
    uint64_t size_self = 0, size = 0;
    uint32_t elided = 0, external = 0, snapshots = 0, logs = 0;
};
struct Blob {  // a stored object in malloc'ed memory (handed to the caller as it is: no copy at the end)
    uint8_t* p = nullptr;
    size_t n = 0;
};
Blob blob_of(const std::string& s) {
    Blob b;
    b.p = (uint8_t*)std::malloc(s.size() ? s.size() : 1);
    b.n = s.size();
    if (b.p) std::memcpy(b.p, s.data(), s.size());
    return b;
}
struct Ctx {
    int k;
    std::vector<Blob> objects;                   // in save order, de-duplicated
    std::map<std::string, size_t> by_cid;        // cid -> index in objects
    std::vector<std::string> cids;
    bool nomem = false;
    ~Ctx() {
        for (Blob& b : objects) std::free(b.p);  // (released objects are nulled)
    }
};
std::string cid_of(const uint8_t* p, size_t n) {  // testing.rs:172-183: CIDv1, codec 0x12, sha2-256
    uint8_t d[32];
    sha256_host(p, n, d);
    std::string c("\x01\x12\x12\x20", 4);
    c.append((const char*)d, 32);
    return c;
}

void put_u32(std::string& s, uint32_t v) {
    const char b[4] = {(char)(v >> 24), (char)(v >> 16), (char)(v >> 8), (char)v};
    s.append(b, 4);
}
std::string header(uint8_t node_type) {  // resolver.rs:130-133
    std::string s;
    s.push_back((char)0xDC);
    s.push_back((char)0xE0);
    put_u32(s, 1);
    s.push_back((char)node_type);
    return s;
}
size_t esize(int dtype) { return (dtype == DCDF_I32 || dtype == DCDF_F32) ? 4 : 8; }

// CIDs of a batch of small host-resident objects (nested Superchunk nodes, Links)
int hash_objects(const std::vector<const std::string*>& objs, std::vector<std::string>& cids) {
    cids.clear();
    for (auto* o : objs) cids.push_back(cid_of((const uint8_t*)o->data(), o->size()));
    return DCDF_OK;
}
// resolver.save: remember the object under its CID (content addressing makes identical objects one object)
void save(Ctx& cx, const std::string& cid, Blob obj) {  // (takes ownership)
    if (!obj.p) cx.nomem = true;
    if (cx.by_cid.count(cid) || !obj.p) {
        std::free(obj.p);
        return;
    }
    cx.by_cid[cid] = cx.objects.size();
    cx.objects.push_back(obj);
    cx.cids.push_back(cid);
}
// Dac::from(values).write_to on the device
int dac_bytes(const std::vector<int64_t>& v, std::string& out) {
    DevBuf d_v, d_out, d_tmp, d_len;
    const size_t n = v.size();
    K2R_HIP(d_v.alloc(std::max<size_t>(n, 1) * 8));
    K2R_HIP(d_out.alloc(1 + 8 * (n + 8 + (n / 128 + 1) * 4 + ((n + 31) / 32) * 4) + 64));
    K2R_HIP(d_tmp.alloc(n + 16));
    K2R_HIP(d_len.alloc(8));
    if (n) K2R_HIP(hipMemcpy(d_v.p, v.data(), n * 8, hipMemcpyHostToDevice));
    K2R_HIP(launch_dac_pack(d_v.as<int64_t>(), n, d_out.as<uint8_t>(), d_tmp.as<uint8_t>(), d_len.as<uint64_t>(), 0));
    uint64_t len = 0;
    K2R_HIP(hipMemcpy(&len, d_len.p, 8, hipMemcpyDeviceToHost));
    out.resize(len);
    K2R_HIP(hipMemcpy(&out[0], d_out.p, len, hipMemcpyDeviceToHost));
    return DCDF_OK;
}
}  // namespace
namespace k2r {
int suggest_fraction_batch(const dcdf_tile_desc* tiles, size_t n, int32_t* out_round, int32_t* out_bits, int32_t* status);
}
namespace {
uint32_t levels_needed(uint64_t side, int k) { return ref_levels(side, (uint32_t)k); }  // superchunk.rs:98-101 (f64 formula)

// One Superchunk::build over the DEVICE view `buf` (superchunk.rs:88-270).
struct StepTimer {  // K2R_SC_TIMING=1: wall time of the assembly's steps on stderr (diagnostics)
    const bool on = std::getenv("K2R_SC_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "k2r-sc %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};
// K2R_SC_TIMING: a timeline of the banded assembly on stderr (ms since the first mark of the process)
void trace(const char* what, long a = -1) {
    static const bool on = std::getenv("K2R_SC_TIMING") != nullptr;
    if (!on) return;
    static const auto t0 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "k2r-sc-trace %9.2f ms  %s %ld\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what, a);
}

// A host view on its way to HBM, band of rows after band of rows, on a thread of its own: the assembly below starts on the tiles of
// a band as soon as its rows (of every instant) have landed, and the bands behind it keep the host-to-device link busy meanwhile.
// Dense rows and row-pitched views are copied straight from the caller's array (the runtime page-locks pageable memory on the
// way and reaches the link's rate; staging them through page-locked buffers with host threads was measured slower: the copies
// take the cores the download's hashing needs); other strides are gathered row by row into a staging vector first.
struct Upload {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv;
    uint32_t rows_ready = 0;
    std::vector<uint32_t> edges;  // band b = rows [edges[b], edges[b + 1]) of every instant; empty or two entries: one band
    int rc = DCDF_OK;
    struct Piece {
        uint32_t t, r0, r1, band;
    };
    std::vector<Piece> pieces;            // band after band, instant after instant
    std::vector<uint32_t> band_end;       // rows uploaded once bands 0..b are complete
    std::vector<uint32_t> band_left;      // pieces of band b still to land (under mu)
    uint32_t bands_done = 0;
    std::atomic<size_t> next{0};
    int wait_rows(uint32_t upto) {  // rows [0, upto) of every instant are in HBM (or the upload failed)
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return rows_ready >= upto || rc != DCDF_OK; });
        return rc;
    }
    void landed(uint32_t band) {
        {
            std::lock_guard<std::mutex> lk(mu);
            band_left[band]--;
            while (bands_done < band_left.size() && band_left[bands_done] == 0) {
                rows_ready = band_end[bands_done++];
                trace("band uploaded", (long)bands_done - 1);
            }
        }
        cv.notify_all();
    }
    void fail(int code) {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (rc == DCDF_OK) rc = code;
        }
        cv.notify_all();
    }
    void* registered = nullptr;
    int start(const dcdf_tile_desc& src, uint8_t* dst) {
        const uint64_t T = src.instants, R = src.rows;
        // Page-lock the caller's array for the duration of the call when the view is dense enough for that to pay: the pieces below
        // then go out as plain asynchronous DMA (a piece of pageable memory is locked and unlocked by the runtime per copy: 32 GB/s
        // measured against 56 for one large copy).  Not possible (read-only mappings, already registered, ...): pageable copies.
        if (src.stride_c == 1 && src.stride_r >= (int64_t)src.cols && src.stride_t >= src.stride_r * (int64_t)R &&
            std::getenv("K2R_NO_HOST_REGISTER") == nullptr) {
            const size_t es = esize(src.dtype);
            const size_t span = (size_t)(((int64_t)(T - 1) * src.stride_t + (int64_t)(R - 1) * src.stride_r + (int64_t)src.cols) * (int64_t)es);
            const size_t dense = (size_t)(T * R * src.cols) * es;
            if (span <= dense + dense / 4 && dense >= (64u << 20)) {
                trace("upload: registering the host view, MB", (long)(span >> 20));
                if (hipHostRegister((void*)src.base, span, hipHostRegisterDefault) == hipSuccess) registered = (void*)src.base;
                else (void)hipGetLastError();
                trace("upload: registered", registered ? 1 : 0);
            }
        }
        if (edges.size() < 2) edges = {0u, (uint32_t)R};
        for (size_t band = 0; band + 1 < edges.size(); band++) {
            const uint64_t b0 = edges[band], b1 = edges[band + 1];
            for (uint64_t t = 0; t < T; t++) pieces.push_back(Piece{(uint32_t)t, (uint32_t)b0, (uint32_t)b1, (uint32_t)band});
            band_end.push_back((uint32_t)b1);
            band_left.push_back((uint32_t)T);
        }
        // (one thread issues the pieces; more of them were measured no faster, page-locked or not)
        size_t nt = 1;
        if (const char* e = std::getenv("K2R_UP_THREADS")) nt = (size_t)std::max(1, std::min(16, std::atoi(e)));
        nt = std::min(nt, pieces.size());
        const int device = Runtime::get().device;
        for (size_t j = 0; j < nt; j++)
            th.emplace_back([this, src, dst, device] {
                const size_t es = esize(src.dtype);
                const uint64_t R = src.rows, Cc = src.cols;
                const int64_t st = src.stride_t, sr = src.stride_r, sc = src.stride_c;
                const uint8_t* const base = (const uint8_t*)src.base;
                hipStream_t stream = nullptr;
                hipError_t he = hipSetDevice(device);
                if (he == hipSuccess) he = StreamPool::get().take(true, &stream);
                std::vector<uint8_t> gather;
                // at most two pieces of this thread in flight: the session tables and the (min, max) results of the bands already
                // here travel on the same copy engines, and behind a whole band of queued pieces they waited 5 ms per small copy
                hipEvent_t ev[2] = {nullptr, nullptr};
                int64_t pending[2] = {-1, -1};
                for (int i = 0; i < 2 && he == hipSuccess; i++) he = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
                auto retire = [&](int i) {
                    if (pending[i] < 0) return;
                    if (he == hipSuccess) he = hipEventSynchronize(ev[i]);
                    if (he == hipSuccess) landed(pieces[(size_t)pending[i]].band);
                    pending[i] = -1;
                };
                for (size_t k = 0; he == hipSuccess; k++) {
                    const size_t p = next.fetch_add(1);
                    if (p >= pieces.size()) break;
                    const int slot = (int)(k & 1);
                    retire(slot);
                    if (he != hipSuccess) break;
                    const Piece& pc = pieces[p];
                    const uint64_t nr = pc.r1 - pc.r0;
                    uint8_t* const d = dst + (size_t)(((uint64_t)pc.t * R + pc.r0) * Cc) * es;
                    const uint8_t* const from = base + ((int64_t)pc.t * st + (int64_t)pc.r0 * sr) * (int64_t)es;
                    if (sc == 1 && sr == (int64_t)Cc) {
                        he = hipMemcpyAsync(d, from, (size_t)(nr * Cc) * es, hipMemcpyHostToDevice, stream);  // dense rows
                    } else if (sc == 1 && sr > (int64_t)Cc) {
                        // a row pitch (a cropped window of a larger array): a 2-D copy does the gathering
                        he = hipMemcpy2DAsync(d, (size_t)Cc * es, from, (size_t)sr * es, (size_t)Cc * es, (size_t)nr, hipMemcpyHostToDevice, stream);
                    } else {
                        // other strides (transposed / reversed / stepped views): gathered row by row
                        gather.resize((size_t)(nr * Cc) * es);
                        uint8_t* g = gather.data();
                        for (uint64_t r = 0; r < nr; r++) {
                            const uint8_t* row = from + (int64_t)r * sr * (int64_t)es;
                            if (es == 4)
                                for (uint64_t c = 0; c < Cc; c++, g += 4) *(uint32_t*)g = *(const uint32_t*)(row + (int64_t)c * sc * 4);
                            else
                                for (uint64_t c = 0; c < Cc; c++, g += 8) *(uint64_t*)g = *(const uint64_t*)(row + (int64_t)c * sc * 8);
                        }
                        he = hipMemcpyAsync(d, gather.data(), gather.size(), hipMemcpyHostToDevice, stream);
                        if (he == hipSuccess) he = hipStreamSynchronize(stream);  // (the staging is reused)
                    }
                    if (he == hipSuccess) he = hipEventRecord(ev[slot], stream);
                    if (he == hipSuccess) pending[slot] = (int64_t)p;
                }
                retire(0);
                retire(1);
                StreamPool::get().give(true, stream);
                for (int i = 0; i < 2; i++)
                    if (ev[i]) (void)hipEventDestroy(ev[i]);
                if (he != hipSuccess) fail(k2r::map_hip_error(he));
            });
        return DCDF_OK;
    }
    void join() {
        for (auto& t : th)
            if (t.joinable()) t.join();
        if (registered) {
            (void)hipHostUnregister(registered);
            registered = nullptr;
            trace("upload: unregistered");
        }
    }
    ~Upload() { join(); }
};

int build_level(Ctx& cx, const dcdf_tile_desc& buf, const uint32_t* levels, size_t n_levels, Level* out, Upload* up = nullptr) {
    StepTimer tm;
    const int k = cx.k;
    const uint32_t instants = buf.instants, rows = buf.rows, cols = buf.cols;
    const uint32_t total_levels = levels_needed(std::max(rows, cols), k);
    uint32_t user = 0;
    for (size_t i = 0; i < n_levels; i++) user += levels[i];
    if (n_levels < 2 || user != total_levels) return DCDF_ERR_BAD_ARG;  // superchunk.rs:104-110 panics
    uint64_t sidelen = 1;
    for (uint32_t i = 0; i < total_levels; i++) sidelen *= (uint64_t)k;
    const uint32_t* sublevels = levels + 1;
    const size_t n_sub = n_levels - 1;
    const bool at_bottom = n_sub == 1;
    uint64_t subsidelen = 1;
    for (uint32_t i = 0; i < levels[0]; i++) subsidelen *= (uint64_t)k;
    const uint64_t chunks_sidelen = sidelen / subsidelen;
    const size_t n_tiles = (size_t)(subsidelen * subsidelen);
    const size_t es = esize(buf.dtype);
    // ---- the grid (superchunk.rs:127-142) and its per-instant (min, max) on the device ----
    std::vector<dcdf_tile_desc> tiles(n_tiles);
    std::vector<char> inside(n_tiles, 0);
    for (uint64_t row = 0; row < subsidelen; row++)
        for (uint64_t col = 0; col < subsidelen; col++) {
            const size_t i = (size_t)(row * subsidelen + col);
            const uint64_t top = row * chunks_sidelen, left = col * chunks_sidelen;
            if (top >= rows || left >= cols) continue;
            const uint64_t bottom = std::min<uint64_t>(top + chunks_sidelen, rows), right = std::min<uint64_t>(left + chunks_sidelen, cols);
            dcdf_tile_desc t = buf;
            t.base = (const uint8_t*)buf.base + ((int64_t)top * buf.stride_r + (int64_t)left * buf.stride_c) * (int64_t)es;
            t.rows = (uint32_t)(bottom - top);
            t.cols = (uint32_t)(right - left);
            tiles[i] = t;
            inside[i] = 1;
        }
    std::vector<int64_t> mm(2ull * n_tiles * instants, 0);  // [tile][instant][2]
    std::vector<char> elided(n_tiles, 1);
    struct Sub {
        size_t tile;
        bool chunk;
        std::string obj;   // a nested Superchunk node
        Blob blob;         // a framed sub-chunk, filled by the download
        uint64_t size;
        uint32_t snapshots, logs;
    };
    struct SubsGuard {  // (blobs not handed over to the context yet)
        std::vector<Sub>& v;
        ~SubsGuard() {
            for (Sub& s : v) std::free(s.blob.p);
        }
    };
    std::vector<Sub> subs;
    subs.reserve(n_tiles);  // (the download thread fills entries while later bands append theirs: no reallocation)
    std::vector<std::string> chunk_cid(n_tiles);  // CIDs of the sub-chunk objects, by index into subs
    // ---- the sessions' results come to the host on a thread of their own when the level is assembled in bands ----
    struct Job {
        dcdf_encoder* e = nullptr;
        std::vector<size_t> chunk_sub;  // session tile -> index into subs
    };
    const std::string hd = header(2) + std::string(1, (char)4);  // NODE_MMSTRUCT3, NODE_SUBCHUNK (mmstruct.rs:215-218)
    // the bytes come to the host through the pinned ring and land, framed, in their final objects: header + tag written, then the
    // chunk bytes copied in by the download's worker threads, which hash the object there and then (its CID: SHA-256 on the host
    // while the next slot is in flight -- k2r_sha256_host.h)
    auto fetch = [&](Job& j) -> int {
        int rc = k2r::encoder_download(
            j.e,
            [&](size_t q, uint64_t len) -> uint8_t* {
                Blob& o = subs[j.chunk_sub[q]].blob;
                o.n = hd.size() + len;
                o.p = (uint8_t*)std::malloc(o.n);
                if (!o.p) return nullptr;
                std::memcpy(o.p, hd.data(), hd.size());
                return o.p + hd.size();
            },
            [&](size_t q) {
                const Blob& o = subs[j.chunk_sub[q]].blob;
                chunk_cid[j.chunk_sub[q]] = cid_of(o.p, o.n);
            });
        for (size_t q = 0; rc == DCDF_OK && q < j.chunk_sub.size(); q++) {
            uint64_t len = 0;
            uint32_t ns = 0, nl = 0;
            int32_t st = 0;
            (void)dcdf_encoder_result(j.e, q, &st, &len, &ns, &nl, nullptr);
            Sub& sb = subs[j.chunk_sub[q]];
            sb.size = len + 1;
            sb.snapshots = ns;
            sb.logs = nl;
        }
        dcdf_encoder_destroy(j.e);
        j.e = nullptr;
        return rc;
    };
    struct Fetcher {  // jobs in, first failure out
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::deque<Job*> q;
        bool closing = false;
        int rc = DCDF_OK;
        std::vector<std::unique_ptr<Job>> jobs;
        int close() {  // after the last push: wait for the queue to drain
            {
                std::lock_guard<std::mutex> lk(mu);
                closing = true;
            }
            cv.notify_all();
            if (th.joinable()) th.join();
            return rc;
        }
        ~Fetcher() {
            (void)close();
            for (auto& j : jobs)
                if (j->e) dcdf_encoder_destroy(j->e);
        }
    };
    SubsGuard subs_guard{subs};  // (declared before the fetcher: its thread is gone before the blobs are)
    Fetcher fx;
    auto can_elide_tile = [&](size_t i) {
        for (uint32_t t = 0; t < instants; t++)
            if (mm[2ull * (i * instants + t)] != mm[2ull * (i * instants + t) + 1]) return false;
        return true;
    };
    // ---- bands of tile rows: one when the view is in HBM already (or below the top level), several behind an upload ----
    std::vector<uint64_t> tr_edges{0, subsidelen};  // in tile rows
    if (up && at_bottom && up->edges.size() > 2) {
        tr_edges.clear();
        for (size_t b = 0; b + 1 < up->edges.size(); b++) tr_edges.push_back(up->edges[b] / chunks_sidelen);
        tr_edges.push_back(subsidelen);  // (the tile rows below the view's last row are outside: nothing to wait for)
    }
    const bool threaded = tr_edges.size() > 2;
    if (threaded) {
        const int device = Runtime::get().device;
        fx.th = std::thread([&fx, &fetch, device] {
            (void)hipSetDevice(device);
            for (;;) {
                Job* j;
                {
                    std::unique_lock<std::mutex> lk(fx.mu);
                    fx.cv.wait(lk, [&] { return !fx.q.empty() || fx.closing; });
                    if (fx.q.empty()) return;
                    j = fx.q.front();
                    fx.q.pop_front();
                }
                trace("fetch: starts");
                const int rc = fetch(*j);
                trace("fetch: done");
                if (rc != DCDF_OK) {
                    std::lock_guard<std::mutex> lk(fx.mu);
                    if (fx.rc == DCDF_OK) fx.rc = rc;
                }
            }
        });
    }
    for (size_t band = 0; band + 1 < tr_edges.size(); band++) {
        const uint64_t tr0 = tr_edges[band], tr1 = tr_edges[band + 1];
        const size_t i0 = (size_t)(tr0 * subsidelen), i1 = (size_t)(tr1 * subsidelen);
        if (up) {
            trace("main: waits for band", (long)band);
            const int urc = up->wait_rows((uint32_t)std::min<uint64_t>(tr1 * chunks_sidelen, rows));
            if (urc != DCDF_OK) return urc;
            trace("main: has band", (long)band);
        }
        // -- per-instant (min, max) of the band's tiles on the device --
        std::vector<MinMaxTile> mt;
        std::vector<size_t> mt_tile;
        for (size_t i = i0; i < i1; i++)
            if (inside[i]) {
                const dcdf_tile_desc& t = tiles[i];
                mt.push_back(MinMaxTile{t.base, t.stride_t, t.stride_r, t.stride_c, t.rows, t.cols});
                mt_tile.push_back(i);
            }
        if (!mt.empty()) {
            DevBuf d_mt, d_out, d_st;
            K2R_HIP(d_mt.alloc(mt.size() * sizeof(MinMaxTile)));
            K2R_HIP(d_out.alloc(mt.size() * instants * 16ull));
            K2R_HIP(d_st.alloc(4));
            K2R_HIP(hipMemcpy(d_mt.p, mt.data(), mt.size() * sizeof(MinMaxTile), hipMemcpyHostToDevice));
            K2R_HIP(hipMemset(d_st.p, 0, 4));
            hipLaunchKernelGGL(k_tile_minmax, dim3((uint32_t)(mt.size() * instants)), dim3(256), 0, 0, d_mt.as<MinMaxTile>(), instants, buf.dtype,
                               (uint32_t)buf.fractional_bits, (uint32_t)buf.round, d_out.as<int64_t>(), d_st.as<int32_t>());
            K2R_HIP(hipGetLastError());
            std::vector<int64_t> got(mt.size() * instants * 2ull);
            K2R_HIP(hipMemcpy(got.data(), d_out.p, got.size() * 8, hipMemcpyDeviceToHost));
            int32_t st = 0;
            K2R_HIP(hipMemcpy(&st, d_st.p, 4, hipMemcpyDeviceToHost));
            if (st != 0) return map_status(st);  // to_fixed panics (fixed.rs:39-70)
            for (size_t q = 0; q < mt.size(); q++)
                std::memcpy(&mm[2ull * mt_tile[q] * instants], &got[2ull * q * instants], 16ull * instants);
        }
        if (!threaded) tm.lap("tile min/max");
        // -- fractional bits of every float tile that will be built (fixed.rs:96-159), for the whole band at once --
        std::vector<size_t> frac_of(i1 - i0, 0);
        std::vector<int32_t> frac_rnd, frac_bits, frac_st;
        if (buf.dtype == DCDF_F32 || buf.dtype == DCDF_F64) {
            std::vector<dcdf_tile_desc> ft;
            for (size_t i = i0; i < i1; i++)
                if (inside[i] && !can_elide_tile(i)) {
                    frac_of[i - i0] = ft.size();
                    ft.push_back(tiles[i]);
                }
            frac_rnd.resize(ft.size());
            frac_bits.resize(ft.size());
            frac_st.resize(ft.size());
            if (!ft.empty()) {
                const int rc = suggest_fraction_batch(ft.data(), ft.size(), frac_rnd.data(), frac_bits.data(), frac_st.data());
                if (rc != DCDF_OK) return rc;
            }
        }
        // -- elision, sub-builds (superchunk.rs:144-181) --
        std::vector<dcdf_tile_desc> chunk_descs;
        std::unique_ptr<Job> job(new Job());
        for (size_t i = i0; i < i1; i++) {
            if (!inside[i] || can_elide_tile(i)) continue;
            elided[i] = 0;
            dcdf_tile_desc t = tiles[i];
            bool build_subchunk = at_bottom;
            if (!at_bottom) build_subchunk = levels_needed(std::max(t.rows, t.cols), k) <= sublevels[0];  // superchunk.rs:155-165
            if (t.dtype == DCDF_F32 || t.dtype == DCDF_F64) {  // sub_buffer.compute_fractional_bits() (mmbuffer.rs:596-613)
                const size_t fi = frac_of[i - i0];  // (all tiles of the band in two launches, before this loop)
                if (frac_st[fi] != DCDF_OK) return frac_st[fi];
                int32_t bits = frac_bits[fi];
                if (t.round) bits = std::min<int32_t>(bits, t.fractional_bits);
                else if (frac_rnd[fi]) return DCDF_ERR_PRECISION;  // panic!("loss of precision")
                t.fractional_bits = (uint8_t)bits;
            }
            Sub sb{};
            sb.tile = i;
            sb.chunk = build_subchunk;
            if (build_subchunk) {
                chunk_descs.push_back(t);
                job->chunk_sub.push_back(subs.size());
            } else {
                Level sub;
                const int rc = build_level(cx, t, sublevels, n_sub, &sub);
                if (rc != DCDF_OK) return rc;
                sb.obj = sub.node;
                sb.size = sub.size_self + 1;  // MMStruct3::size (mmstruct.rs:187-197)
                sb.snapshots = sub.snapshots;
                sb.logs = sub.logs;
            }
            subs.push_back(std::move(sb));
        }
        if (chunk_descs.empty()) continue;
        // every Chunk::build of the band in ONE launch (superchunk.rs:169) on a device-resident session
        if (!threaded) tm.lap("fractional bits, tiling");
        if (threaded) trace("main: min/max, elision done; session create");
        int rc = dcdf_encoder_create(chunk_descs.data(), chunk_descs.size(), k, 0, &job->e);
        if (rc != DCDF_OK) return rc;
        Job* const jp = job.get();
        fx.jobs.push_back(std::move(job));  // (owns the session from here on)
        if (!threaded) tm.lap("encoder_create");
        if (threaded) trace("main: session run");
        rc = dcdf_encoder_run(jp->e, nullptr);
        if (rc != DCDF_OK) return rc;
        if (threaded) trace("main: session done");
        if (!threaded) tm.lap("encoder_run");
        for (size_t q = 0; q < chunk_descs.size(); q++) {
            int32_t st = 0;
            rc = dcdf_encoder_result(jp->e, q, &st, nullptr, nullptr, nullptr, nullptr);
            if (rc != DCDF_OK) return rc;
            if (st != DCDF_OK) return st;
        }
        if (threaded) {
            {
                std::lock_guard<std::mutex> lk(fx.mu);
                fx.q.push_back(jp);
            }
            fx.cv.notify_all();
        } else {
            rc = fetch(*jp);
            if (rc != DCDF_OK) return rc;
            tm.lap("download + framing + sha256");
        }
    }
    {
        const int frc = fx.close();
        if (frc != DCDF_OK) return frc;
    }
    if (threaded) tm.lap("bands: upload | min/max, sessions | download, sha256");
    // ---- instant-major min / max (superchunk.rs:190-198) ----
    std::vector<int64_t> mins(n_tiles * (size_t)instants), maxs(n_tiles * (size_t)instants);
    for (uint32_t t = 0; t < instants; t++)
        for (size_t i = 0; i < n_tiles; i++) {
            mins[(size_t)t * n_tiles + i] = mm[2ull * (i * instants + t)];
            maxs[(size_t)t * n_tiles + i] = mm[2ull * (i * instants + t) + 1];
        }
    // ---- references with de-duplication (superchunk.rs:199-236) ----
    std::vector<const std::string*> to_hash;  // (only the nested superchunk nodes are left to hash: small)
    std::vector<size_t> hash_sub;
    for (size_t q = 0; q < subs.size(); q++)
        if (chunk_cid[q].empty()) {
            to_hash.push_back(&subs[q].obj);
            hash_sub.push_back(q);
        }
    std::vector<std::string> hashed;
    int rc = hash_objects(to_hash, hashed);
    if (rc != DCDF_OK) return rc;
    std::vector<std::string> cids(chunk_cid.begin(), chunk_cid.begin() + (long)subs.size());
    for (size_t j = 0; j < hash_sub.size(); j++) cids[hash_sub[j]] = hashed[j];
    std::vector<std::string> external;
    std::map<std::string, uint32_t> ext_index;
    std::vector<int64_t> refs(n_tiles, -1);
    uint64_t sizes = 0;
    out->elided = 0; out->snapshots = 0; out->logs = 0;
    size_t si = 0;
    for (size_t i = 0; i < n_tiles; i++) {
        if (elided[i]) {
            out->elided++;
            continue;
        }
        Sub& s = subs[si];
        const std::string& cid = cids[si];
        si++;
        sizes += s.size;
        if (s.chunk) {
            save(cx, cid, s.blob);
            s.blob = Blob{};
        } else {
            save(cx, cid, blob_of(s.obj));
        }
        auto it = ext_index.find(cid);
        uint32_t index;
        if (it == ext_index.end()) {
            index = (uint32_t)external.size();
            external.push_back(cid);
            ext_index[cid] = index;
        } else {
            index = it->second;
        }
        refs[i] = index;
        out->snapshots += s.snapshots;
        out->logs += s.logs;
    }
    tm.lap("references");
    std::string links = header(1);  // NODE_LINKS (links.rs:65-76)
    put_u32(links, (uint32_t)external.size());
    for (const std::string& c : external) links += c;
    const uint64_t size_external = 7 + 4 + 36ull * external.size();
    std::vector<std::string> lcid;
    rc = hash_objects({&links}, lcid);
    if (rc != DCDF_OK) return rc;
    save(cx, lcid[0], blob_of(links));
    // ---- the node (superchunk.rs:672-706) ----
    std::string body;
    put_u32(body, instants);
    put_u32(body, rows);
    put_u32(body, cols);
    put_u32(body, (uint32_t)sidelen);
    body.push_back((char)levels[0]);
    put_u32(body, (uint32_t)chunks_sidelen);
    put_u32(body, (uint32_t)subsidelen);
    body.push_back((char)((buf.dtype == DCDF_F32 || buf.dtype == DCDF_F64) ? buf.fractional_bits : 0));
    body.push_back((char)buf.dtype);
    put_u32(body, (uint32_t)n_tiles);
    for (size_t i = 0; i < n_tiles; i++) {  // Reference::write_to (superchunk.rs:843-861)
        if (refs[i] < 0) body.push_back(0);
        else {
            body.push_back(2);
            put_u32(body, (uint32_t)refs[i]);
        }
    }
    body += lcid[0];
    put_u32(body, 0);  // n_local
    std::string dmax, dmin;
    rc = dac_bytes(maxs, dmax);
    if (rc != DCDF_OK) return rc;
    rc = dac_bytes(mins, dmin);
    if (rc != DCDF_OK) return rc;
    body += dmax;
    body += dmin;
    out->node = header(2);
    out->node.push_back(5);  // NODE_SUPERCHUNK
    out->node += body;
    // Superchunk::size() (superchunk.rs:652-670): header + every field save_to writes EXCEPT the `encoding` byte of
    // superchunk.rs:692 -- the reference's formula leaves it out, and a drop-in reports what the reference reports
    out->size_self = 7 + body.size() - 1;
    out->size = out->size_self + size_external + sizes;
    out->external = (uint32_t)external.size();
    return DCDF_OK;
}

}  // namespace

extern "C" void dcdf_free_superchunk(dcdf_superchunk* s) {
    if (!s) return;
    for (size_t i = 0; i < s->n_objects; i++) std::free(s->objects[i].bytes);
    std::free(s->objects);
    std::free(s);
}

extern "C" int dcdf_superchunk_build(const dcdf_tile_desc* buffer, const uint32_t* levels, size_t n_levels, int k, int mem,
                                     dcdf_superchunk** out) {
    if (!buffer || !levels || !out || n_levels < 2 || k < 2 || k > 16 || (mem != DCDF_MEM_HOST && mem != DCDF_MEM_DEVICE)) return DCDF_ERR_BAD_ARG;
    if (!buffer->base || buffer->instants == 0 || buffer->rows == 0 || buffer->cols == 0) return DCDF_ERR_BAD_ARG;
    if (buffer->dtype != DCDF_I32 && buffer->dtype != DCDF_I64 && buffer->dtype != DCDF_F32 && buffer->dtype != DCDF_F64) return DCDF_ERR_BAD_ARG;
    if (!Runtime::get().ok) return DCDF_ERR_NO_DEVICE;
    dcdf_tile_desc dev = *buffer;
    DevBuf stage;
    StepTimer tm_up;
    Upload up;  // (joined before `stage` is released)
    const bool from_host = mem == DCDF_MEM_HOST;
    if (from_host) {  // the whole view goes to HBM once; every tile below is a strided view of that copy
        const size_t es = esize(buffer->dtype);
        const uint64_t T = buffer->instants, R = buffer->rows, Cc = buffer->cols;
        K2R_HIP(stage.alloc_pooled((size_t)(T * R * Cc) * es));
        dev.base = stage.p;
        dev.stride_c = 1;
        dev.stride_r = buffer->cols;
        dev.stride_t = (int64_t)buffer->rows * buffer->cols;
        // Bands: whole rows of bottom-level tiles, a few hundred MB each (K2R_SC_BAND_MB; 0 = one band), when the top level's tiles
        // are chunks.  The first band's tiles are being encoded while the second is on the link, and so on (build_level).
        uint64_t band_mb = 512;
        if (const char* bm = std::getenv("K2R_SC_BAND_MB")) band_mb = (uint64_t)std::max(0, std::atoi(bm));
        uint32_t band_rows = (uint32_t)R;
        uint64_t cs = 1;  // rows of one tile of the top level's grid
        if (n_levels == 2 && band_mb) {
            for (uint32_t i = 0; i < levels[1] && cs < R; i++) cs *= (uint64_t)k;
            const uint64_t bytes_per_tile_row = T * std::min<uint64_t>(cs, R) * Cc * es;
            const uint64_t tile_rows = std::max<uint64_t>(1, ((band_mb << 20) + bytes_per_tile_row - 1) / bytes_per_tile_row);
            if (cs * tile_rows < R) band_rows = (uint32_t)(cs * tile_rows);
        }
        if (band_rows < R) {
            for (uint64_t r = 0; r < R; r += band_rows) up.edges.push_back((uint32_t)r);
            up.edges.push_back((uint32_t)R);
            // the last band is what the call still has to encode and fetch once the upload is over: halve it (whole tile rows)
            const size_t nb = up.edges.size() - 1;
            const uint64_t last0 = up.edges[nb - 1], tiles = (R - last0 + cs - 1) / cs;
            if (tiles >= 2) up.edges.insert(up.edges.end() - 1, (uint32_t)(last0 + (tiles - tiles / 2) * cs));
        }
        trace("call: upload starts");
        const int urc = up.start(*buffer, (uint8_t*)stage.p);
        if (urc != DCDF_OK) return urc;
    }
    Ctx cx;
    cx.k = k;
    Level top;
    const int rc = build_level(cx, dev, levels, n_levels, &top, from_host ? &up : nullptr);
    up.join();
    if (from_host) tm_up.lap("whole call below the node's own hash");
    if (rc != DCDF_OK) return rc;
    std::vector<std::string> tcid;
    const int rc2 = hash_objects({&top.node}, tcid);
    if (rc2 != DCDF_OK) return rc2;
    dcdf_superchunk* s = (dcdf_superchunk*)std::calloc(1, sizeof(dcdf_superchunk));
    if (!s) return DCDF_ERR_NOMEM;
    s->n_objects = cx.objects.size() + 1;
    s->objects = (dcdf_stored_object*)std::calloc(s->n_objects, sizeof(dcdf_stored_object));
    if (!s->objects) {
        std::free(s);
        return DCDF_ERR_NOMEM;
    }
    if (cx.nomem) {
        std::free(s->objects);
        std::free(s);
        return DCDF_ERR_NOMEM;
    }
    for (size_t i = 0; i + 1 < s->n_objects; i++) {  // the objects change hands as they are (no copy)
        s->objects[i].bytes = cx.objects[i].p;
        s->objects[i].len = cx.objects[i].n;
        std::memcpy(s->objects[i].cid, cx.cids[i].data(), 36);
        cx.objects[i].p = nullptr;
    }
    {
        const Blob t = blob_of(top.node);
        if (!t.p) {
            dcdf_free_superchunk(s);
            return DCDF_ERR_NOMEM;
        }
        s->objects[s->n_objects - 1].bytes = t.p;
        s->objects[s->n_objects - 1].len = t.n;
        std::memcpy(s->objects[s->n_objects - 1].cid, tcid[0].data(), 36);
    }
    s->size = top.size;
    s->elided = top.elided;
    s->local = 0;
    s->external = top.external;
    s->snapshots = top.snapshots;
    s->logs = top.logs;
    *out = s;
    return DCDF_OK;
}
