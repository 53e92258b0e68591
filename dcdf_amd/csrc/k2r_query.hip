// k2r_query.hip -- C ABI, query side: open (parse + upload), get / fill_cell / fill_window / search,
// single-chunk and batched.  Kernels walk the serialized big-endian bytes in HBM via k2r_decode.h.
#include <hip/hip_runtime.h>

#include <chrono>
#include <mutex>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <type_traits>
#include <vector>

#include "k2r_decode.h"
#include "k2r_runtime.h"

using namespace k2r;

struct dcdf_chunk {
    // one per instant, stream order == instant order.  A chunk opened from device memory (dcdf_chunk_open_batch) has them on
    // the device only; the few host-side readers fetch them on first use (host_descs)
    mutable std::vector<InstDesc> descs;
    mutable std::once_flag descs_once;
    uint32_t k0 = 0, sidelen0 = 0;  // of instant 0 (== every instant's)
    uint32_t instants = 0, rows = 0, cols = 0, n_blocks = 0;
    int32_t encoding = 0;
    uint32_t fbits = 0;
    size_t len = 0;
    DevBuf d_bytes, d_descs;
    // where the device-side views live: the chunk's own buffers (dcdf_chunk_open) or a slab shared by a batch
    // (dcdf_chunk_open_batch); make_ref() reads only these
    const uint8_t* p_bytes = nullptr;
    const InstDesc* p_descs = nullptr;
    const void* p_top = nullptr;
    const void* p_top_mm = nullptr;
    std::shared_ptr<void> store;  // keeps a batch's slab alive until its last chunk is closed
    // k = 2, sidelen 32..256: for every instant the walk's state at each node of side 16 (k_top_table, built at open): the wave
    // walks of fill_window / search start there instead of at the root (an item begins with
    // the entries of the squares it meets)
    DevBuf d_top, d_top_mm;
    uint32_t top_g = 0;  // squares per side (sidelen / 16), 0 = no table
    // every stored value of every instant lies in [-2^30, 2^30) (from the root extremes): the query walks then run on 32-bit
    // values (NodeStT<int32_t>).  (A crafted chunk whose inner Dac values contradict its roots decodes to different garbage than
    // with 64-bit arithmetic; no address depends on a value.)
    bool narrow32 = false;
    // per instant: a single-node UNIFORM log over a multi-node snapshot.  The reference's search (log.rs:519-702) never reads
    // eqB[0] and descends the snapshot with the log's (min, max) pair as if it were "equal": its result there is not the set of
    // cells in range, so such instants are searched by the per-thread replica of that descent, not by the decoding wave walk.
    std::vector<uint8_t> search_quirk;
};

// the host copy of a chunk's instant descriptors
static const std::vector<InstDesc>& host_descs(const dcdf_chunk* h) {
    std::call_once(h->descs_once, [h] {
        if (h->descs.empty() && h->p_descs && h->instants) {
            h->descs.resize(h->instants);
            if (hipMemcpy(h->descs.data(), h->p_descs, (size_t)h->instants * sizeof(InstDesc), hipMemcpyDeviceToHost) != hipSuccess) {
                (void)hipGetLastError();
                h->descs.clear();
            }
        }
    });
    return h->descs;
}

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

struct TopEnt {  // the walk's state at one node (log.rs:360-361): both first-child indices NONE = its square has the one value mt + ms
    uint32_t bt, bs;
    int32_t mt, ms;  // (a chunk with a value beyond int32 gets no table: k_top_table reports it)
};
struct TopMM {  // smallest and largest value inside the same square (the reference's own pruning bounds: log.rs:573-574)
    int32_t vmin, vmax;
};
struct ChunkRef {  // device-visible handle of an opened chunk
    const uint8_t* bytes;
    const InstDesc* descs;
    uint32_t instants, rows, cols, fbits;
    const TopEnt* top;  // [instant][top_g * top_g] or null
    const TopMM* top_mm;  // the same squares' value ranges (search prunes with them)
    uint32_t top_g, _pad;
};

// ---- host-side parser (chunk.rs:247-266, block.rs:99-109, snapshot.rs:62-81, log.rs:68-89,
//      bitmap.rs:142-164, dac.rs:48-63): records byte offsets instead of materialising vectors ----
struct Cursor {
    const uint8_t* p;
    size_t n, pos = 0;
    bool ok = true;
    K2R_HD bool need(size_t k) {
        if (!ok || pos + k > n || pos + k < pos) ok = false;
        return ok;
    }
    K2R_HD uint8_t u8() { return need(1) ? p[pos++] : 0; }
    K2R_HD uint32_t u32() {
        if (!need(4)) return 0;
        uint32_t v = load_be32(p + pos);
        pos += 4;
        return v;
    }
    K2R_HD void skip(size_t k) {
        if (need(k)) pos += k;
    }
};
K2R_HD void parse_bitmap(Cursor& c, BmDesc& d) {
    d.len = c.u32();
    d.k = c.u32();
    if (d.k == 0) c.ok = false;
    if (!c.ok) return;
    d.idx_off = (uint32_t)c.pos;
    c.skip(4ull * (d.len / 32 / d.k));
    d.words_off = (uint32_t)c.pos;
    c.skip(4ull * ((d.len + 31) / 32));
}
K2R_HD void parse_dac(Cursor& c, DacDesc& d) {
    d = DacDesc{};
    d.nlev = c.u8();
    if (d.nlev > 8) c.ok = false;
    for (uint32_t l = 0; l < d.nlev && c.ok; l++) {
        parse_bitmap(c, d.bm[l]);
        d.bytes_off[l] = (uint32_t)c.pos;
        c.skip(d.bm[l].len);
    }
}
K2R_HD void parse_inst(Cursor& c, InstDesc& d, bool is_log, uint32_t snap) {
    d = InstDesc{};
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


// one (query, instant) of a search: its window bitmap
struct SearchItem {
    uint32_t query, instant;  // instant is absolute within the chunk
    uint64_t bits_off;        // u32 words: the item's window bitmap, row-major over the query window (per-thread descent)
    uint32_t w0, ncb;         // wave walk: first of the item's pieces (<= 64 x 64 cells from the window's origin, row-major; a bitmap
                              // of 64 rows x 2 words each) and pieces per row; w0 == SI_FLAT: the flat bitmap above is the one in use
};
constexpr uint32_t SI_FLAT = 0xffffffffu;
// ---- wave-cooperative window decode ---------------------------------------------------------------------------------------
// fill_window the MI355X way: one WAVE per (query, instant, sub-window of at most 32 x 32 cells) walks the nodes that
// cover the sub-window ONCE, level by level (snapshot.rs:237-301, log.rs:349-508 are depth-first recursions over the same
// nodes): lane = (frontier node, child).  A node's rank / Dac hops are paid once per node instead of once per cell; the
// frontier lives in LDS and is compacted with ballot + mbcnt; uniform or "equal" subtrees become rectangle fills done by
// the whole wave; the last two levels (a node of side k and its cells) are finished by the lane that owns the node.
struct WinItem {
    uint32_t chunk, inst;
    uint16_t top, bottom, left, right;  // sub-window, chunk coordinates, half-open
    uint32_t out_sr;                     // output row stride in elements (column stride 1)
    uint64_t out_off;                    // element offset of cell (top, left) of this instant in `out`
};
constexpr int WQ_CAP = 192;             // frontier entries per wave: nodes of side >= k^2 meeting a 32 x 32 window, all levels
constexpr uint32_t WQ_NONE = 0xffffffffu;
struct WaveQ {
    uint32_t it[WQ_CAP], is[WQ_CAP], org[WQ_CAP];  // first child of the node in the log / snapshot tree (or NONE), origin row << 16 | col
    int64_t mt[WQ_CAP], ms[WQ_CAP];                 // log.rs:360-361 max_t, max_s
};
struct GpuExecScan {  // inclusive prefix sum over the 64 lanes (DPP), as GpuExec::wave_incl_scan
    __device__ __forceinline__ static uint32_t incl(uint32_t v) {
        int x = (int)v;
        x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
        x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
        x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
        x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
        x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
        x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
        return (uint32_t)x;
    }
};
// V = int64_t in general; int32_t for chunks whose stored values all lie in [-2^30, 2^30) (dcdf_chunk::narrow32: every node
// extreme and every log difference then fits 32 bits), which halves the walk's arithmetic and its register footprint
template <class V>
struct NodeStT {
    uint32_t bt, bs;  // index of the node's FIRST CHILD in the log / snapshot tree (1 + rank(T, node) * k^2), or NONE
    V mt, ms;         // log.rs:360-361 max_t, max_s
};
typedef NodeStT<int64_t> NodeSt;
// One step of the synchronized descent (log.rs:392-505; snapshot.rs:281-299 when there is no log): child c of a node.
// Returns true when the child's whole square has one value (*val), else the child's state in *o.  The rank that locates
// the child's own children is computed HERE, next to the child's other loads (they are independent of each other), so that
// the next level does not start with a round trip of its own.
__device__ __forceinline__ bool window_child(const uint8_t* b, const InstDesc& S, const InstDesc* L, const NodeSt& p, uint32_t c,
                                             uint32_t k2, NodeSt* o, int64_t* val) {
    const bool has_t = p.bt != WQ_NONE, has_s = p.bs != WQ_NONE;
    const uint32_t it_ = has_t ? p.bt + c : 0u, is_ = has_s ? p.bs + c : 0u;
    const int64_t mt_ = has_t ? dacd_get(b, L->mx, it_) : p.mt;             // log.rs:397-400
    const int64_t ms_ = has_s ? p.ms - dacd_get(b, S.mx, is_) : p.ms;       // log.rs:412-415
    const bool leaf_t = has_t ? (it_ >= L->T.len || !bmd_get(b, L->T, it_)) : true;
    const bool leaf_s = has_s ? (is_ >= S.T.len || !bmd_get(b, S.T, is_)) : true;
    const uint32_t rt = has_t ? bmd_rank(b, L->T, it_) : 0u, rs = has_s ? bmd_rank(b, S.T, is_) : 0u;
    *val = mt_ + ms_;
    if (leaf_t && leaf_s) return true;
    o->mt = mt_;
    o->ms = ms_;
    if (leaf_s) {
        o->bt = 1 + rt * k2;
        o->bs = WQ_NONE;
        return false;
    }
    if (leaf_t) {
        // rank0(T, it_ + 1) - 1 with T[it_] == 0 is it_ - rank(T, it_)
        if (has_t && !bmd_get(b, L->E, it_ - rt)) return true;  // uniform, not "equal" (log.rs:452-467)
        o->bt = WQ_NONE;
        o->bs = 1 + rs * k2;
        return false;
    }
    o->bt = 1 + rt * k2;
    o->bs = 1 + rs * k2;
    return false;
}

__global__ void __launch_bounds__(256)
k_window_wave(const ChunkRef* __restrict__ chunks, const WinItem* __restrict__ items, uint32_t n_items, void* out, int32_t out_dtype) {
    __shared__ WaveQ wq[4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    WaveQ& q = wq[wave];
    for (uint32_t item = blockIdx.x * 4u + (uint32_t)wave; item < n_items; item += gridDim.x * 4u) {
        const WinItem I = items[item];
        const ChunkRef C = chunks[I.chunk];
        const uint8_t* const b = C.bytes;
        const InstDesc& D = C.descs[I.inst];
        const bool has_log = D.is_log != 0;
        const InstDesc& S = has_log ? C.descs[D.snap] : D;
        const InstDesc* const L = has_log ? &D : nullptr;
        const uint32_t k = D.k, k2 = k * k;
        const uint32_t wtop = I.top, wbot = I.bottom, wleft = I.left, wright = I.right;
        auto put = [&](uint32_t r, uint32_t c, int64_t v) {
            store_typed(out, (int64_t)I.out_off + (int64_t)(r - wtop) * I.out_sr + (c - wleft), out_dtype, v, C.fbits);
        };
        // rectangle [r0, r1) x [c0, c1) (already clipped to the sub-window) <- v, by the whole wave
        auto fill_wave = [&](uint32_t r0, uint32_t r1, uint32_t c0, uint32_t c1, int64_t v) {
            const uint32_t w = c1 - c0, area = (r1 - r0) * w;
            for (uint32_t i = (uint32_t)lane; i < area; i += 64) put(r0 + i / w, c0 + i % w, v);
        };
        // ---- roots (snapshot.rs:211-219, log.rs:315-328) ----
        const bool single_s = !bmd_get(b, S.T, 0);
        const bool single_t = has_log ? !bmd_get(b, L->T, 0) : true;
        const int64_t max_s0 = dacd_get(b, S.mx, 0), max_t0 = has_log ? dacd_get(b, L->mx, 0) : 0;
        const bool all_one = has_log ? (single_t && (single_s || !bmd_get(b, L->E, 0))) : single_s;
        if (all_one) {
            fill_wave(wtop, wbot, wleft, wright, max_t0 + max_s0);
            continue;
        }
        if (lane == 0) {
            q.it[0] = (has_log && !single_t) ? 1u : WQ_NONE;  // children of the root start at 1 (rank(T, 0) == 0)
            q.is[0] = single_s ? WQ_NONE : 1u;
            q.org[0] = 0;
            q.mt[0] = max_t0;
            q.ms[0] = max_s0;
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t lo = 0, hi = 1, side = D.sidelen;
        const uint32_t per = 64u / k2;  // frontier nodes per step (lane = node * k2 + child)
        const uint32_t myn = (uint32_t)lane / k2, myc = (uint32_t)lane % k2;
        // ---- level by level while the children are still at least k x k ----
        while (side > k2) {
            const uint32_t cs = side / k;
            uint32_t next = hi;
            for (uint32_t base = lo; base < hi; base += per) {
                const uint32_t n = base + myn;
                const bool live = myn < per && n < hi;
                bool push = false, fill = false;
                NodeSt o{};
                int64_t val = 0;
                uint32_t r0 = 0, r1 = 0, c0 = 0, c1 = 0, org = 0;
                if (live) {
                    const NodeSt p{q.it[n], q.is[n], q.mt[n], q.ms[n]};
                    const uint32_t po = q.org[n];
                    const uint32_t cr = (po >> 16) + (myc / k) * cs, cc = (po & 0xffffu) + (myc % k) * cs;
                    r0 = cr > wtop ? cr : wtop; r1 = cr + cs < wbot ? cr + cs : wbot;
                    c0 = cc > wleft ? cc : wleft; c1 = cc + cs < wright ? cc + cs : wright;
                    if (r0 < r1 && c0 < c1) {
                        fill = window_child(b, S, L, p, myc, k2, &o, &val);
                        push = !fill;
                        org = (cr << 16) | cc;
                    }
                }
                const unsigned long long bp = __builtin_amdgcn_ballot_w64(push);
                if (push) {
                    const uint32_t pos = next + __builtin_amdgcn_mbcnt_hi((uint32_t)(bp >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bp, 0u));
                    if (pos < (uint32_t)WQ_CAP) {
                        q.it[pos] = o.bt; q.is[pos] = o.bs; q.org[pos] = org; q.mt[pos] = o.mt; q.ms[pos] = o.ms;
                    }
                }
                next += (uint32_t)__builtin_popcountll(bp);
                // fills: small ones by their lane, the others by the wave
                const bool small = (r1 - r0) * (c1 - c0) <= 4;
                if (fill && small)
                    for (uint32_t r = r0; r < r1; r++)
                        for (uint32_t c = c0; c < c1; c++) put(r, c, val);
                unsigned long long bf = __builtin_amdgcn_ballot_w64(fill && !small);
                while (bf) {
                    const int l = __builtin_ctzll(bf);
                    bf &= bf - 1;
                    const uint32_t rr = (uint32_t)__builtin_amdgcn_readlane((int)((r0 << 16) | r1), l);
                    const uint32_t cc = (uint32_t)__builtin_amdgcn_readlane((int)((c0 << 16) | c1), l);
                    const uint32_t vlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)val, l);
                    const uint32_t vhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)val >> 32), l);
                    fill_wave(rr >> 16, rr & 0xffffu, cc >> 16, cc & 0xffffu, (int64_t)(((uint64_t)vhi << 32) | vlo));
                }
            }
            __builtin_amdgcn_wave_barrier();
            lo = hi;
            hi = next < (uint32_t)WQ_CAP ? next : (uint32_t)WQ_CAP;  // (cannot overflow for sub-windows of <= 32 x 32: see the host)
            side = cs;
        }
        // ---- the last two levels: lane = (node of side <= k^2, child); the child's cells are finished by the lane ----
        {
            const uint32_t cs = side / k;  // k (then the grandchildren are cells) or 1 (the children are cells)
            for (uint32_t base = lo; base < hi; base += per) {
                const uint32_t n = base + myn;
                if (!(myn < per && n < hi)) continue;
                const NodeSt p{q.it[n], q.is[n], q.mt[n], q.ms[n]};
                const uint32_t po = q.org[n];
                const uint32_t cr = (po >> 16) + (myc / k) * cs, cc = (po & 0xffffu) + (myc % k) * cs;
                const uint32_t r0 = cr > wtop ? cr : wtop, r1 = cr + cs < wbot ? cr + cs : wbot;
                const uint32_t c0 = cc > wleft ? cc : wleft, c1 = cc + cs < wright ? cc + cs : wright;
                if (!(r0 < r1 && c0 < c1)) continue;
                NodeSt o{};
                int64_t val = 0;
                const bool fill = window_child(b, S, L, p, myc, k2, &o, &val);
                if (fill || cs == 1) {  // (a cell is always a leaf of both trees)
                    for (uint32_t r = r0; r < r1; r++)
                        for (uint32_t c = c0; c < c1; c++) put(r, c, val);
                    continue;
                }
                if (cs != k) {  // sidelen not a power of k (malformed input): plain per-cell descents
                    for (uint32_t r = r0; r < r1; r++)
                        for (uint32_t c = c0; c < c1; c++) put(r, c, inst_get(b, C.descs, I.inst, r, c));
                    continue;
                }
                if (k == 2) {  // the four cells at once: their loads are independent of each other
                    int64_t v[4];
                    NodeSt oo{};
#pragma unroll
                    for (int g = 0; g < 4; g++) (void)window_child(b, S, L, o, (uint32_t)g, k2, &oo, &v[g]);
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const uint32_t r = cr + (uint32_t)(g >> 1), c = cc + (uint32_t)(g & 1);
                        if (r >= r0 && r < r1 && c >= c0 && c < c1) put(r, c, v[g]);
                    }
                    continue;
                }
                for (uint32_t r = r0; r < r1; r++)
                    for (uint32_t c = c0; c < c1; c++) {
                        NodeSt oo{};
                        int64_t v = 0;
                        (void)window_child(b, S, L, o, (r - cr) * k + (c - cc), k2, &oo, &v);
                        put(r, c, v);
                    }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}


// ---- pruned search, any arity with k * k <= 64 -------------------------------------------------------------------------
// The reference's search is a pruned descent (snapshot.rs:347-421, log.rs:553-702): a subtree whose [min, max] misses [lower, upper]
// is skipped, one that lies inside is reported whole.  The same walk as k_window_wave -- lane = (frontier node, child), level by
// level, frontier in LDS -- carrying the node's exact extremes: max_t = Lmax_t + max_s as the window walk has it, and
// min_t = Lmin_t[rank(T_t, i)] + min_s for a node that is internal in the log (log.rs:148), min_s + (max_t - max_s) where the log
// is "equal" or has ended (t = s + constant there), with min_s carried down the snapshot (Lmin_s[rank(T_s, i)] = child_min -
// parent_min for internal nodes, snapshot.rs:140; min = max for its leaves).  Matches are bits of the search item's flat window
// bitmap.  The result set is the reference's wherever its bounds are true bounds; the one shape where they are not (SearchExtra:
// single-node uniform log over a multi-node snapshot) is evaluated as the data it is, on the snapshot with the root's difference.
struct SearchExtra {
    int64_t lower, upper;
    // The reference's Log::search_window has no case for a single-node UNIFORM log over a multi-node snapshot (log.rs:527-548):
    // it never reads eqB[0], seeds min_t with an empty Dac's 0 and descends the snapshot as if the log were "equal" with the
    // root's difference.  Read as data, its result for such an instant is: every cell when min_s(root) >= lower and c <= upper
    // (c = the instant's one value), no cell when min_s(root) > upper or c < lower, and otherwise the cells with
    // lower <= s(cell) + (c - max_s(root)) <= upper.  quirk != 0 makes the walk do exactly that instead of the decode of the
    // instant's true values.
    uint32_t quirk, _pad;
};
struct SearchSt {
    uint32_t bt, bs;
    int64_t mt, ms, mns;  // max_t - max_s so far / max_s / min_s of the node
};
struct WaveQS {
    uint32_t it[WQ_CAP], is[WQ_CAP], org[WQ_CAP];
    int64_t mt[WQ_CAP], ms[WQ_CAP], mns[WQ_CAP];
};
// child c of node p: true = the child's square holds ONE value (*mx); else its state in *o and its extremes in *mn / *mx
__device__ __forceinline__ bool search_child(const uint8_t* b, const InstDesc& S, const InstDesc* L, const SearchSt& p, uint32_t c, uint32_t k2,
                                             SearchSt* o, int64_t* mn, int64_t* mx) {
    const bool has_t = p.bt != WQ_NONE, has_s = p.bs != WQ_NONE;
    const uint32_t it_ = has_t ? p.bt + c : 0u, is_ = has_s ? p.bs + c : 0u;
    const int64_t mt_ = has_t ? dacd_get(b, L->mx, it_) : p.mt;
    const int64_t ms_ = has_s ? p.ms - dacd_get(b, S.mx, is_) : p.ms;
    const bool leaf_t = has_t ? (it_ >= L->T.len || !bmd_get(b, L->T, it_)) : true;
    const bool leaf_s = has_s ? (is_ >= S.T.len || !bmd_get(b, S.T, is_)) : true;
    const uint32_t rt = has_t ? bmd_rank(b, L->T, it_) : 0u, rs = has_s ? bmd_rank(b, S.T, is_) : 0u;
    const int64_t mns_ = has_s ? (leaf_s ? ms_ : p.mns + dacd_get(b, S.mn, rs)) : p.mns;
    *mx = mt_ + ms_;
    *mn = (has_t && !leaf_t) ? dacd_get(b, L->mn, rt) + mns_ : mt_ + mns_;
    if (leaf_t && leaf_s) return true;
    o->mt = mt_;
    o->ms = ms_;
    o->mns = mns_;
    if (leaf_s) {
        o->bt = 1 + rt * k2;
        o->bs = WQ_NONE;
        return false;
    }
    if (leaf_t) {
        if (has_t && !bmd_get(b, L->E, it_ - rt)) return true;  // uniform, not "equal" (log.rs:452-467)
        o->bt = WQ_NONE;
        o->bs = 1 + rs * k2;
        return false;
    }
    o->bt = 1 + rt * k2;
    o->bs = 1 + rs * k2;
    return false;
}
__global__ void __launch_bounds__(256)
k_search_wave(const ChunkRef* __restrict__ chunks, const WinItem* __restrict__ items, uint32_t n_items, const SearchExtra* __restrict__ sx,
              uint32_t* __restrict__ bits, uint32_t* __restrict__ counts) {
    __shared__ WaveQS wq[4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    WaveQS& q = wq[wave];
    for (uint32_t item = blockIdx.x * 4u + (uint32_t)wave; item < n_items; item += gridDim.x * 4u) {
        const WinItem I = items[item];
        const SearchExtra X = sx[item];
        const ChunkRef C = chunks[I.chunk];
        const uint8_t* const b = C.bytes;
        const InstDesc& D = C.descs[I.inst];
        const bool quirk = X.quirk != 0;
        const bool has_log = D.is_log != 0 && !quirk;
        const InstDesc& S = D.is_log != 0 ? C.descs[D.snap] : D;
        const InstDesc* const L = has_log ? &D : nullptr;
        const uint32_t k = D.k, k2 = k * k;
        const uint32_t wtop = I.top, wbot = I.bottom, wleft = I.left, wright = I.right;
        const int64_t lower = X.lower, upper = X.upper;
        uint32_t cnt = 0;
        // I.out_off = bit index of cell (top, left) in the item's flat window bitmap, I.out_sr = the window's width
        auto mark = [&](uint32_t r, uint32_t c) {
            const uint64_t e = I.out_off + (uint64_t)(r - wtop) * I.out_sr + (c - wleft);
            atomicOr(&bits[e >> 5], 1u << (e & 31u));
            cnt++;
        };
        auto mark_wave = [&](uint32_t r0, uint32_t r1, uint32_t c0, uint32_t c1) {
            const uint32_t w = c1 - c0, area = (r1 - r0) * w;
            for (uint32_t i = (uint32_t)lane; i < area; i += 64) mark(r0 + i / w, c0 + i % w);
        };
        // ---- roots ----
        const bool single_s = !bmd_get(b, S.T, 0);
        const bool single_t = has_log ? !bmd_get(b, L->T, 0) : true;
        const int64_t max_s0 = dacd_get(b, S.mx, 0), min_s0 = single_s ? max_s0 : dacd_get(b, S.mn, 0);  // (a leaf root has no Lmin entry)
        int64_t max_t0 = has_log ? dacd_get(b, L->mx, 0) : 0;
        bool done = false;
        if (quirk) {  // (log.rs:527-586 on this shape: SearchExtra)
            const int64_t c1 = dacd_get(b, D.mx, 0) + max_s0;
            if (min_s0 >= lower && c1 <= upper) {
                mark_wave(wtop, wbot, wleft, wright);
                done = true;
            } else if (min_s0 > upper || c1 < lower) {
                done = true;
            }
            max_t0 = c1 - max_s0;  // the walk below: s(cell) + (c - max_s(root))
        } else {
            const bool all_one = has_log ? (single_t && (single_s || !bmd_get(b, L->E, 0))) : single_s;
            if (all_one) {
                const int64_t v = max_t0 + max_s0;
                if (lower <= v && v <= upper) mark_wave(wtop, wbot, wleft, wright);
                done = true;
            }
        }
        if (!done) {
            if (lane == 0) {
                q.it[0] = (has_log && !single_t) ? 1u : WQ_NONE;
                q.is[0] = single_s ? WQ_NONE : 1u;
                q.org[0] = 0;
                q.mt[0] = max_t0;
                q.ms[0] = max_s0;
                q.mns[0] = min_s0;
            }
            __builtin_amdgcn_wave_barrier();
            uint32_t lo = 0, hi = 1, side = D.sidelen;
            const uint32_t per = 64u / k2;
            const uint32_t myn = (uint32_t)lane / k2, myc = (uint32_t)lane % k2;
            while (side > k2) {
                const uint32_t cs = side / k;
                uint32_t next = hi;
                for (uint32_t base = lo; base < hi; base += per) {
                    const uint32_t n = base + myn;
                    const bool live = myn < per && n < hi;
                    bool push = false, fill = false;
                    SearchSt o{};
                    uint32_t r0 = 0, r1 = 0, c0 = 0, c1 = 0, org = 0;
                    if (live) {
                        const SearchSt p{q.it[n], q.is[n], q.mt[n], q.ms[n], q.mns[n]};
                        const uint32_t po = q.org[n];
                        const uint32_t cr = (po >> 16) + (myc / k) * cs, cc = (po & 0xffffu) + (myc % k) * cs;
                        r0 = cr > wtop ? cr : wtop; r1 = cr + cs < wbot ? cr + cs : wbot;
                        c0 = cc > wleft ? cc : wleft; c1 = cc + cs < wright ? cc + cs : wright;
                        if (r0 < r1 && c0 < c1) {
                            int64_t mn = 0, mx = 0;
                            const bool one = search_child(b, S, L, p, myc, k2, &o, &mn, &mx);
                            if (one) fill = lower <= mx && mx <= upper;
                            else if (mx < lower || mn > upper) fill = false;      // nothing of this subtree is in range
                            else if (mn >= lower && mx <= upper) fill = true;     // all of it is
                            else push = true;
                            org = (cr << 16) | cc;
                        }
                    }
                    const unsigned long long bp = __builtin_amdgcn_ballot_w64(push);
                    if (push) {
                        const uint32_t pos = next + __builtin_amdgcn_mbcnt_hi((uint32_t)(bp >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bp, 0u));
                        if (pos < (uint32_t)WQ_CAP) {
                            q.it[pos] = o.bt; q.is[pos] = o.bs; q.org[pos] = org; q.mt[pos] = o.mt; q.ms[pos] = o.ms; q.mns[pos] = o.mns;
                        }
                    }
                    next += (uint32_t)__builtin_popcountll(bp);
                    const bool small = (r1 - r0) * (c1 - c0) <= 4;
                    if (fill && small)
                        for (uint32_t r = r0; r < r1; r++)
                            for (uint32_t c = c0; c < c1; c++) mark(r, c);
                    unsigned long long bf = __builtin_amdgcn_ballot_w64(fill && !small);
                    while (bf) {
                        const int l = __builtin_ctzll(bf);
                        bf &= bf - 1;
                        const uint32_t rr = (uint32_t)__builtin_amdgcn_readlane((int)((r0 << 16) | r1), l);
                        const uint32_t cc = (uint32_t)__builtin_amdgcn_readlane((int)((c0 << 16) | c1), l);
                        mark_wave(rr >> 16, rr & 0xffffu, cc >> 16, cc & 0xffffu);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                lo = hi;
                hi = next < (uint32_t)WQ_CAP ? next : (uint32_t)WQ_CAP;
                side = cs;
            }
            // ---- the last two levels ----
            {
                const uint32_t cs = side / k;
                for (uint32_t base = lo; base < hi; base += per) {
                    const uint32_t n = base + myn;
                    if (!(myn < per && n < hi)) continue;
                    const SearchSt p{q.it[n], q.is[n], q.mt[n], q.ms[n], q.mns[n]};
                    const uint32_t po = q.org[n];
                    const uint32_t cr = (po >> 16) + (myc / k) * cs, cc = (po & 0xffffu) + (myc % k) * cs;
                    const uint32_t r0 = cr > wtop ? cr : wtop, r1 = cr + cs < wbot ? cr + cs : wbot;
                    const uint32_t c0 = cc > wleft ? cc : wleft, c1 = cc + cs < wright ? cc + cs : wright;
                    if (!(r0 < r1 && c0 < c1)) continue;
                    SearchSt o{};
                    int64_t mn = 0, mx = 0;
                    const bool one = search_child(b, S, L, p, myc, k2, &o, &mn, &mx);
                    if (one || cs == 1) {  // (a cell is always a leaf of both trees)
                        if (lower <= mx && mx <= upper)
                            for (uint32_t r = r0; r < r1; r++)
                                for (uint32_t c = c0; c < c1; c++) mark(r, c);
                        continue;
                    }
                    if (mx < lower || mn > upper) continue;
                    if (mn >= lower && mx <= upper) {
                        for (uint32_t r = r0; r < r1; r++)
                            for (uint32_t c = c0; c < c1; c++) mark(r, c);
                        continue;
                    }
                    const int64_t shift = quirk ? max_t0 : 0;  // (per-cell descents below decode the true instant; the quirk walks the snapshot)
                    if (cs != k) {  // sidelen not a power of k (malformed input): plain per-cell descents
                        for (uint32_t r = r0; r < r1; r++)
                            for (uint32_t c = c0; c < c1; c++) {
                                const int64_t v = inst_get(b, C.descs, quirk ? D.snap : I.inst, r, c) + shift;
                                if (lower <= v && v <= upper) mark(r, c);
                            }
                        continue;
                    }
                    for (uint32_t r = r0; r < r1; r++)
                        for (uint32_t c = c0; c < c1; c++) {
                            SearchSt oo{};
                            int64_t vn = 0, v = 0;
                            (void)search_child(b, S, L, o, (r - cr) * k + (c - cc), k2, &oo, &vn, &v);
                            if (lower <= v && v <= upper) mark(r, c);
                        }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // the item's count: this wave's matches, summed over its lanes
        for (int o2 = 32; o2 > 0; o2 >>= 1) cnt += __shfl_down(cnt, o2, 64);
        if (lane == 0 && cnt) atomicAdd(&counts[X._pad], cnt);
    }
}

// ---- k = 2, node-wise: one lane per frontier NODE, its four children share their loads ---------------------------------
// Siblings are adjacent in every stream (children of a node sit at base .. base + 3): their four T bits and ranks come from
// ONE 16-byte load of the rank block (+ its index word), their four Lmax bytes from ONE 4-byte load, their continuation bits
// from the same kind of block load, the second bytes of the long ones from one more 4-byte load.  ~14 loads per node for both
// trees instead of ~18 per CHILD.
struct Blk16 {
    uint32_t w[4];
} __attribute__((aligned(1)));
// The chunk bytes are global memory, but a pointer loaded from a table is "generic" to the compiler and every access through
// it a FLAT instruction (slower, and counted in both wait counters): the node-wise walk uses address-space-1 pointers.
typedef const __attribute__((address_space(1))) uint8_t* gbytes;
__device__ __forceinline__ uint32_t gld32(gbytes p) {  // unaligned 4-byte load, native order
    typedef uint32_t __attribute__((aligned(1))) u32u;
    return *(const __attribute__((address_space(1))) u32u*)p;
}
__device__ __forceinline__ uint32_t gld_be32(gbytes p) { return __builtin_bswap32(gld32(p)); }
__device__ __forceinline__ bool gbm_get(gbytes b, const BmDesc& d, uint32_t i) {
    const uint32_t w = i >> 5;
    if (w >= (d.len + 31) / 32) return false;
    return (gld_be32(b + d.words_off + 4 * w) >> (31 - (i & 31))) & 1u;
}
// rank1(T, i) and the four bits i .. i+3 (bit i = 8), bits at or beyond d.len read as 0.  Needs d.k == 4 and i < d.len.
__device__ __forceinline__ uint32_t rank_nib(gbytes b, const BmDesc& d, uint32_t i, uint32_t* nib) {
    const uint32_t w0 = i >> 5, blk = w0 >> 2, sh = i & 31u;
    uint32_t cnt = blk ? gld_be32(b + d.idx_off + 4 * (blk - 1)) : 0u;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(1)));
    const u32x4 raw = *(const __attribute__((address_space(1))) u32x4*)(b + d.words_off + 16 * blk);
    const uint32_t w[4] = {__builtin_bswap32(raw.x), __builtin_bswap32(raw.y), __builtin_bswap32(raw.z), __builtin_bswap32(raw.w)};
    const uint32_t q0 = w0 & 3u;
    uint32_t x = w[0], nx = w[1];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if ((uint32_t)q < q0) cnt += popc32(w[q]);
        if ((uint32_t)q == q0) {
            x = w[q];
            nx = q < 3 ? w[q + 1] : 0u;
        }
    }
    if (q0 == 3 && sh > 28) nx = gld_be32(b + d.words_off + 4 * (w0 + 1));  // the group of four straddles the block's end (3 in 128)
    if (sh) cnt += popc32(x >> (32 - sh));
    uint32_t n4 = (uint32_t)(((((uint64_t)x << 32) | nx) >> (60 - sh)) & 15u);
    const uint32_t valid = d.len - i;  // > 0
    if (valid < 4) n4 &= (0xfu << (4 - valid)) & 0xfu;
    *nib = n4;
    return cnt;
}
// What the node-wise walk needs of one tree, copied out of its InstDesc once per item (wave-uniform: lives in SGPRs instead
// of being re-fetched through a pointer the compiler must assume the output stores alias).
struct TreeRef {
    BmDesc T, E, c0, c1;      // T, eqB, continuation bitmaps of the Lmax Dac's planes 0 and 1
    uint32_t by0, by1, nlev;  // Lmax plane-0 / plane-1 bytes, number of planes
};
typedef const __attribute__((address_space(1))) InstDesc* gdesc;
__device__ __forceinline__ BmDesc bm_copy(const __attribute__((address_space(1))) BmDesc* p) { return BmDesc{p->len, p->k, p->idx_off, p->words_off}; }
__device__ __forceinline__ TreeRef tree_ref(gdesc p) {
    TreeRef t;
    t.T = bm_copy(&p->T); t.E = bm_copy(&p->E); t.c0 = bm_copy(&p->mx.bm[0]); t.c1 = bm_copy(&p->mx.bm[1]);
    t.by0 = p->mx.bytes_off[0]; t.by1 = p->mx.bytes_off[1]; t.nlev = p->mx.nlev;
    return t;
}
__device__ __forceinline__ TreeRef tree_ref(const InstDesc& d) {
    TreeRef t;
    t.T = d.T; t.E = d.E; t.c0 = d.mx.bm[0]; t.c1 = d.mx.bm[1];
    t.by0 = d.mx.bytes_off[0]; t.by1 = d.mx.bytes_off[1]; t.nlev = d.mx.nlev;
    return t;
}
// the four Lmax values at index i .. i+3 (those at or beyond the Dac's length: 0); `full` = the whole Dac, for values of
// three or more bytes (rare)
template <class V>
__device__ __forceinline__ void dac4(gbytes b, const TreeRef& t, const DacDesc& full, uint32_t i, V (&out)[4]) {
    typedef typename std::conditional<sizeof(V) == 4, uint32_t, uint64_t>::type U;
    const uint32_t len = t.c0.len;
    const uint32_t b0 = gld32(b + t.by0 + i);
    uint32_t cb = 0, r0 = 0;
    if (t.nlev > 1) r0 = rank_nib(b, t.c0, i, &cb);
    uint32_t cb1 = 0;
    const uint32_t b1 = gld32(b + t.by1 + r0);  // (no branch: with a single plane by1 = r0 = 0, a harmless read of the chunk's first bytes)
    if (t.nlev > 2 && cb) (void)rank_nib(b, t.c1, r0, &cb1);
    uint32_t q = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        U n = (b0 >> (8 * c)) & 0xffu;
        const bool more = (cb >> (3 - c)) & 1u;
        if (more) {
            if ((cb1 >> (3 - q)) & 1u) {  // three or more bytes: the general walk
                out[c] = i + c < len ? (V)dacd_get((const uint8_t*)b, full, i + c) : (V)0;
                q++;
                continue;
            }
            n |= (U)((b1 >> (8 * q)) & 0xffu) << 8;
            q++;
        }
        out[c] = i + c < len ? (V)((n >> 1) ^ ((U)0 - (n & 1))) : (V)0;
    }
}
template <class V>
struct KidsT {
    NodeStT<V> st[4];
    V val[4];
    uint32_t fill;  // bit c: child c's whole square has the single value val[c]
};
typedef KidsT<int64_t> Kids;
// the four children of node p (log.rs:392-505 / snapshot.rs:281-299 for all of i, j at once)
// bits i .. i + 3 of a bitmap (bit i = 8), bits beyond its words read as 0 like gbm_get; two loads, no branch
__device__ __forceinline__ uint32_t gbm_get4(gbytes b, const BmDesc& d, uint32_t i) {
    const uint32_t nw = (d.len + 31) / 32, w = i >> 5, sh = i & 31u;
    const uint32_t w0 = w < nw ? w : 0u, w1 = w + 1 < nw ? w + 1 : 0u;
    uint32_t x = gld_be32(b + d.words_off + 4 * w0), y = gld_be32(b + d.words_off + 4 * w1);
    x = w < nw ? x : 0u;
    y = w + 1 < nw ? y : 0u;
    return (uint32_t)(((((uint64_t)x << 32) | y) >> (60 - sh)) & 15u);
}
template <class V>
__device__ __forceinline__ void expand4(gbytes b, const TreeRef& S, const DacDesc& Sfull, const TreeRef& L, const DacDesc& Lfull,
                                        const NodeStT<V>& p, KidsT<V>* o) {
    // Every read below is unconditional (a side that has nothing to read reads index 0 and drops the result): behind
    // `if (has_t)` / `if (has_s)` / per-child branches the log's chain of dependent loads, the snapshot's and up to four eqB
    // reads ran one after the other; this way they are in flight together.
    const bool has_t = p.bt != WQ_NONE, has_s = p.bs != WQ_NONE;
    const bool cells_t = !has_t || p.bt >= L.T.len, cells_s = !has_s || p.bs >= S.T.len;  // the children are beyond T: cells
    V dt[4], ds[4];
    dac4(b, L, Lfull, has_t ? p.bt : 0u, dt);
    dac4(b, S, Sfull, has_s ? p.bs : 0u, ds);
    uint32_t tt = 0, ts = 0;  // T nibbles (bit c = 8 >> c) and rank of the first child
    uint32_t rt = rank_nib(b, L.T, cells_t ? 0u : p.bt, &tt), rs = rank_nib(b, S.T, cells_s ? 0u : p.bs, &ts);
    if (cells_t) { tt = 0; rt = 0; }
    if (cells_s) { ts = 0; rs = 0; }
    // eqB bits of the children with T = 0 (log.rs:452-467): child c's is eqB[p.bt + c - rank(T, p.bt + c)] = the (zeros among the
    // children before c)-th bit from eqB[p.bt - rt] on -- four consecutive bits at most
    const uint32_t eq4 = gbm_get4(b, L.E, cells_t ? 0u : p.bt - rt);
    V vt[4], vs[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        vt[c] = has_t ? dt[c] : p.mt;
        vs[c] = has_s ? ds[c] : (V)0;
    }
    o->fill = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const bool bit_t = (tt >> (3 - c)) & 1u, bit_s = (ts >> (3 - c)) & 1u;
        const bool leaf_t = !has_t || cells_t || !bit_t, leaf_s = !has_s || cells_s || !bit_s;
        const uint32_t before_t = popc32(tt >> (4 - c));
        const uint32_t rtc = rt + before_t, rsc = rs + popc32(ts >> (4 - c));  // rank(T, base + c)
        const V mt_ = vt[c], ms_ = has_s ? p.ms - vs[c] : p.ms;
        o->val[c] = mt_ + ms_;
        NodeStT<V>& n = o->st[c];
        n.mt = mt_;
        n.ms = ms_;
        n.bt = WQ_NONE;
        n.bs = WQ_NONE;
        if (leaf_t && leaf_s) {
            o->fill |= 1u << c;
        } else if (leaf_s) {
            n.bt = 1 + rtc * 4;
        } else if (leaf_t) {
            const bool eq = (eq4 >> (3 - ((uint32_t)c - before_t))) & 1u;
            if (has_t && !cells_t && !eq) o->fill |= 1u << c;  // uniform, not "equal" (log.rs:452-467)
            else n.bs = 1 + rsc * 4;
        } else {
            n.bt = 1 + rtc * 4;
            n.bs = 1 + rsc * 4;
        }
    }
}

// (k = 2, items of at most 64 x 64 cells: they meet at most 5 x 5 nodes of side 16, 9 x 9 of side 8, 17 x 17 of side 4 -- 395
//  frontier entries below the top table; walking from the root adds at most 1 + 4 + 4 + 9 above them)
constexpr int WQ2_CAP = 448;
template <class V>
struct WaveQ2T {
    uint32_t it[WQ2_CAP], is[WQ2_CAP], org[WQ2_CAP];
    V mt[WQ2_CAP], ms[WQ2_CAP];
};
typedef WaveQ2T<int64_t> WaveQ2;
// What a search item adds to its WinItem (search = the same walk; instead of storing a cell it tests lower <= v <= upper and
// sets the cell's bit in the item's own bitmap -- 64 rows x 2 words: word = 2 * (row - top) + (column - left) / 32 -- at out[item * 128];
// no two waves share a word, so there is nothing atomic about it and nothing to clear beforehand).
// MW = waves per SIMD the register allocator must leave room for (4 for the 32-bit walk; the 64-bit one's frontier leaves LDS
// for 3 workgroups per CU, so it is built for 3); DENSE64: the batched form's output (int64, unit column stride);
// SEARCH: mark matches (out = the bitmaps, sx = one SearchExtra per item) instead of storing values
template <int MW, bool DENSE64, bool SEARCH = false, class V = int64_t, bool USE_TOP = true>
__global__ void __launch_bounds__(256, MW)
k_window_wave2(const ChunkRef* __restrict__ chunks, const WinItem* __restrict__ items, uint32_t n_items, void* out, int32_t out_dtype,
               const SearchExtra* __restrict__ sx = nullptr) {
    typedef NodeStT<V> NodeSt;
    typedef KidsT<V> Kids;
    __shared__ WaveQ2T<V> wq[4];
    __shared__ uint32_t wbits[4][128];  // SEARCH: the item's matches, two words per row (bit = column - wleft)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    WaveQ2T<V>& q = wq[wave];
    uint32_t* const rowbits = wbits[wave];
    for (uint32_t item = blockIdx.x * 4u + (uint32_t)wave; item < n_items; item += gridDim.x * 4u) {
        const WinItem I = items[item];
        const ChunkRef C = chunks[I.chunk];
        const uint8_t* const b = C.bytes;
        const gbytes gb = (gbytes)C.bytes;
        const gdesc gD = (gdesc)C.descs + I.inst;
        const bool has_log = gD->is_log != 0;
        const gdesc gS = has_log ? (gdesc)C.descs + gD->snap : gD;
        const InstDesc& D = C.descs[I.inst];                       // (generic views: only for the rare general Dac walk)
        const InstDesc& SD = has_log ? C.descs[gD->snap] : D;
        const TreeRef S = tree_ref(gS), L = tree_ref(gD);          // (L is only looked at when has_log)
        const uint32_t sidelen0 = gD->sidelen;
        const uint32_t wtop = I.top, wbot = I.bottom, wleft = I.left, wright = I.right, osr = I.out_sr;
        const int64_t obase = (int64_t)I.out_off - (int64_t)wtop * osr - (int64_t)wleft;  // element offset of chunk cell (0, 0)
        int64_t s_lo = 0, s_hi = 0;
        bool quirk = false;
        if (SEARCH) {
            s_lo = sx[item].lower;
            s_hi = sx[item].upper;
            quirk = __builtin_amdgcn_readfirstlane((int)sx[item].quirk) != 0;
            rowbits[2 * lane] = 0;
            rowbits[2 * lane + 1] = 0;
            __builtin_amdgcn_wave_barrier();
        }
        auto put = [&](uint32_t r, uint32_t c, int64_t v) {
            if (SEARCH) {
                const uint32_t j = c - wleft;
                if (s_lo <= v && v <= s_hi) atomicOr(&rowbits[2 * (r - wtop) + (j >> 5)], 1u << (j & 31u));
                return;
            }
            const int64_t off = obase + (int64_t)(r * osr + c);
            if (DENSE64) ((int64_t*)out)[off] = v;
            else store_typed(out, off, out_dtype, v, C.fbits);
        };
        auto flush_bits = [&]() {
            if (!SEARCH) return;
            __builtin_amdgcn_wave_barrier();
            ((uint32_t*)out)[(uint64_t)item * 128u + 2u * (uint32_t)lane] = rowbits[2 * lane];  // (rows beyond the item: 0)
            ((uint32_t*)out)[(uint64_t)item * 128u + 2u * (uint32_t)lane + 1u] = rowbits[2 * lane + 1];
            __builtin_amdgcn_wave_barrier();
        };
        auto fill_wave = [&](uint32_t r0, uint32_t r1, uint32_t c0, uint32_t c1, int64_t v) {
            if (SEARCH) {  // (v is wave-uniform) a rectangle of one value: whole row segments at once
                if (s_lo <= v && v <= s_hi && (uint32_t)lane < r1 - r0 && c1 > c0) {
                    const uint32_t w1 = c1 - c0;
                    const uint64_t m = (w1 >= 64u ? ~0ull : ((1ull << w1) - 1ull)) << (c0 - wleft);
                    uint32_t* const rw = &rowbits[2 * (r0 - wtop + (uint32_t)lane)];
                    if ((uint32_t)m) atomicOr(rw, (uint32_t)m);
                    if ((uint32_t)(m >> 32)) atomicOr(rw + 1, (uint32_t)(m >> 32));
                }
                return;
            }
            const uint32_t w = c1 - c0, area = (r1 - r0) * w;
            if (w <= 32 && area <= 1024) {                             // (every fill below the top table)
                const uint32_t inv = (65536u + w - 1) / w;             // i / w == (i * inv) >> 16 for i < 2048
                for (uint32_t i = (uint32_t)lane; i < area; i += 64) {
                    const uint32_t rr = (i * inv) >> 16;
                    put(r0 + rr, c0 + i - rr * w, v);
                }
            } else {
                for (uint32_t i = (uint32_t)lane; i < area; i += 64) put(r0 + i / w, c0 + i % w, v);
            }
        };
        uint32_t lo = 0, hi = 1, side = sidelen0;
        if (USE_TOP && C.top && !(SEARCH && quirk)) {  // start at the (up to 5 x 5) nodes of side 16 that meet the item (k_top_table)
            typedef __attribute__((address_space(1))) const TopEnt* gtop;
            const uint32_t r16 = wtop >> 4, c16 = wleft >> 4, nr16 = ((wbot - 1u) >> 4) - r16 + 1u, nc16 = ((wright - 1u) >> 4) - c16 + 1u;  // <= 5 each
            const uint32_t qi = (uint32_t)lane / nc16, qj = (uint32_t)lane - qi * nc16;
            const uint32_t cr = (r16 + qi) << 4, cc = (c16 + qj) << 4;
            const bool mine = (uint32_t)lane < nr16 * nc16;
            uint32_t ebt = WQ_NONE, ebs = WQ_NONE;
            int64_t emt = 0, ems = 0;
            if (mine) {
                const gtop e = (gtop)C.top + ((size_t)I.inst * C.top_g + (cr >> 4)) * C.top_g + (cc >> 4);
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 x = *(__attribute__((address_space(1))) const u32x4*)e;
                ebt = x.x; ebs = x.y; emt = (int32_t)x.z; ems = (int32_t)x.w;
            }
            // search: the square's value range decides without a descent when it lies outside or inside [lower, upper]
            // (the reference prunes the same way at every node, log.rs:575-586; here once, at side 16)
            bool skip = false, force = false;
            if (SEARCH && mine) {
                typedef __attribute__((address_space(1))) const TopMM* gmm;
                const gmm m = (gmm)C.top_mm + ((size_t)I.inst * C.top_g + (cr >> 4)) * C.top_g + (cc >> 4);
                const int64_t vmin = m->vmin, vmax = m->vmax;
                skip = vmax < s_lo || vmin > s_hi;
                force = !skip && s_lo <= vmin && vmax <= s_hi;
            }
            const bool isfill = mine && !skip && !force && ebt == WQ_NONE && ebs == WQ_NONE;
            const bool push = mine && !skip && !force && !isfill;
            if (SEARCH) {
                unsigned long long bo = __builtin_amdgcn_ballot_w64(force);
                while (bo) {  // every cell of the square matches: its part of the sub-window at once
                    const int l = __builtin_ctzll(bo);
                    bo &= bo - 1;
                    const uint32_t rr = (uint32_t)__builtin_amdgcn_readlane((int)cr, l), ccl = (uint32_t)__builtin_amdgcn_readlane((int)cc, l);
                    fill_wave(rr > wtop ? rr : wtop, rr + 16 < wbot ? rr + 16 : wbot, ccl > wleft ? ccl : wleft, ccl + 16 < wright ? ccl + 16 : wright, s_lo);
                }
            }
            const unsigned long long bp = __builtin_amdgcn_ballot_w64(push);
            if (push) {
                const uint32_t pos = __builtin_amdgcn_mbcnt_lo((uint32_t)bp, 0u);
                q.it[pos] = ebt; q.is[pos] = ebs; q.org[pos] = (cr << 16) | cc; q.mt[pos] = (V)emt; q.ms[pos] = (V)ems;
            }
            unsigned long long bf = __builtin_amdgcn_ballot_w64(isfill);
            while (bf) {  // squares of one value: their part of the sub-window, by the whole wave
                const int l = __builtin_ctzll(bf);
                bf &= bf - 1;
                const uint32_t rr = (uint32_t)__builtin_amdgcn_readlane((int)cr, l), ccl = (uint32_t)__builtin_amdgcn_readlane((int)cc, l);
                const int64_t v = (int64_t)__builtin_amdgcn_readlane((int)(int32_t)emt, l) + (int64_t)__builtin_amdgcn_readlane((int)(int32_t)ems, l);
                fill_wave(rr > wtop ? rr : wtop, rr + 16 < wbot ? rr + 16 : wbot, ccl > wleft ? ccl : wleft, ccl + 16 < wright ? ccl + 16 : wright, v);
            }
            hi = (uint32_t)__builtin_popcountll(bp);
            side = 16;
            if (hi == 0) {
                flush_bits();
                continue;
            }
        } else {
            const bool single_s = !gbm_get(gb, S.T, 0);
            const bool single_t = has_log ? !gbm_get(gb, L.T, 0) : true;
            const int64_t max_s0 = dacd_get(b, SD.mx, 0), max_t0 = has_log ? dacd_get(b, D.mx, 0) : 0;
            const bool all_one = has_log ? (single_t && (single_s || !gbm_get(gb, L.E, 0))) : single_s;
            if (SEARCH && quirk) {  // (see SearchExtra: the reference's result for this shape, log.rs:527-586)
                const int64_t min_s0 = dacd_get(b, SD.mn, 0), c1 = max_t0 + max_s0;
                if (min_s0 >= s_lo && c1 <= s_hi) {
                    fill_wave(wtop, wbot, wleft, wright, s_lo);  // every cell
                    flush_bits();
                    continue;
                }
                if (min_s0 > s_hi || c1 < s_lo) {
                    flush_bits();
                    continue;
                }
                // else: the snapshot's tree with the root's difference on top of every value (the walk below, it = NONE)
            } else if (all_one) {
                fill_wave(wtop, wbot, wleft, wright, max_t0 + max_s0);
                flush_bits();
                continue;
            }
            if (lane == 0) {
                q.it[0] = (has_log && !single_t) ? 1u : WQ_NONE;
                q.is[0] = single_s ? WQ_NONE : 1u;
                q.org[0] = 0;
                q.mt[0] = (V)max_t0;
                q.ms[0] = (V)max_s0;
            }
        }
        __builtin_amdgcn_wave_barrier();
        while (side > 4) {  // lane = frontier node; children of side >= 4 go to the next frontier
            const uint32_t cs = side >> 1;
            uint32_t next = hi;
            for (uint32_t base = lo; base < hi; base += 64) {
                const uint32_t n = base + (uint32_t)lane;
                const bool live = n < hi;
                Kids kd;
                kd.fill = 0;
                uint32_t po = 0, inter = 0;  // children meeting the sub-window
                if (live) {
                    const NodeSt p{q.it[n], q.is[n], q.mt[n], q.ms[n]};
                    po = q.org[n];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const uint32_t cr = (po >> 16) + (uint32_t)(c >> 1) * cs, cc = (po & 0xffffu) + (uint32_t)(c & 1) * cs;
                        if (cr < wbot && cr + cs > wtop && cc < wright && cc + cs > wleft) inter |= 1u << c;
                    }
                    if (inter) expand4(gb, S, SD.mx, L, D.mx, p, &kd);
                }
                const uint32_t pushm = inter & ~kd.fill, fillm = inter & kd.fill;
                const uint32_t np = popc32(pushm);
                const uint32_t inc = GpuExecScan::incl(np);
                uint32_t pos = next + inc - np;
#pragma unroll
                for (int c = 0; c < 4; c++)
                    if ((pushm >> c) & 1u) {
                        if (pos < (uint32_t)WQ2_CAP) {
                            const uint32_t cr = (po >> 16) + (uint32_t)(c >> 1) * cs, cc = (po & 0xffffu) + (uint32_t)(c & 1) * cs;
                            q.it[pos] = kd.st[c].bt; q.is[pos] = kd.st[c].bs; q.org[pos] = (cr << 16) | cc; q.mt[pos] = kd.st[c].mt; q.ms[pos] = kd.st[c].ms;
                        }
                        pos++;
                    }
                next += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
#pragma unroll
                for (int c = 0; c < 4; c++) {  // uniform / not-"equal" children: their part of the sub-window, by the whole wave
                    unsigned long long bf = __builtin_amdgcn_ballot_w64((fillm >> c) & 1u);
                    const uint32_t cr = (po >> 16) + (uint32_t)(c >> 1) * cs, cc = (po & 0xffffu) + (uint32_t)(c & 1) * cs;
                    while (bf) {
                        const int l = __builtin_ctzll(bf);
                        bf &= bf - 1;
                        const uint32_t rr = (uint32_t)__builtin_amdgcn_readlane((int)cr, l), ccl = (uint32_t)__builtin_amdgcn_readlane((int)cc, l);
                        const uint32_t vlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)kd.val[c], l);
                        const uint32_t vhi = sizeof(V) == 4 ? (uint32_t)((int32_t)vlo >> 31)
                                                            : (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)(int64_t)kd.val[c] >> 32), l);
                        fill_wave(rr > wtop ? rr : wtop, rr + cs < wbot ? rr + cs : wbot, ccl > wleft ? ccl : wleft, ccl + cs < wright ? ccl + cs : wright,
                                  (int64_t)(((uint64_t)vhi << 32) | vlo));
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            lo = hi;
            hi = next < (uint32_t)WQ2_CAP ? next : (uint32_t)WQ2_CAP;
            side = cs;
        }
        // ---- nodes of side 4: the lane finishes its 16 cells ----
#ifdef K2R_DIAG_NO_FINAL
        hi = lo;  // (diagnostic build: time of the upper levels alone; results are wrong)
#endif
        // A lane expands one node of side 4 into its quads.  A quad of one value is written at once; the OPEN quads (two Dac
        // reads each, the expensive part) of all the lanes are compacted into the free tail of the frontier and then taken one
        // per lane: typically a third of the quads are open, so a lane-per-node loop over all four would spend two thirds of
        // its Dac reads on masked-off lanes.
        auto put_quad = [&](uint32_t cr, uint32_t cc, const V (&v)[4]) {
            const bool in = cr >= wtop && cr + 2 <= wbot && cc >= wleft && cc + 2 <= wright;
            if (DENSE64 && !SEARCH && in) {  // two cells of a row per store (16 bytes, any 8-byte alignment)
                typedef long long ll2 __attribute__((ext_vector_type(2)));
                typedef ll2 __attribute__((aligned(8))) ll2u;
                int64_t* const o0 = (int64_t*)out + (obase + (int64_t)(cr * osr + cc));
                *(__attribute__((address_space(1))) ll2u*)o0 = ll2{(long long)v[0], (long long)v[1]};
                *(__attribute__((address_space(1))) ll2u*)(o0 + osr) = ll2{(long long)v[2], (long long)v[3]};
                return;
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t r = cr + (uint32_t)(e >> 1), cl = cc + (uint32_t)(e & 1);
                if (in || (r >= wtop && r < wbot && cl >= wleft && cl < wright)) put(r, cl, v[e]);
            }
        };
        const uint32_t room = ((uint32_t)WQ2_CAP - hi) / 4u;            // nodes per round whose quads fit the tail
        const uint32_t per_round = room < 64u ? room : 64u;
        for (uint32_t base = lo; base < hi; base += (per_round ? per_round : 64u)) {
            const uint32_t n = base + (uint32_t)lane;
            const bool live = n < hi && (per_round == 0 || (uint32_t)lane < per_round);
            Kids kd;
            kd.fill = 0;
            uint32_t pr = 0, pc = 0, openm = 0;
            if (live) {
                const NodeSt p{q.it[n], q.is[n], q.mt[n], q.ms[n]};
                const uint32_t po = q.org[n];
                pr = po >> 16;
                pc = po & 0xffffu;
                expand4(gb, S, SD.mx, L, D.mx, p, &kd);
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const uint32_t cr = pr + 2u * (uint32_t)(c >> 1), cc = pc + 2u * (uint32_t)(c & 1);
                    if (!(cr < wbot && cr + 2 > wtop && cc < wright && cc + 2 > wleft)) continue;  // outside the item
                    if ((kd.fill >> c) & 1u) {
                        const V v[4] = {kd.val[c], kd.val[c], kd.val[c], kd.val[c]};
                        put_quad(cr, cc, v);
                    } else {
                        openm |= 1u << c;
                    }
                }
            }
            if (per_round == 0) {  // (no room in the frontier's tail: the lane finishes its own quads)
#pragma unroll
                for (int c = 0; c < 4; c++)
                    if ((openm >> c) & 1u) {
                        Kids g;
                        expand4(gb, S, SD.mx, L, D.mx, kd.st[c], &g);
                        put_quad(pr + 2u * (uint32_t)(c >> 1), pc + 2u * (uint32_t)(c & 1), g.val);
                    }
                continue;
            }
            const uint32_t np = popc32(openm);
            const uint32_t inc = GpuExecScan::incl(np);
            uint32_t pos = hi + inc - np;
#pragma unroll
            for (int c = 0; c < 4; c++)
                if ((openm >> c) & 1u) {
                    q.it[pos] = kd.st[c].bt; q.is[pos] = kd.st[c].bs; q.mt[pos] = kd.st[c].mt; q.ms[pos] = kd.st[c].ms;
                    q.org[pos] = ((pr + 2u * (uint32_t)(c >> 1)) << 16) | (pc + 2u * (uint32_t)(c & 1));
                    pos++;
                }
            const uint32_t nopen = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            __builtin_amdgcn_wave_barrier();
            for (uint32_t kb = 0; kb < nopen; kb += 64) {  // one open quad per lane: its four cells (expand4 at the cell level:
                const uint32_t m = kb + (uint32_t)lane;     // log.rs:404-420, snapshot.rs:281-299)
                if (m < nopen) {
                    const NodeSt k{q.it[hi + m], q.is[hi + m], q.mt[hi + m], q.ms[hi + m]};
                    const uint32_t ko = q.org[hi + m];
                    const bool ht = k.bt != WQ_NONE, hs = k.bs != WQ_NONE;
                    V vt[4], vs[4], v[4];
                    dac4(gb, L, D.mx, ht ? k.bt : 0u, vt);  // (both unconditional: the two chains of loads are in flight together)
                    dac4(gb, S, SD.mx, hs ? k.bs : 0u, vs);
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = (ht ? vt[e] : k.mt) + (hs ? k.ms - vs[e] : k.ms);
                    put_quad(ko >> 16, ko & 0xffffu, v);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        flush_bits();
        __builtin_amdgcn_wave_barrier();
    }
}
// The walk's state at every node of side 16, for every instant of one chunk (dcdf_chunk::d_top): one wave per instant walks
// the top of the tree(s) breadth-first -- 1, 4, 16, ... nodes -- with the same expand4 as the query walks.
__device__ __forceinline__ void top_table_inst(const ChunkRef& C, const uint32_t inst, TopEnt* __restrict__ table, TopMM* __restrict__ table_mm,
                                               uint32_t* __restrict__ overflow, WaveQ2& q, int64_t* qmt, int64_t* qms) {
    const int lane = threadIdx.x;
    const uint32_t G = C.top_g;
    const uint8_t* const b = C.bytes;
    const gbytes gb = (gbytes)C.bytes;
    const gdesc gD = (gdesc)C.descs + inst;
    const bool has_log = gD->is_log != 0;
    const gdesc gS = has_log ? (gdesc)C.descs + gD->snap : gD;
    const InstDesc& D = C.descs[inst];
    const InstDesc& SD = has_log ? C.descs[gD->snap] : D;
    const TreeRef S = tree_ref(gS), L = tree_ref(gD);
    TopEnt* const out = table + (size_t)inst * G * G;
    TopMM* const omm = table_mm + (size_t)inst * G * G;
    // every 16-square of the node at (r, c), side sd: the walk's state there and the range of the values inside
    auto put_square = [&](uint32_t r, uint32_t c, uint32_t sd, uint32_t bt, uint32_t bs, int64_t mt, int64_t ms, int64_t vmin, int64_t vmax) {
        if (mt != (int32_t)mt || ms != (int32_t)ms || vmin != (int32_t)vmin || vmax != (int32_t)vmax) *overflow = 1;
        const TopEnt e{bt, bs, (int32_t)mt, (int32_t)ms};
        const TopMM m{(int32_t)vmin, (int32_t)vmax};
        const uint32_t n = sd >> 4;
        for (uint32_t i = 0; i < n * n; i++) {
            const uint32_t at = ((r >> 4) + i / n) * G + (c >> 4) + i % n;
            out[at] = e;
            omm[at] = m;
        }
    };
    const bool single_s = !gbm_get(gb, S.T, 0);
    const bool single_t = has_log ? !gbm_get(gb, L.T, 0) : true;
    const int64_t max_s0 = dacd_get(b, SD.mx, 0), max_t0 = has_log ? dacd_get(b, D.mx, 0) : 0;
    const int64_t min_s0 = dacd_get(b, SD.mn, 0), min_t0 = has_log ? dacd_get(b, D.mn, 0) : 0;
    const bool all_one = has_log ? (single_t && (single_s || !gbm_get(gb, L.E, 0))) : single_s;
    if (all_one) {
        if (lane == 0) put_square(0, 0, G * 16, WQ_NONE, WQ_NONE, max_t0, max_s0, max_t0 + max_s0, max_t0 + max_s0);
        return;
    }
    if (lane == 0) {
        q.it[0] = (has_log && !single_t) ? 1u : WQ_NONE;
        q.is[0] = single_s ? WQ_NONE : 1u;
        q.org[0] = 0;
        q.mt[0] = max_t0;
        q.ms[0] = max_s0;
        qmt[0] = min_t0;
        qms[0] = min_s0;
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t lo = 0, hi = 1;
    for (uint32_t side = gD->sidelen; side > 16; side >>= 1) {  // (at most 64 nodes of side 32 at the last step: one pass per level)
        const uint32_t cs = side >> 1, n = lo + (uint32_t)lane;
        const bool live = n < hi;
        Kids kd;
        kd.fill = 0;
        uint32_t po = 0;
        NodeSt p{WQ_NONE, WQ_NONE, 0, 0};
        int64_t pmin_t = 0, pmin_s = 0;
        if (live) {
            p = NodeSt{q.it[n], q.is[n], q.mt[n], q.ms[n]};
            pmin_t = qmt[n];
            pmin_s = qms[n];
            po = q.org[n];
            expand4(gb, S, SD.mx, L, D.mx, p, &kd);
        }
        const uint32_t pushm = live ? (~kd.fill & 15u) : 0u, np = popc32(pushm);
        const uint32_t inc = GpuExecScan::incl(np);
        uint32_t pos = hi + inc - np;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (!live) continue;
            const uint32_t cr = (po >> 16) + (uint32_t)(c >> 1) * cs, cc = (po & 0xffffu) + (uint32_t)(c & 1) * cs;
            // the child's minima, as the reference's search carries them (log.rs:640-676; snapshot.rs:391 without a log)
            const bool has_t = p.bt != WQ_NONE, has_s = p.bs != WQ_NONE;
            const uint32_t it_ = has_t ? p.bt + (uint32_t)c : 0u, is_ = has_s ? p.bs + (uint32_t)c : 0u;
            const bool leaf_t = has_t ? (it_ >= D.T.len || !bmd_get(b, D.T, it_)) : true;
            const bool leaf_s = has_s ? (is_ >= SD.T.len || !bmd_get(b, SD.T, is_)) : true;
            const int64_t mt_ = kd.st[c].mt, ms_ = kd.st[c].ms;
            int64_t min_t_ = has_t ? (leaf_t ? pmin_t : dacd_get(b, D.mn, bmd_rank(b, D.T, it_))) : pmin_t;
            int64_t min_s_ = has_s ? (leaf_s ? pmin_s : pmin_s + dacd_get(b, SD.mn, bmd_rank(b, SD.T, is_))) : pmin_s;
            if (leaf_s) min_s_ = ms_;
            if (leaf_t) {
                min_t_ = mt_;
                if (has_t && it_ < D.T.len && !bmd_get(b, D.E, bmd_rank0(b, D.T, it_ + 1) - 1)) min_t_ = ms_ + mt_ - min_s_;
            }
            const int64_t vmax = ms_ + mt_, vmin = min_s_ + min_t_;
            if ((kd.fill >> c) & 1u) {
                put_square(cr, cc, cs, WQ_NONE, WQ_NONE, mt_, ms_, vmax, vmax);
            } else if (cs == 16) {
                put_square(cr, cc, 16, kd.st[c].bt, kd.st[c].bs, mt_, ms_, vmin, vmax);
            } else {
                q.it[pos] = kd.st[c].bt; q.is[pos] = kd.st[c].bs; q.org[pos] = (cr << 16) | cc; q.mt[pos] = mt_; q.ms[pos] = ms_;
                qmt[pos] = min_t_;
                qms[pos] = min_s_;
                pos++;
            }
        }
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        __builtin_amdgcn_wave_barrier();
        lo = hi;
        hi = cs == 16 ? hi : hi + tot;
    }
}
__global__ void __launch_bounds__(64)
k_top_table(ChunkRef C, TopEnt* __restrict__ table, TopMM* __restrict__ table_mm, uint32_t* __restrict__ overflow) {
    __shared__ WaveQ2 q;
    __shared__ int64_t qmt[WQ2_CAP], qms[WQ2_CAP];  // the frontier nodes' min_t, min_s (log.rs:360-361)
    top_table_inst(C, blockIdx.x, table, table_mm, overflow, q, qmt, qms);
}
// the same for every instant of MANY chunks in one launch (dcdf_chunk_open_batch): workgroup = one (chunk, instant);
// inst_chunk[global instant] = its chunk, first_inst[chunk] = the chunk's first global instant; each chunk's tables are where
// its ChunkRef says; overflow[chunk] != 0 afterwards = a value beyond int32 (that chunk is then walked from the root)
__global__ void __launch_bounds__(64)
k_top_table_batch(const ChunkRef* __restrict__ refs, const uint32_t* __restrict__ inst_chunk, const uint32_t* __restrict__ first_inst,
                  uint32_t* __restrict__ overflow) {
    __shared__ WaveQ2 q;
    __shared__ int64_t qmt[WQ2_CAP], qms[WQ2_CAP];
    const uint32_t ci = inst_chunk[blockIdx.x];
    const ChunkRef C = refs[ci];
    if (C.top_g == 0) return;
    top_table_inst(C, blockIdx.x - first_inst[ci], (TopEnt*)C.top, (TopMM*)C.top_mm, overflow + ci, q, qmt, qms);
}

// ---- opening chunks whose bytes are already in device memory: the parse of chunk.rs:247-266 by one thread per chunk ----
struct OpenMeta {
    uint32_t ok, encoding, fbits, n_blocks, n_inst, k, rows, cols, sidelen, narrow32;
};
// count != 0: only count the instants (descs may be null); else fill descs[first[i] ..] and the per-instant quirk flags
__global__ void __launch_bounds__(64)
k_parse_chunks(const uint8_t* __restrict__ slab, const uint64_t* __restrict__ offs, const uint64_t* __restrict__ lens, uint32_t n,
               const uint32_t* __restrict__ first, InstDesc* __restrict__ descs, uint8_t* __restrict__ quirk, OpenMeta* __restrict__ meta,
               int count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* const b = slab + offs[i];
    Cursor cur{b, (size_t)lens[i]};
    OpenMeta m{};
    m.encoding = cur.u8();
    m.fbits = cur.u8();
    m.n_blocks = cur.u32();
    bool ok = m.encoding == DCDF_I32 || m.encoding == DCDF_I64 || m.encoding == DCDF_F32 || m.encoding == DCDF_F64;
    uint32_t ni = 0;
    InstDesc* const D = count ? nullptr : descs + first[i];
    bool narrow = true;
    for (uint32_t blk = 0; blk < m.n_blocks && cur.ok && ok; blk++) {
        const uint32_t n_inst = cur.u8();  // block.rs:100
        if (n_inst == 0) ok = false;
        const uint32_t snap = ni;
        for (uint32_t j = 0; j < n_inst && cur.ok && ok; j++, ni++) {
            InstDesc d;
            parse_inst(cur, d, j > 0, snap);
            if (!cur.ok) break;
            if (ni == 0) {
                m.k = d.k; m.rows = d.rows; m.cols = d.cols; m.sidelen = d.sidelen;
            } else if (d.k != m.k || d.rows != m.rows || d.cols != m.cols || d.sidelen != m.sidelen) ok = false;
            if (d.T.k != 4 || (d.is_log && d.E.k != 4) || d.rows == 0 || d.cols == 0 || d.sidelen < (d.rows > d.cols ? d.rows : d.cols)) ok = false;
            if (!count && ok) {
                D[ni] = d;
                int64_t hi = dacd_get(b, d.mx, 0), lo = dacd_get(b, d.mn, 0);
                if (d.is_log) {  // log roots are differences against the snapshot's (log.rs:133,148)
                    hi += dacd_get(b, D[snap].mx, 0);
                    lo += dacd_get(b, D[snap].mn, 0);
                }
                const int64_t lim = (int64_t)1 << 30;
                if (hi < -lim || hi >= lim || lo < -lim || lo >= lim) narrow = false;
                quirk[first[i] + ni] = (d.is_log && !bmd_get(b, d.T, 0) && !bmd_get(b, d.E, 0) && bmd_get(b, D[snap].T, 0)) ? 1 : 0;
            }
        }
    }
    m.ok = (ok && cur.ok && ni > 0 && cur.pos == lens[i]) ? 1u : 0u;
    m.n_inst = ni;
    m.narrow32 = narrow ? 1u : 0u;
    meta[i] = m;
}
// Structural validation of the instants k_parse_chunks described: the checks dcdf_chunk_open makes on the host (every count a
// decoder relies on against the bitmaps' popcounts, the rank index of every BitMap), by one wave per instant.  bad[chunk] != 0
// afterwards: the stream is not a chunk of this format; its handle is refused before any walk chases its indices.
__device__ uint64_t wave_ones(const uint8_t* b, const BmDesc& d, bool& index_ok) {  // popcount of the bitmap (all lanes get it)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t W = (d.len + 31) / 32, nidx = d.len / 128;  // bitmap.rs:70
    uint32_t base = 0;
    for (uint32_t g0 = 0; 4 * g0 < W; g0 += 64) {  // a lane per group of four words = per entry of the rank index
        const uint32_t g = g0 + lane;
        uint32_t cnt = 0;
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
            const uint32_t w = 4 * g + i;
            if (w < W) {
                uint32_t x = load_be32(b + d.words_off + 4 * w);
                const uint32_t left = d.len - 32 * w;
                if (left < 32) x &= ~(0xffffffffu >> left);  // padding bits do not count
                cnt += popc32(x);
            }
        }
        const uint32_t inc = GpuExecScan::incl(cnt);
        if (d.k == 4 && g < nidx && load_be32(b + d.idx_off + 4 * g) != base + inc) index_ok = false;  // bitmap.rs:97-104
        base += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
    return base;
}
__device__ bool wave_dac_ok(const uint8_t* b, const DacDesc& d, uint64_t expect_len, bool& index_ok) {
    if (expect_len == 0) return d.nlev == 0;
    if (d.nlev == 0 || d.nlev > 8 || d.bm[0].len != expect_len) return false;
    for (uint32_t l = 0; l < d.nlev; l++) {
        if (d.bm[l].k != 4) return false;
        const uint64_t next = wave_ones(b, d.bm[l], index_ok);
        if (l + 1 < d.nlev ? d.bm[l + 1].len != next : next != 0) return false;  // dac.rs:83-90: every hop lands in the next plane
    }
    return true;
}
__global__ void __launch_bounds__(64)
k_validate_insts(const uint8_t* __restrict__ slab, const uint64_t* __restrict__ offs, const uint32_t* __restrict__ inst_chunk,
                 const InstDesc* __restrict__ descs, uint32_t* __restrict__ bad) {
    const uint32_t ci = inst_chunk[blockIdx.x];
    const uint8_t* const b = slab + offs[ci];
    const InstDesc& d = descs[blockIdx.x];
    bool index_ok = true, ok = true;
    if (d.k < 2 || d.k > 255 || d.rows == 0 || d.cols == 0 || d.T.k != 4 || (d.is_log && d.E.k != 4)) ok = false;
    if (ok) {
        const uint64_t internal = wave_ones(b, d.T, index_ok);
        const uint64_t visited = 1 + (uint64_t)d.k * d.k * internal;  // snapshot.rs:177: k^2 children per internal node
        if (d.T.len > visited) ok = false;
        if (ok && (!wave_dac_ok(b, d.mx, visited, index_ok) || !wave_dac_ok(b, d.mn, internal, index_ok))) ok = false;
        if (ok && d.is_log) {
            if (d.E.len != d.T.len - internal) ok = false;  // one eqB bit per T = 0 (log.rs:137-144)
            else (void)wave_ones(b, d.E, index_ok);
        }
    }
    if (__ballot(!ok || !index_ok) != 0 && (threadIdx.x & 63u) == 0) atomicOr(&bad[ci], 1u);
}
// counts of the (query, instant) items the wave walk marked: one thread each over the bitmaps of the item's pieces
__global__ void __launch_bounds__(64)
k_search_count(const uint32_t* __restrict__ wbits, const SearchItem* __restrict__ items, const WinQuery* __restrict__ qs, uint32_t n,
               uint32_t* __restrict__ counts) {
    const uint32_t it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= n) return;
    const SearchItem I = items[it];
    if (I.w0 == SI_FLAT) return;  // (counted by k_search_cells)
    const WinQuery Q = qs[I.query];
    const uint32_t nrb = (Q.bottom - Q.top + 63u) >> 6;
    const uint4* w = (const uint4*)(wbits + (uint64_t)I.w0 * 128u);
    uint32_t cnt = 0;
    for (uint32_t i = 0; i < nrb * I.ncb * 32u; i++) {
        const uint4 x = w[i];
        cnt += popc32(x.x) + popc32(x.y) + popc32(x.z) + popc32(x.w);
    }
    counts[it] = cnt;
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

// the same with explicit output positions (fill_cell batches: query q's series goes to out + out_offset[q])
__global__ void __launch_bounds__(256)
k_get_at(const ChunkRef* __restrict__ chunks, const PointQuery* __restrict__ qs, uint32_t nq, int64_t* __restrict__ out,
         const uint64_t* __restrict__ at) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const PointQuery Q = qs[q];
    const ChunkRef C = chunks[Q.chunk];
    out[at[q]] = inst_get(C.bytes, C.descs, Q.instant, Q.row, Q.col);
}
// encoded chunks, wherever they lie in device memory, into one slab (16-byte aligned starts)
struct SlabItem {
    const uint8_t* src;
    uint64_t len, dst_off;
};
__global__ void __launch_bounds__(256) k_slab_pack(const SlabItem* __restrict__ items, uint32_t n, uint8_t* __restrict__ dst) {
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const SlabItem it = items[i];
        if (((uintptr_t)it.src & 15u) == 0) {
            const uint4* s4 = (const uint4*)it.src;
            uint4* d4 = (uint4*)(dst + it.dst_off);
            const uint64_t nv = it.len / 16;
            for (uint64_t v = threadIdx.x; v < nv; v += blockDim.x) d4[v] = s4[v];
            for (uint64_t b = 16 * nv + threadIdx.x; b < it.len; b += blockDim.x) dst[it.dst_off + b] = it.src[b];
        } else {
            for (uint64_t b = threadIdx.x; b < it.len; b += blockDim.x) dst[it.dst_off + b] = it.src[b];
        }
    }
}

// search pass 1 for chunks the wave walk does not take (k != 2): decode and test.  One WAVE per (query, instant) item; the
// lanes stride over the window's cells, each a root-to-leaf descent (inst_get), and mark the cells in range in the item's flat
// window bitmap.  The reference's search is a pruned descent whose result is exactly that set (its bounds are true bounds) --
// except for the single-node-uniform-log shape of SearchExtra, which is evaluated here as the data it is: all cells, no cell, or
// the cells with lower <= s(cell) + (c - max_s(root)) <= upper.
__global__ void __launch_bounds__(64)
k_search_cells(const ChunkRef* __restrict__ chunks, const WinQuery* __restrict__ qs, const SearchItem* __restrict__ items,
               uint32_t n_items, const uint8_t* __restrict__ quirk, uint32_t* __restrict__ bits, uint32_t* __restrict__ counts) {
    const uint32_t it = blockIdx.x;
    if (it >= n_items) return;
    const SearchItem I = items[it];
    if (I.w0 != SI_FLAT) return;  // (marked by the wave walk)
    const int lane = threadIdx.x;
    const WinQuery Q = qs[I.query];
    const ChunkRef C = chunks[Q.chunk];
    const uint8_t* const b = C.bytes;
    const uint32_t wc = Q.right - Q.left, ncell = (Q.bottom - Q.top) * wc;
    uint32_t* const bw = bits + I.bits_off;
    const InstDesc& D = C.descs[I.instant];
    bool all = false, none = false;
    int64_t shift = 0;
    uint32_t from = I.instant;
    if (quirk[it]) {  // (log.rs:527-586 on this shape)
        const InstDesc& S = C.descs[D.snap];
        const int64_t max_s0 = dacd_get(b, S.mx, 0), min_s0 = dacd_get(b, S.mn, 0), c1 = dacd_get(b, D.mx, 0) + max_s0;
        all = min_s0 >= Q.lower && c1 <= Q.upper;
        none = !all && (min_s0 > Q.upper || c1 < Q.lower);
        shift = c1 - max_s0;
        from = D.snap;
    }
    uint32_t cnt = 0;
    if (!none) {
        for (uint32_t e = (uint32_t)lane; e < ncell; e += 64) {
            bool hit = all;
            if (!all) {
                const int64_t v = inst_get(b, C.descs, from, Q.top + e / wc, Q.left + e % wc) + shift;
                hit = Q.lower <= v && v <= Q.upper;
            }
            if (hit) {
                atomicOr(&bw[e >> 5], 1u << (e & 31u));
                cnt++;
            }
        }
    }
    // wave total (DPP-free: a shuffle ladder is fine here)
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    if (lane == 0) counts[it] = cnt;
}
// search pass 2: expand the bitmaps into sorted (instant,row,col) triples
__global__ void __launch_bounds__(64)
k_search_emit(const WinQuery* __restrict__ qs, const SearchItem* __restrict__ items, uint32_t n_items,
              const uint32_t* __restrict__ bits, const uint32_t* __restrict__ wbits, const uint64_t* __restrict__ offs,
              uint32_t* __restrict__ out) {
    const uint32_t it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= n_items) return;
    const SearchItem I = items[it];
    const WinQuery Q = qs[I.query];
    const uint32_t wc = Q.right - Q.left, nbits = (Q.bottom - Q.top) * wc;
    const uint32_t* bw = bits + I.bits_off;
    uint32_t* o = out + 3 * offs[it];
    // origin of the chunk inside its raster (dcdf_raster_search_batch; zero for chunk-level searches): a search does not use
    // out_off / _pad otherwise
    const uint32_t ot = Q._pad, orow = (uint32_t)Q.out_off, ocol = (uint32_t)(Q.out_off >> 32);
    if (I.w0 != SI_FLAT) {  // the pieces' bitmaps (64 rows x 2 words) of the wave walk: rows in order, pieces left to right
        for (uint32_t r = Q.top; r < Q.bottom; r++) {
            const uint32_t rb = (r - Q.top) >> 6, rr = (r - Q.top) & 63u;
            for (uint32_t cw = 0; cw < 2u * I.ncb; cw++) {
                uint32_t x = wbits[((uint64_t)I.w0 + rb * I.ncb + (cw >> 1)) * 128u + 2u * rr + (cw & 1u)];
                while (x) {
                    const uint32_t j = (uint32_t)__builtin_ctz(x);
                    x &= x - 1;
                    o[0] = ot + I.instant;
                    o[1] = orow + r;
                    o[2] = ocol + Q.left + 32u * cw + j;
                    o += 3;
                }
            }
        }
        return;
    }
    for (uint32_t w = 0; w < (nbits + 31) / 32; w++) {
        uint32_t x = bw[w];
        while (x) {
            const uint32_t p = 32 * w + (uint32_t)__builtin_ctz(x);
            x &= x - 1;
            o[0] = ot + I.instant;
            o[1] = orow + Q.top + p / wc;
            o[2] = ocol + Q.left + p % wc;
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
    // Structural validation (chunks arrive by CID from an untrusted store; the reference would panic on a malformed one, the
    // GPU must not chase out-of-range indices): every count the decoders rely on is checked against the bitmaps' popcounts.
    {
        bool index_ok = true;
        auto ones = [&](const BmDesc& d) -> uint64_t {  // popcount of the bitmap; checks its rank index on the way (bitmap.rs:97-104)
            uint64_t n = 0;
            for (uint32_t w = 0; w < (d.len + 31) / 32; w++) {
                uint32_t x = load_be32(bytes + d.words_off + 4 * w);
                const uint32_t left = d.len - 32 * w;
                if (left < 32) x &= ~(0xffffffffu >> left);  // padding bits do not count
                n += (uint64_t)__builtin_popcount(x);
                if (d.k == 4 && (w & 3) == 3 && (w >> 2) < d.len / 128 && load_be32(bytes + d.idx_off + 4 * (w >> 2)) != (uint32_t)n) index_ok = false;
            }
            return n;
        };
        auto dac_ok = [&](const DacDesc& d, uint64_t expect_len) -> bool {
            if (expect_len == 0) return d.nlev == 0;
            if (d.nlev == 0 || d.bm[0].len != expect_len) return false;
            for (uint32_t l = 0; l < d.nlev; l++) {
                if (d.bm[l].k != 4) return false;
                const uint64_t next = ones(d.bm[l]);
                if (l + 1 < d.nlev ? d.bm[l + 1].len != next : next != 0) return false;  // dac.rs:83-90: every hop lands in the next plane
            }
            return true;
        };
        for (const InstDesc& d : c->descs) {
            if (d.k < 2 || d.k > 255 || d.rows == 0 || d.cols == 0) return DCDF_ERR_FORMAT;
            // snapshot.rs:118-119: k^ceil(ln(max)/ln(k)) in f64 -- what the reference writes (625 for a 125-wide tile with k = 5)
            if (d.sidelen != ref_sidelen(std::max(d.rows, d.cols), d.k)) return DCDF_ERR_FORMAT;
            if (d.T.k != 4 || (d.is_log && d.E.k != 4)) return DCDF_ERR_FORMAT;  // bitmap.rs:69,130
            const uint64_t internal = ones(d.T);
            const uint64_t visited = 1 + (uint64_t)d.k * d.k * internal;          // snapshot.rs:177: four children per internal node
            if (d.T.len > visited) return DCDF_ERR_FORMAT;
            if (!dac_ok(d.mx, visited) || !dac_ok(d.mn, internal)) return DCDF_ERR_FORMAT;
            if (d.is_log && d.E.len != d.T.len - internal) return DCDF_ERR_FORMAT;  // one eqB bit per T = 0 (log.rs:137-144)
            if (d.is_log) (void)ones(d.E);
            if (!index_ok) return DCDF_ERR_FORMAT;
        }
    }
    c->instants = (uint32_t)c->descs.size();
    c->k0 = c->descs[0].k;
    c->sidelen0 = c->descs[0].sidelen;
    c->rows = c->descs[0].rows;  // chunk.rs:119-123
    c->cols = c->descs[0].cols;
    for (const InstDesc& d : c->descs)
        if (d.rows != c->rows || d.cols != c->cols || d.k != c->descs[0].k || d.sidelen != c->descs[0].sidelen ||
            d.sidelen < std::max(d.rows, d.cols))
            return DCDF_ERR_FORMAT;
    c->len = len;
    c->narrow32 = true;
    for (size_t i = 0; i < c->descs.size(); i++) {
        const InstDesc& D = c->descs[i];
        int64_t hi = dacd_get(bytes, D.mx, 0), lo = dacd_get(bytes, D.mn, 0);
        if (D.is_log) {  // log roots are differences against the snapshot's (log.rs:133,148)
            hi += dacd_get(bytes, c->descs[D.snap].mx, 0);
            lo += dacd_get(bytes, c->descs[D.snap].mn, 0);
        }
        const int64_t lim = (int64_t)1 << 30;
        if (hi < -lim || hi >= lim || lo < -lim || lo >= lim) c->narrow32 = false;
    }
    c->search_quirk.assign(c->descs.size(), 0);
    for (size_t i = 0; i < c->descs.size(); i++) {
        const InstDesc& L = c->descs[i];
        if (L.is_log && !bmd_get(bytes, L.T, 0) && !bmd_get(bytes, L.E, 0) && bmd_get(bytes, c->descs[L.snap].T, 0)) c->search_quirk[i] = 1;
    }
    K2R_HIP(c->d_bytes.alloc(len + 64));  // (slack: the wave decoder reads whole 16-byte blocks / 4-byte groups at the tail)
    K2R_HIP(hipMemcpy(c->d_bytes.p, bytes, len, hipMemcpyHostToDevice));
    K2R_HIP(c->d_descs.alloc(c->descs.size() * sizeof(InstDesc)));
    K2R_HIP(hipMemcpy(c->d_descs.p, c->descs.data(), c->descs.size() * sizeof(InstDesc), hipMemcpyHostToDevice));
    if (c->descs[0].k == 2 && c->descs[0].sidelen >= 32 && c->descs[0].sidelen <= 256 && !std::getenv("K2R_NO_TOP_TABLE")) {
        const uint32_t g = c->descs[0].sidelen / 16;
        const size_t tbytes = (size_t)c->instants * g * g * sizeof(TopEnt);
        K2R_HIP(c->d_top_mm.alloc((size_t)c->instants * g * g * sizeof(TopMM)));
        K2R_HIP(c->d_top.alloc(tbytes + 4));  // (+ the "a value does not fit int32" word)
        uint32_t* const d_ovf = (uint32_t*)(c->d_top.as<uint8_t>() + tbytes);
        K2R_HIP(hipMemset(d_ovf, 0, 4));
        ChunkRef ref{c->d_bytes.as<uint8_t>(), c->d_descs.as<InstDesc>(), c->instants, c->rows, c->cols, c->fbits, nullptr, nullptr, g, 0};
        hipLaunchKernelGGL(k_top_table, dim3(c->instants), dim3(64), 0, 0, ref, c->d_top.as<TopEnt>(), c->d_top_mm.as<TopMM>(), d_ovf);
        K2R_HIP(hipGetLastError());
        uint32_t ovf = 0;
        K2R_HIP(hipMemcpy(&ovf, d_ovf, 4, hipMemcpyDeviceToHost));
        if (!ovf) c->top_g = g;
    }
    c->p_bytes = c->d_bytes.as<uint8_t>();
    c->p_descs = c->d_descs.as<InstDesc>();
    c->p_top = c->d_top.p;
    c->p_top_mm = c->d_top_mm.p;
    *h = c.release();
    return DCDF_OK;
}
extern "C" void dcdf_chunk_close(dcdf_chunk* h) { delete h; }

namespace {
struct BatchSlab {  // what the chunks of one dcdf_chunk_open_batch share
    DevBuf bytes, descs, top, top_mm, refs;
};
}  // namespace

// Many chunks at once.  mem = DCDF_MEM_HOST: dcdf_chunk_open one by one (full structural validation).  mem = DCDF_MEM_DEVICE:
// the bytes are where an encoder session left them (dcdf_encoder_result's device pointers) -- they are packed into one slab,
// parsed ON the device (k_parse_chunks, one thread per chunk: bounds-checked walk of the same layout, no host copy of the
// bytes), and the side-16 tables of all their instants are built by ONE launch (k_top_table_batch, a wave per instant).
// Device input gets the same structural validation as dcdf_chunk_open's (k_validate_insts: popcounts against the Dac and eqB
// lengths, rank indexes), a wave per instant.  status (may be NULL) receives one code per chunk; out[i] is NULL where it is not 0.
extern "C" int dcdf_chunk_open_batch(const uint8_t* const* bytes, const uint64_t* lens, size_t n, int mem, dcdf_chunk** out,
                                     int32_t* status) {
    if (!bytes || !lens || !out || n == 0 || n > 0x7fffffffu || (mem != DCDF_MEM_HOST && mem != DCDF_MEM_DEVICE)) return DCDF_ERR_BAD_ARG;
    if (!Runtime::get().ok) return DCDF_ERR_NO_DEVICE;
    for (size_t i = 0; i < n; i++) out[i] = nullptr;
    if (mem == DCDF_MEM_HOST) {
        int first_err = DCDF_OK;
        for (size_t i = 0; i < n; i++) {
            const int rc = dcdf_chunk_open(bytes[i], (size_t)lens[i], &out[i]);
            if (status) status[i] = rc;
            if (rc != DCDF_OK && first_err == DCDF_OK) first_err = rc;
        }
        return status ? DCDF_OK : first_err;
    }
    struct OpenTimer {  // K2R_OPEN_TIMING=1: wall time of the steps on stderr (diagnostics; the laps synchronise the device)
        const bool on = std::getenv("K2R_OPEN_TIMING") != nullptr;
        std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
        void lap(const char* what) {
            if (!on) return;
            (void)hipDeviceSynchronize();
            const auto u = std::chrono::steady_clock::now();
            std::fprintf(stderr, "k2r-open %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(u - t).count());
            t = u;
        }
    } tm;
    auto slab = std::make_shared<BatchSlab>();
    std::vector<SlabItem> items(n);
    std::vector<uint64_t> offs(n);
    uint64_t tot = 0;
    for (size_t i = 0; i < n; i++) {
        if (!bytes[i] || lens[i] < 6 || lens[i] > 0xfffffff0ull) return DCDF_ERR_BAD_ARG;
        offs[i] = tot;
        items[i] = SlabItem{bytes[i], lens[i], tot};
        tot += (lens[i] + 64 + 15) & ~15ull;  // (slack: the wave decoder reads whole 16-byte blocks at a stream's tail)
    }
    DevBuf d_items, d_offs, d_lens, d_first, d_meta, d_quirk, d_ic, d_ovf;
    K2R_HIP(slab->bytes.alloc_pooled(tot));  // (gigabytes: hipMalloc takes 4 .. 60 ms for them, depending on what the driver has to clear)
    K2R_HIP(d_items.alloc(n * sizeof(SlabItem)));
    K2R_HIP(hipMemcpy(d_items.p, items.data(), n * sizeof(SlabItem), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_slab_pack, dim3((uint32_t)std::min<size_t>(n, 8192)), dim3(256), 0, 0, d_items.as<SlabItem>(), (uint32_t)n,
                       slab->bytes.as<uint8_t>());
    K2R_HIP(hipGetLastError());
    tm.lap("slab alloc + pack");
    K2R_HIP(d_offs.alloc(n * 8));
    K2R_HIP(d_lens.alloc(n * 8));
    K2R_HIP(hipMemcpy(d_offs.p, offs.data(), n * 8, hipMemcpyHostToDevice));
    K2R_HIP(hipMemcpy(d_lens.p, lens, n * 8, hipMemcpyHostToDevice));
    K2R_HIP(d_meta.alloc(n * sizeof(OpenMeta)));
    const uint32_t pgrid = (uint32_t)((n + 63) / 64);
    // pass 1: instants per chunk
    hipLaunchKernelGGL(k_parse_chunks, dim3(pgrid), dim3(64), 0, 0, slab->bytes.as<uint8_t>(), d_offs.as<uint64_t>(), d_lens.as<uint64_t>(),
                       (uint32_t)n, (const uint32_t*)nullptr, (InstDesc*)nullptr, (uint8_t*)nullptr, d_meta.as<OpenMeta>(), 1);
    K2R_HIP(hipGetLastError());
    std::vector<OpenMeta> meta(n);
    K2R_HIP(hipMemcpy(meta.data(), d_meta.p, n * sizeof(OpenMeta), hipMemcpyDeviceToHost));
    tm.lap("parse pass 1");
    std::vector<uint32_t> first(n + 1, 0);
    for (size_t i = 0; i < n; i++) first[i + 1] = first[i] + (meta[i].ok ? meta[i].n_inst : 0u);
    const uint32_t total_inst = first[n];
    if (total_inst == 0) {
        if (status) for (size_t i = 0; i < n; i++) status[i] = DCDF_ERR_FORMAT;
        return status ? DCDF_OK : DCDF_ERR_FORMAT;
    }
    // pass 2: the descriptors (chunks that failed pass 1 get a zero-length stream: parsed as malformed again, nothing stored)
    std::vector<uint64_t> lens2(lens, lens + n);
    for (size_t i = 0; i < n; i++)
        if (!meta[i].ok) lens2[i] = 0;
    K2R_HIP(hipMemcpy(d_lens.p, lens2.data(), n * 8, hipMemcpyHostToDevice));
    K2R_HIP(d_first.alloc((n + 1) * 4));
    K2R_HIP(hipMemcpy(d_first.p, first.data(), (n + 1) * 4, hipMemcpyHostToDevice));
    K2R_HIP(slab->descs.alloc((size_t)total_inst * sizeof(InstDesc)));
    K2R_HIP(d_quirk.alloc(total_inst));
    hipLaunchKernelGGL(k_parse_chunks, dim3(pgrid), dim3(64), 0, 0, slab->bytes.as<uint8_t>(), d_offs.as<uint64_t>(), d_lens.as<uint64_t>(),
                       (uint32_t)n, d_first.as<uint32_t>(), slab->descs.as<InstDesc>(), d_quirk.as<uint8_t>(), d_meta.as<OpenMeta>(), 0);
    K2R_HIP(hipGetLastError());
    std::vector<OpenMeta> meta2(n);
    K2R_HIP(hipMemcpy(meta2.data(), d_meta.p, n * sizeof(OpenMeta), hipMemcpyDeviceToHost));
    std::vector<uint8_t> quirk(total_inst);
    K2R_HIP(hipMemcpy(quirk.data(), d_quirk.p, total_inst, hipMemcpyDeviceToHost));
    tm.lap("parse pass 2 + descs D2H");
    // structural validation of every instant (the host entry point's checks, a wave per instant), before anything walks them
    std::vector<uint32_t> inst_chunk(total_inst);
    for (size_t i = 0; i < n; i++)
        for (uint32_t j = first[i]; j < first[i + 1]; j++) inst_chunk[j] = (uint32_t)i;
    K2R_HIP(d_ic.alloc((size_t)total_inst * 4));
    K2R_HIP(hipMemcpy(d_ic.p, inst_chunk.data(), (size_t)total_inst * 4, hipMemcpyHostToDevice));
    {
        DevBuf d_bad;
        K2R_HIP(d_bad.alloc(n * 4));
        K2R_HIP(hipMemset(d_bad.p, 0, n * 4));
        hipLaunchKernelGGL(k_validate_insts, dim3(total_inst), dim3(64), 0, 0, slab->bytes.as<uint8_t>(), d_offs.as<uint64_t>(), d_ic.as<uint32_t>(),
                           slab->descs.as<InstDesc>(), d_bad.as<uint32_t>());
        K2R_HIP(hipGetLastError());
        std::vector<uint32_t> bad(n);
        K2R_HIP(hipMemcpy(bad.data(), d_bad.p, n * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; i++)
            if (bad[i]) meta2[i].ok = 0;
    }
    tm.lap("validation");
    // side-16 tables for the k = 2 chunks of sidelen 32..256: one slab, one launch
    const bool want_top = !std::getenv("K2R_NO_TOP_TABLE");
    std::vector<uint64_t> top_off(n, 0);
    std::vector<uint32_t> top_g(n, 0);
    uint64_t squares = 0;
    for (size_t i = 0; i < n; i++) {
        if (!meta[i].ok || !meta2[i].ok) continue;
        // the depth the reference computes (k2r_runtime.h ref_sidelen): a stream that disagrees is not a chunk of this format
        if (meta[i].sidelen != ref_sidelen(std::max(meta[i].rows, meta[i].cols), meta[i].k)) {
            meta2[i].ok = 0;
            continue;
        }
        if (want_top && meta[i].k == 2 && meta[i].sidelen >= 32 && meta[i].sidelen <= 256) {
            top_g[i] = meta[i].sidelen / 16;
            top_off[i] = squares;
            squares += (uint64_t)meta[i].n_inst * top_g[i] * top_g[i];
        }
    }
    std::vector<ChunkRef> refs(n);
    if (squares) {
        K2R_HIP(slab->top.alloc(squares * sizeof(TopEnt)));
        K2R_HIP(slab->top_mm.alloc(squares * sizeof(TopMM)));
    }
    for (size_t i = 0; i < n; i++) {
        const bool ok = meta[i].ok && meta2[i].ok;
        refs[i] = ChunkRef{slab->bytes.as<uint8_t>() + offs[i], slab->descs.as<InstDesc>() + first[i], ok ? meta[i].n_inst : 0u, meta[i].rows,
                           meta[i].cols, meta[i].fbits, top_g[i] ? slab->top.as<TopEnt>() + top_off[i] : nullptr,
                           top_g[i] ? slab->top_mm.as<TopMM>() + top_off[i] : nullptr, ok ? top_g[i] : 0u, 0};
    }
    K2R_HIP(slab->refs.alloc(n * sizeof(ChunkRef)));
    K2R_HIP(hipMemcpy(slab->refs.p, refs.data(), n * sizeof(ChunkRef), hipMemcpyHostToDevice));
    std::vector<uint32_t> ovf(n, 0);
    if (squares) {
        K2R_HIP(d_ovf.alloc(n * 4));
        K2R_HIP(hipMemset(d_ovf.p, 0, n * 4));
        hipLaunchKernelGGL(k_top_table_batch, dim3(total_inst), dim3(64), 0, 0, slab->refs.as<ChunkRef>(), d_ic.as<uint32_t>(), d_first.as<uint32_t>(),
                           d_ovf.as<uint32_t>());
        K2R_HIP(hipGetLastError());
        K2R_HIP(hipMemcpy(ovf.data(), d_ovf.p, n * 4, hipMemcpyDeviceToHost));
    }
    tm.lap("top tables");
    K2R_HIP(hipDeviceSynchronize());
    int first_err = DCDF_OK;
    for (size_t i = 0; i < n; i++) {
        const bool ok = meta[i].ok && meta2[i].ok;
        if (status) status[i] = ok ? DCDF_OK : DCDF_ERR_FORMAT;
        if (!ok) {
            if (first_err == DCDF_OK) first_err = DCDF_ERR_FORMAT;
            continue;
        }
        std::unique_ptr<dcdf_chunk> c(new (std::nothrow) dcdf_chunk());
        if (!c) {
            for (size_t j = 0; j < i; j++) {
                delete out[j];
                out[j] = nullptr;
            }
            return DCDF_ERR_NOMEM;
        }
        c->instants = meta[i].n_inst;
        c->k0 = meta[i].k;
        c->sidelen0 = meta[i].sidelen;
        c->rows = meta[i].rows;
        c->cols = meta[i].cols;
        c->n_blocks = meta[i].n_blocks;
        c->encoding = (int32_t)meta[i].encoding;
        c->fbits = meta[i].fbits;
        c->len = (size_t)lens[i];
        c->narrow32 = meta2[i].narrow32 != 0;
        c->search_quirk.assign(quirk.begin() + first[i], quirk.begin() + first[i + 1]);
        c->top_g = (top_g[i] && !ovf[i]) ? top_g[i] : 0;
        c->p_bytes = refs[i].bytes;
        c->p_descs = refs[i].descs;
        c->p_top = refs[i].top;
        c->p_top_mm = refs[i].top_mm;
        c->store = slab;
        out[i] = c.release();
    }
    tm.lap("handles");
    return status ? DCDF_OK : first_err;
}
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

// Byte range of every instant's Snapshot / Log inside the chunk (off[i] .. off[i + 1]; off has instants + 1 entries) and
// the instant of the snapshot its block starts with: what a decode of instant i can touch at most (decode path:
// "encoded bytes of the touched chunks' touched structures").  Host metadata only.
extern "C" int dcdf_chunk_instant_layout(const dcdf_chunk* h, uint64_t* off, uint32_t* snapshot_of) {
    if (!h || !off) return DCDF_ERR_BAD_ARG;
    const std::vector<InstDesc>& descs = host_descs(h);
    if (descs.size() != h->instants) return DCDF_ERR_INTERNAL;
    for (uint32_t i = 0; i < h->instants; i++) {
        off[i] = (uint64_t)descs[i].T.idx_off - 8 - 13;  // BitMap header (len, k) and the 13-byte instant header before it
        if (snapshot_of) snapshot_of[i] = descs[i].snap;
    }
    off[h->instants] = h->len;
    return DCDF_OK;
}

static ChunkRef make_ref(const dcdf_chunk* h) {
    return ChunkRef{h->p_bytes, h->p_descs, h->instants, h->rows, h->cols, h->fbits,
                    h->top_g ? (const TopEnt*)h->p_top : nullptr, h->top_g ? (const TopMM*)h->p_top_mm : nullptr, h->top_g, 0};
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
// A host thread's page of pinned, device-visible memory: single get / short fill_cell calls put their queries there and the
// kernel writes the answers there -- no allocation and no explicit copy per call (the round-1 form did three hipMalloc, two
// H2D copies and one D2H copy per cell).
namespace {
struct PinnedPage {
    static constexpr size_t kPoints = 1024;
    ChunkRef* ref = nullptr;  // [1]
    PointQuery* q = nullptr;  // [kPoints]
    int64_t* out = nullptr;   // [kPoints]
    void* base = nullptr;
    bool init() {
        if (base) return true;
        const size_t bytes = 256 + kPoints * (sizeof(PointQuery) + 8);
        if (hipHostMalloc(&base, bytes, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            base = nullptr;
            return false;
        }
        ref = (ChunkRef*)base;
        q = (PointQuery*)((uint8_t*)base + 256);
        out = (int64_t*)((uint8_t*)base + 256 + kPoints * sizeof(PointQuery));
        return true;
    }
    // (never freed: a thread_local destructor of the main thread would run after the HIP runtime has shut down; one page per
    // host thread that ever asked a single point lives as long as the process)
};
thread_local PinnedPage g_page;
}  // namespace

static int run_points(const dcdf_chunk* h, const std::vector<PointQuery>& pq, int64_t* out) {
    const uint32_t n = (uint32_t)pq.size();
    if (n <= PinnedPage::kPoints && g_page.init()) {
        *g_page.ref = make_ref(h);
        std::memcpy(g_page.q, pq.data(), n * sizeof(PointQuery));
        hipLaunchKernelGGL(k_get, dim3((n + 255) / 256), dim3(256), 0, 0, g_page.ref, g_page.q, n, g_page.out);
        K2R_HIP(hipGetLastError());
        K2R_HIP(hipStreamSynchronize(0));
        std::memcpy(out, g_page.out, n * 8ull);
        return DCDF_OK;
    }
    const ChunkRef ref = make_ref(h);
    DevBuf d_ref, d_q, d_o;
    K2R_HIP(d_ref.alloc(sizeof(ref)));
    K2R_HIP(hipMemcpy(d_ref.p, &ref, sizeof(ref), hipMemcpyHostToDevice));
    K2R_HIP(d_q.alloc(pq.size() * sizeof(PointQuery)));
    K2R_HIP(hipMemcpy(d_q.p, pq.data(), pq.size() * sizeof(PointQuery), hipMemcpyHostToDevice));
    K2R_HIP(d_o.alloc(pq.size() * 8));
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

static void dedup_chunks(dcdf_chunk* const* chunks, size_t nq, std::vector<uint32_t>& idx, std::vector<const dcdf_chunk*>& uniq);
static int upload_refs(dcdf_chunk* const* chunks, const std::vector<uint32_t>& uniq_of_query, size_t nuniq,
                       const std::vector<const dcdf_chunk*>& uniq, DevBuf& d_refs);
// many (chunk, point) pairs -- or (chunk, cell series) -- in ONE launch (Superchunk::get / fill_cell route many points,
// superchunk.rs:313-400); at[] = output position of each point, or null for "in order"
static int run_points_multi(dcdf_chunk* const* chunks, size_t nchunks_q, const std::vector<uint32_t>& chunk_of_point,
                            std::vector<PointQuery>& pq, const std::vector<uint64_t>* at, uint64_t out_elems, int64_t* out, int out_mem,
                            float* kernel_ms) {
    std::vector<uint32_t> cidx;
    std::vector<const dcdf_chunk*> uniq;
    dedup_chunks(chunks, nchunks_q, cidx, uniq);
    for (size_t i = 0; i < pq.size(); i++) pq[i].chunk = cidx[chunk_of_point[i]];
    DevBuf d_refs, d_q, d_at, d_o;
    int rc = upload_refs(chunks, cidx, uniq.size(), uniq, d_refs);
    if (rc != DCDF_OK) return rc;
    const uint32_t n = (uint32_t)pq.size();
    K2R_HIP(d_q.alloc(pq.size() * sizeof(PointQuery)));
    K2R_HIP(hipMemcpy(d_q.p, pq.data(), pq.size() * sizeof(PointQuery), hipMemcpyHostToDevice));
    int64_t* dst = out;
    if (out_mem == DCDF_MEM_HOST) {
        K2R_HIP(d_o.alloc(out_elems * 8));
        dst = d_o.as<int64_t>();
    }
    if (at) {
        K2R_HIP(d_at.alloc(at->size() * 8));
        K2R_HIP(hipMemcpy(d_at.p, at->data(), at->size() * 8, hipMemcpyHostToDevice));
    }
    EventPair ev;
    K2R_HIP(ev.create());
    K2R_HIP(hipEventRecord(ev.e0, 0));
    if (at) hipLaunchKernelGGL(k_get_at, dim3((n + 255) / 256), dim3(256), 0, 0, d_refs.as<ChunkRef>(), d_q.as<PointQuery>(), n, dst, d_at.as<uint64_t>());
    else hipLaunchKernelGGL(k_get, dim3((n + 255) / 256), dim3(256), 0, 0, d_refs.as<ChunkRef>(), d_q.as<PointQuery>(), n, dst);
    K2R_HIP(hipEventRecord(ev.e1, 0));
    K2R_HIP(hipGetLastError());
    if (out_mem == DCDF_MEM_HOST) K2R_HIP(hipMemcpy(out, d_o.p, out_elems * 8, hipMemcpyDeviceToHost));
    else K2R_HIP(hipDeviceSynchronize());
    float ms = 0.f;
    K2R_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
    if (kernel_ms) *kernel_ms = ms;
    return DCDF_OK;
}
extern "C" int dcdf_query_get_batch(dcdf_chunk* const* chunks, const uint32_t* points, size_t n, int64_t* out, int out_mem,
                                    float* kernel_ms) {
    if (!chunks || !points || !out || n == 0 || n > 0x7fffffffu || (out_mem != DCDF_MEM_HOST && out_mem != DCDF_MEM_DEVICE)) return DCDF_ERR_BAD_ARG;
    if (!Runtime::get().ok) return DCDF_ERR_NO_DEVICE;
    std::vector<PointQuery> pq(n);
    std::vector<uint32_t> cop(n);
    for (size_t i = 0; i < n; i++) {
        const dcdf_chunk* h = chunks[i];
        if (!h) return DCDF_ERR_BAD_ARG;
        const uint32_t t = points[3 * i], r = points[3 * i + 1], c = points[3 * i + 2];
        if (t >= h->instants || r >= h->rows || c >= h->cols) return DCDF_ERR_BOUNDS;
        pq[i] = PointQuery{0, t, r, c};
        cop[i] = (uint32_t)i;
    }
    return run_points_multi(chunks, n, cop, pq, nullptr, n, out, out_mem, kernel_ms);
}
extern "C" int dcdf_query_fill_cell_batch(dcdf_chunk* const* chunks, const uint32_t* cells, size_t n, int64_t* out,
                                          const uint64_t* out_offset, int out_mem, float* kernel_ms) {
    if (!chunks || !cells || !out || !out_offset || n == 0 || n > 0x7fffffffu || (out_mem != DCDF_MEM_HOST && out_mem != DCDF_MEM_DEVICE)) return DCDF_ERR_BAD_ARG;
    if (!Runtime::get().ok) return DCDF_ERR_NO_DEVICE;
    std::vector<PointQuery> pq;
    std::vector<uint32_t> cop;
    std::vector<uint64_t> at;
    uint64_t hi = 0;
    for (size_t i = 0; i < n; i++) {
        const dcdf_chunk* h = chunks[i];
        if (!h) return DCDF_ERR_BAD_ARG;
        uint32_t a = cells[4 * i], b = cells[4 * i + 1];
        const uint32_t r = cells[4 * i + 2], c = cells[4 * i + 3];
        if (a > b) std::swap(a, b);
        if (b > h->instants || r >= h->rows || c >= h->cols) return DCDF_ERR_BOUNDS;
        for (uint32_t t = a; t < b; t++) {
            pq.push_back(PointQuery{0, t, r, c});
            cop.push_back((uint32_t)i);
            at.push_back(out_offset[i] + (t - a));
        }
        hi = std::max<uint64_t>(hi, out_offset[i] + (b - a));
    }
    if (pq.empty()) return DCDF_OK;
    if (pq.size() > 0x7fffffffu) return DCDF_ERR_CAPACITY;
    if (out_mem == DCDF_MEM_HOST) {  // dense staging on the device, then scattered into the caller's array
        std::vector<uint64_t> dense(pq.size());
        for (size_t i = 0; i < dense.size(); i++) dense[i] = i;
        std::vector<int64_t> tmp(pq.size());
        const int rc = run_points_multi(chunks, n, cop, pq, nullptr, pq.size(), tmp.data(), DCDF_MEM_HOST, kernel_ms);
        if (rc != DCDF_OK) return rc;
        for (size_t i = 0; i < tmp.size(); i++) out[at[i]] = tmp[i];
        (void)hi;
        return DCDF_OK;
    }
    return run_points_multi(chunks, n, cop, pq, &at, hi, out, DCDF_MEM_DEVICE, kernel_ms);
}


// (query, instant, sub-window) items of the wave kernel: sub-windows are the 32 x 32 squares of the chunk's grid that the
// window meets, so that a frontier level never exceeds what a wave's LDS queue holds (k2r::WQ_CAP)
// node_wise: pieces of at most 64 x 64 cells from the window's origin (k_window_wave2); else the squares of the chunk's 32-grid
// the window meets (k_window_wave)
// out_base = element offset of cell (c.start, c.top, c.left); sr / st = row and instant strides of the array the window is
// written into (0 = the window's own dense layout; a piece of a larger window passes the parent's)
static void window_items(uint32_t chunk, const dcdf_cube& c, uint64_t out_base, std::vector<WinItem>& items, bool node_wise,
                         uint64_t sr = 0, uint64_t st = 0) {
    const uint64_t wc = sr ? sr : (uint64_t)(c.right - c.left), wr_wc = st ? st : (uint64_t)(c.bottom - c.top) * wc;
    const uint32_t step = node_wise ? 64u : 32u;
    for (uint32_t t = c.start; t < c.end; t++)
        for (uint32_t r = node_wise ? c.top : (c.top & ~31u); r < c.bottom; r += step)
            for (uint32_t cc = node_wise ? c.left : (c.left & ~31u); cc < c.right; cc += step) {
                WinItem it{};
                it.chunk = chunk;
                it.inst = t;
                it.top = (uint16_t)std::max(r, c.top);
                it.bottom = (uint16_t)std::min(r + step, c.bottom);
                it.left = (uint16_t)std::max(cc, c.left);
                it.right = (uint16_t)std::min(cc + step, c.right);
                it.out_sr = (uint32_t)wc;
                it.out_off = out_base + (uint64_t)(t - c.start) * wr_wc + (uint64_t)(it.top - c.top) * wc + (it.left - c.left);
                items.push_back(it);
            }
}
// the wave kernel handles k * k <= 64 children per node and 16-bit coordinates
static bool wave_kernel_ok(const dcdf_chunk* h) {
    const uint32_t k = h->k0;
    return k * k <= 64 && h->sidelen0 <= 65535;
}
static bool node_kernel_ok(const dcdf_chunk* h) { return h->k0 == 2 && h->sidelen0 >= 4; }
static int launch_window_items_dev(const DevBuf& d_refs, const WinItem* d_items, uint32_t n, void* d_out, int32_t dtype, hipEvent_t e0, hipEvent_t e1,
                                   bool node_wise, bool narrow) {
    const uint32_t grid = std::min<uint32_t>((n + 3) / 4, 256u * 16u);
    if (e0) K2R_HIP(hipEventRecord(e0, 0));
    if (node_wise) {
        if (dtype == DCDF_I64 && narrow) hipLaunchKernelGGL((k_window_wave2<4, true, false, int32_t>), dim3(grid), dim3(256), 0, 0, d_refs.as<ChunkRef>(), d_items, n, d_out, dtype);
        else if (dtype == DCDF_I64) hipLaunchKernelGGL((k_window_wave2<3, true>), dim3(grid), dim3(256), 0, 0, d_refs.as<ChunkRef>(), d_items, n, d_out, dtype);
        else if (narrow) hipLaunchKernelGGL((k_window_wave2<4, false, false, int32_t>), dim3(grid), dim3(256), 0, 0, d_refs.as<ChunkRef>(), d_items, n, d_out, dtype);
        else hipLaunchKernelGGL((k_window_wave2<3, false>), dim3(grid), dim3(256), 0, 0, d_refs.as<ChunkRef>(), d_items, n, d_out, dtype);
    }
    else hipLaunchKernelGGL(k_window_wave, dim3(grid), dim3(256), 0, 0, d_refs.as<ChunkRef>(), d_items, n, d_out, dtype);
    if (e1) K2R_HIP(hipEventRecord(e1, 0));
    K2R_HIP(hipGetLastError());
    K2R_HIP(hipDeviceSynchronize());
    return DCDF_OK;
}
static int launch_window_items(const DevBuf& d_refs, const std::vector<WinItem>& items, void* d_out, int32_t dtype, hipEvent_t e0, hipEvent_t e1,
                               bool node_wise, bool narrow) {
    DevBuf d_items;
    K2R_HIP(d_items.alloc(items.size() * sizeof(WinItem)));
    K2R_HIP(hipMemcpy(d_items.p, items.data(), items.size() * sizeof(WinItem), hipMemcpyHostToDevice));
    return launch_window_items_dev(d_refs, d_items.as<WinItem>(), (uint32_t)items.size(), d_out, dtype, e0, e1, node_wise, narrow);
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
    if (wave_kernel_ok(h)) {
        std::vector<WinItem> items;
        window_items(0, c, 0, items, node_kernel_ok(h));
        const int rc = launch_window_items(d_ref, items, d_o.p, out_dtype, nullptr, nullptr, node_kernel_ok(h), h->narrow32);
        if (rc != DCDF_OK) return rc;
    } else {
        hipLaunchKernelGGL(k_fill_window, dim3(1), dim3(256), 0, 0, d_ref.as<ChunkRef>(), d_q.as<WinQuery>(), 1u, d_o.p,
                           out_dtype, (int64_t)(wr * wc), (int64_t)wc, (int64_t)1, 1);
        K2R_HIP(hipGetLastError());
    }
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

struct SearchCtx {  // a raster's view of its chunks (dcdf_raster_search_batch): nothing to de-duplicate or upload per call
    const DevBuf* refs;        // ChunkRef table, one entry per chunk of the raster
    const uint32_t* chunk_of;  // per query: index into it
    const uint32_t* origin;    // per query: (instant, row, col) of the chunk inside the raster, added to every triple
    bool node_wise, all_narrow;
    bool wave_ok;              // every chunk has k * k <= 64 (k_search_wave for the arities the node walk does not take)
};

static int search_impl(dcdf_chunk* const* chunks, const dcdf_cube* cubes, const int64_t* lower, const int64_t* upper,
                       size_t nq, uint32_t* out, size_t cap, uint64_t* counts, uint64_t* offsets, size_t* total_out,
                       float* kernel_ms, int out_mem = DCDF_MEM_HOST, const SearchCtx* ctx = nullptr) {
    std::vector<uint32_t> cidx;
    std::vector<const dcdf_chunk*> uniq;
    if (!ctx) dedup_chunks(chunks, nq, cidx, uniq);
    std::vector<WinQuery> qs(nq);
    std::vector<SearchItem> items;
    // k = 2 chunks: the wave-cooperative walk of fill_window marks the matches (one wave per piece of <= 64 x 64 cells and
    // instant, each into its own 128-word bitmap; the instants of dcdf_chunk::search_quirk carry a flag); other arities are
    // decoded and tested cell by cell (k_search_cells) into a flat per-item bitmap
    bool node_wise = std::getenv("K2R_SEARCH_DFS") == nullptr;  // (diagnostics: A/B against the cell-by-cell kernel)
    for (const dcdf_chunk* u : uniq) node_wise = node_wise && node_kernel_ok(u);
    if (ctx) node_wise = node_wise && ctx->node_wise;
    // other arities with k * k <= 64: the pruned wave walk k_search_wave, one wave per (item, sub-window of <= 32 x 32 cells), into the
    // item's flat window bitmap; what is left (k > 8) is decoded and tested cell by cell
    bool wave_search = !node_wise && std::getenv("K2R_SEARCH_CELLS") == nullptr;
    for (const dcdf_chunk* u : uniq) wave_search = wave_search && wave_kernel_ok(u);
    if (ctx) wave_search = wave_search && ctx->wave_ok;
    std::vector<WinItem> witems;
    std::vector<SearchExtra> sx;
    std::vector<uint8_t> item_quirk;  // per item of the decode-and-test kernel
    uint64_t bits_words = 0;
    size_t n_dfs = 0;
    for (size_t q = 0; q < nq; q++) {
        if (!chunks[q]) return DCDF_ERR_BAD_ARG;
        const dcdf_cube c = norm_cube(cubes[q]);
        if (!cube_in(chunks[q], c)) return DCDF_ERR_BOUNDS;
        WinQuery& Q = qs[q];
        Q = WinQuery{};
        Q.chunk = ctx ? ctx->chunk_of[q] : cidx[q];
        if (ctx) {
            Q._pad = ctx->origin[3 * q];
            Q.out_off = (uint64_t)ctx->origin[3 * q + 1] | (uint64_t)ctx->origin[3 * q + 2] << 32;
        }
        Q.start = c.start; Q.end = c.end; Q.top = c.top; Q.bottom = c.bottom; Q.left = c.left; Q.right = c.right;
        Q.lower = std::min(lower[q], upper[q]);  // helpers.rs:7-16 via chunk.rs:214
        Q.upper = std::max(lower[q], upper[q]);
        const uint64_t cells = (uint64_t)(c.bottom - c.top) * (c.right - c.left);
        if (cells == 0) continue;
        const uint32_t ncb = (c.right - c.left + 63u) >> 6;
        for (uint32_t i = c.start; i < c.end; i++) {
            if (node_wise) {
                if (witems.size() + 4096 > 0xffffff00u) return DCDF_ERR_CAPACITY;
                items.push_back(SearchItem{(uint32_t)q, i, 0, (uint32_t)witems.size(), ncb});
                for (uint32_t r = c.top; r < c.bottom; r += 64)
                    for (uint32_t cc = c.left; cc < c.right; cc += 64) {
                        WinItem it{};
                        it.chunk = Q.chunk;
                        it.inst = i;
                        it.top = (uint16_t)r;
                        it.bottom = (uint16_t)std::min(r + 64, c.bottom);
                        it.left = (uint16_t)cc;
                        it.right = (uint16_t)std::min(cc + 64, c.right);
                        witems.push_back(it);
                        sx.push_back(SearchExtra{Q.lower, Q.upper, chunks[q]->search_quirk[i] ? 1u : 0u, 0u});
                    }
            } else {
                items.push_back(SearchItem{(uint32_t)q, i, bits_words, SI_FLAT, 0});
                item_quirk.resize(items.size(), 0);
                item_quirk.back() = chunks[q]->search_quirk[i];
                if (wave_search) {
                    if (witems.size() + 4096 > 0xffffff00u) return DCDF_ERR_CAPACITY;
                    const uint32_t wc = c.right - c.left;
                    for (uint32_t r = c.top & ~31u; r < c.bottom; r += 32)
                        for (uint32_t cc = c.left & ~31u; cc < c.right; cc += 32) {
                            WinItem it{};
                            it.chunk = Q.chunk;
                            it.inst = i;
                            it.top = (uint16_t)std::max(r, c.top);
                            it.bottom = (uint16_t)std::min(r + 32, c.bottom);
                            it.left = (uint16_t)std::max(cc, c.left);
                            it.right = (uint16_t)std::min(cc + 32, c.right);
                            it.out_sr = wc;  // bit (r, c) of the item's bitmap = out_off + (r - top) * out_sr + (c - left)
                            it.out_off = bits_words * 32ull + (uint64_t)(it.top - c.top) * wc + (it.left - c.left);
                            witems.push_back(it);
                            sx.push_back(SearchExtra{Q.lower, Q.upper, chunks[q]->search_quirk[i] ? 1u : 0u, (uint32_t)(items.size() - 1)});
                        }
                }
                bits_words += (cells + 31) / 32;
                n_dfs++;
            }
        }
    }
    for (size_t q = 0; q < nq; q++) counts[q] = 0;
    float ms_total = 0.f;
    std::vector<uint32_t> item_counts(items.size());
    DevBuf d_refs_own, d_qs, d_items, d_bits, d_wbits, d_witems, d_sx, d_counts, d_offs, d_out, d_quirk;
    if (!items.empty()) {
        if (!ctx) {
            int rc = upload_refs(chunks, cidx, uniq.size(), uniq, d_refs_own);
            if (rc != DCDF_OK) return rc;
        }
        const DevBuf& d_refs = ctx ? *ctx->refs : d_refs_own;
        K2R_HIP(d_qs.alloc(nq * sizeof(WinQuery)));
        K2R_HIP(hipMemcpy(d_qs.p, qs.data(), nq * sizeof(WinQuery), hipMemcpyHostToDevice));
        K2R_HIP(d_items.alloc(items.size() * sizeof(SearchItem)));
        K2R_HIP(hipMemcpy(d_items.p, items.data(), items.size() * sizeof(SearchItem), hipMemcpyHostToDevice));
        if (n_dfs) {
            K2R_HIP(d_bits.alloc(std::max<uint64_t>(bits_words, 1) * 4));
            K2R_HIP(hipMemset(d_bits.p, 0, std::max<uint64_t>(bits_words, 1) * 4));
        }
        const uint32_t nw = (uint32_t)witems.size();
        if (nw) {
            K2R_HIP(d_witems.alloc(witems.size() * sizeof(WinItem)));
            K2R_HIP(hipMemcpy(d_witems.p, witems.data(), witems.size() * sizeof(WinItem), hipMemcpyHostToDevice));
            K2R_HIP(d_sx.alloc(sx.size() * sizeof(SearchExtra)));
            K2R_HIP(hipMemcpy(d_sx.p, sx.data(), sx.size() * sizeof(SearchExtra), hipMemcpyHostToDevice));
            if (node_wise) K2R_HIP(d_wbits.alloc((size_t)nw * 512));  // (every word is written by the walk: nothing to clear)
        }
        K2R_HIP(d_counts.alloc(items.size() * 4));
        if (wave_search) K2R_HIP(hipMemset(d_counts.p, 0, items.size() * 4));  // (several waves add to an item's count)
        EventPair ev;
        K2R_HIP(ev.create());
        const hipEvent_t e0 = ev.e0, e1 = ev.e1;
        const uint32_t ni = (uint32_t)items.size();
        K2R_HIP(hipEventRecord(e0, 0));
        if (nw && wave_search) {
            hipLaunchKernelGGL(k_search_wave, dim3(std::min<uint32_t>((nw + 3) / 4, 256u * 16u)), dim3(256), 0, 0, d_refs.as<ChunkRef>(),
                               d_witems.as<WinItem>(), nw, d_sx.as<SearchExtra>(), d_bits.as<uint32_t>(), d_counts.as<uint32_t>());
        } else if (nw) {
            bool all_narrow = true;
            for (const dcdf_chunk* u : uniq) all_narrow = all_narrow && u->narrow32;
            if (ctx) all_narrow = ctx->all_narrow;
            if (all_narrow)
                hipLaunchKernelGGL((k_window_wave2<4, false, true, int32_t>), dim3(std::min<uint32_t>((nw + 3) / 4, 256u * 16u)), dim3(256), 0, 0,
                                   d_refs.as<ChunkRef>(), d_witems.as<WinItem>(), nw, d_wbits.p, (int32_t)DCDF_I64, d_sx.as<SearchExtra>());
            else
                hipLaunchKernelGGL((k_window_wave2<3, false, true>), dim3(std::min<uint32_t>((nw + 3) / 4, 256u * 16u)), dim3(256), 0, 0,
                                   d_refs.as<ChunkRef>(), d_witems.as<WinItem>(), nw, d_wbits.p, (int32_t)DCDF_I64, d_sx.as<SearchExtra>());
            hipLaunchKernelGGL(k_search_count, dim3((ni + 63) / 64), dim3(64), 0, 0, d_wbits.as<uint32_t>(), d_items.as<SearchItem>(),
                               d_qs.as<WinQuery>(), ni, d_counts.as<uint32_t>());
        }
        if (n_dfs && !wave_search) {
            item_quirk.resize(items.size(), 0);
            K2R_HIP(d_quirk.alloc(items.size()));
            K2R_HIP(hipMemcpy(d_quirk.p, item_quirk.data(), items.size(), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_search_cells, dim3(ni), dim3(64), 0, 0, d_refs.as<ChunkRef>(), d_qs.as<WinQuery>(),
                               d_items.as<SearchItem>(), ni, d_quirk.as<uint8_t>(), d_bits.as<uint32_t>(), d_counts.as<uint32_t>());
        }
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
            const bool to_dev = out_mem == DCDF_MEM_DEVICE;  // the triples stay where the emit kernel writes them
            if (!to_dev) K2R_HIP(d_out.alloc(run * 12));
            K2R_HIP(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_search_emit, dim3((ni + 63) / 64), dim3(64), 0, 0, d_qs.as<WinQuery>(),
                               d_items.as<SearchItem>(), ni, d_bits.as<uint32_t>(), d_wbits.as<uint32_t>(), d_offs.as<uint64_t>(),
                               to_dev ? out : d_out.as<uint32_t>());
            K2R_HIP(hipEventRecord(e1, 0));
            K2R_HIP(hipGetLastError());
            if (to_dev) K2R_HIP(hipDeviceSynchronize());
            else K2R_HIP(hipMemcpy(out, d_out.p, run * 12, hipMemcpyDeviceToHost));
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

// out_dtype: the element type written (MMBuffer3::set conversion of the stored value, mmbuffer.rs:292-299: i32 / i64 as is,
// f32 / f64 through from_fixed); out_mem: `out` is host or DEVICE memory -- a device `out` is written by the decode kernel
// itself at out_offset[q] (elements), nothing crosses PCIe
static int fill_window_batch_impl(dcdf_chunk* const* chunks, const dcdf_cube* cubes, size_t nq, void* out, int32_t out_dtype, int out_mem,
                                  const uint64_t* out_offset, float* kernel_ms) {
    if (!chunks || !cubes || !out || !out_offset || nq == 0) return DCDF_ERR_BAD_ARG;
    if (out_dtype != DCDF_I32 && out_dtype != DCDF_I64 && out_dtype != DCDF_F32 && out_dtype != DCDF_F64) return DCDF_ERR_BAD_ARG;
    if (out_mem != DCDF_MEM_HOST && out_mem != DCDF_MEM_DEVICE) return DCDF_ERR_BAD_ARG;
    if (!Runtime::get().ok) return DCDF_ERR_NO_DEVICE;
    const size_t es = (out_dtype == DCDF_I32 || out_dtype == DCDF_F32) ? 4 : 8;
    const bool to_dev = out_mem == DCDF_MEM_DEVICE;
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
        Q.out_off = to_dev ? out_offset[q] : total;
        dense = dense && out_offset[q] == total + out_offset[0];
        total += (uint64_t)(c.end - c.start) * (c.bottom - c.top) * (c.right - c.left);
    }
    if (total == 0) return DCDF_OK;
    DevBuf d_refs, d_qs, d_o;
    int rc = upload_refs(chunks, cidx, uniq.size(), uniq, d_refs);
    if (rc != DCDF_OK) return rc;
    K2R_HIP(d_qs.alloc(nq * sizeof(WinQuery)));
    K2R_HIP(hipMemcpy(d_qs.p, qs.data(), nq * sizeof(WinQuery), hipMemcpyHostToDevice));
    if (!to_dev) K2R_HIP(d_o.alloc(total * es));
    void* const d_dst = to_dev ? out : d_o.p;
    EventPair ev;
    K2R_HIP(ev.create());
    bool all_wave = true, all_node = true;
    bool all_narrow = true;
    for (const dcdf_chunk* u : uniq) {
        all_wave = all_wave && wave_kernel_ok(u);
        all_node = all_node && node_kernel_ok(u);
        all_narrow = all_narrow && u->narrow32;
    }
    if (all_wave) {
        std::vector<WinItem> items;
        for (size_t q = 0; q < nq; q++) {
            const dcdf_cube c{qs[q].start, qs[q].end, qs[q].top, qs[q].bottom, qs[q].left, qs[q].right};
            window_items(qs[q].chunk, c, qs[q].out_off, items, all_node);
        }
        rc = launch_window_items(d_refs, items, d_dst, out_dtype, ev.e0, ev.e1, all_node, all_narrow);
        if (rc != DCDF_OK) return rc;
    } else {
        // arities beyond the wave walk (k * k > 64): per-cell descents, int64 only
        if (out_dtype != DCDF_I64) return DCDF_ERR_UNSUPPORTED;
        const uint32_t grid = (uint32_t)std::min<size_t>(nq, 1u << 20);
        K2R_HIP(hipEventRecord(ev.e0, 0));
        hipLaunchKernelGGL(k_fill_window, dim3(grid), dim3(256), 0, 0, d_refs.as<ChunkRef>(), d_qs.as<WinQuery>(),
                           (uint32_t)nq, d_dst, (int32_t)DCDF_I64, (int64_t)0, (int64_t)0, (int64_t)0, 0);
        K2R_HIP(hipEventRecord(ev.e1, 0));
        K2R_HIP(hipGetLastError());
        K2R_HIP(hipDeviceSynchronize());
    }
    if (to_dev) {
        // (already in place)
    } else if (dense) {  // the usual case: windows back to back in query order -> one copy straight into the caller's array
        K2R_HIP(hipMemcpy((uint8_t*)out + out_offset[0] * es, d_o.p, total * es, hipMemcpyDeviceToHost));
    } else {
        std::vector<uint8_t> tmp(total * es);
        K2R_HIP(hipMemcpy(tmp.data(), d_o.p, total * es, hipMemcpyDeviceToHost));
        for (size_t q = 0; q < nq; q++) {
            const uint64_t cells = (uint64_t)(qs[q].end - qs[q].start) * (qs[q].bottom - qs[q].top) * (qs[q].right - qs[q].left);
            if (cells) std::memcpy((uint8_t*)out + out_offset[q] * es, tmp.data() + qs[q].out_off * es, cells * es);
        }
    }
    float ms = 0.f;
    K2R_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
    if (kernel_ms) *kernel_ms = ms;
    return DCDF_OK;
}
extern "C" int dcdf_query_search_batch_mem(dcdf_chunk* const* chunks, const dcdf_cube* cubes, const int64_t* lower,
                                           const int64_t* upper, size_t nq, uint32_t* out, size_t cap, int out_mem,
                                           uint64_t* counts, uint64_t* offsets, float* kernel_ms) {
    if (!chunks || !cubes || !lower || !upper || !counts || !offsets || nq == 0 || (!out && cap)) return DCDF_ERR_BAD_ARG;
    if (out_mem != DCDF_MEM_HOST && out_mem != DCDF_MEM_DEVICE) return DCDF_ERR_BAD_ARG;
    if (!Runtime::get().ok) return DCDF_ERR_NO_DEVICE;
    size_t total = 0;
    return search_impl(chunks, cubes, lower, upper, nq, out, cap, counts, offsets, &total, kernel_ms, out_mem);
}
extern "C" int dcdf_query_fill_window_batch(dcdf_chunk* const* chunks, const dcdf_cube* cubes, size_t nq, int64_t* out,
                                            const uint64_t* out_offset, float* kernel_ms) {
    return fill_window_batch_impl(chunks, cubes, nq, out, (int32_t)DCDF_I64, DCDF_MEM_HOST, out_offset, kernel_ms);
}
extern "C" int dcdf_query_fill_window_batch_typed(dcdf_chunk* const* chunks, const dcdf_cube* cubes, size_t nq, void* out,
                                                  int32_t out_dtype, int out_mem, const uint64_t* out_offset, float* kernel_ms) {
    return fill_window_batch_impl(chunks, cubes, nq, out, out_dtype, out_mem, out_offset, kernel_ms);
}


// ---- a tiled, time-segmented raster of opened chunks: the routing of the layers above, natively ---------------------------------
// Variable::append cuts [instants, rows, cols] into time segments of chunk_size instants (dataset.rs:838) and Superchunk::build
// cuts each segment into tile x tile sub-arrays (superchunk.rs:127-181); reads are routed back the same way (Span::fill_window
// span.rs:190-216 over time, Superchunk::subchunks_for superchunk.rs:589-633 over rows / cols).  dcdf_raster does that split for a
// whole batch of dataset-level cubes on the host in C++ and decodes every piece in ONE launch straight into its place in the
// caller's window (the pieces carry the parent window's strides): no per-piece copies, no reassembly, the chunk table uploaded once.
struct dcdf_raster {
    std::vector<dcdf_chunk*> chunks;  // [(segment * nti + ti) * ntj + tj]
    // d_refs holds device pointers into the chunks' streams and tables: the raster shares the ownership of every slab a batch-opened
    // chunk lives in (dcdf_chunk::store), so closing such a chunk first leaves the raster usable; chunks opened one by one own their
    // buffers themselves and must outlive the raster (dcdf_k2r.h)
    std::vector<std::shared_ptr<void>> keep;
    uint32_t T = 0, R = 0, C = 0, tile = 0, cs = 0, nseg = 0, nti = 0, ntj = 0;
    DevBuf d_refs;
    DevBuf d_quirk;  // [chunk][chunk_size]: dcdf_chunk::search_quirk of every instant (k_raster_search_expand)
    bool all_wave = true, all_node = true, all_narrow = true;
};
extern "C" int dcdf_raster_create(dcdf_chunk* const* chunks, size_t n_chunks, const uint32_t shape[3], uint32_t tile, uint32_t chunk_size,
                                  dcdf_raster** out) {
    if (!chunks || !shape || !out || tile == 0 || chunk_size == 0 || shape[0] == 0 || shape[1] == 0 || shape[2] == 0) return DCDF_ERR_BAD_ARG;
    if (!Runtime::get().ok) return DCDF_ERR_NO_DEVICE;
    std::unique_ptr<dcdf_raster> r(new (std::nothrow) dcdf_raster());
    if (!r) return DCDF_ERR_NOMEM;
    r->T = shape[0]; r->R = shape[1]; r->C = shape[2]; r->tile = tile; r->cs = chunk_size;
    r->nseg = (r->T + chunk_size - 1) / chunk_size;
    r->nti = (r->R + tile - 1) / tile;
    r->ntj = (r->C + tile - 1) / tile;
    if ((uint64_t)r->nseg * r->nti * r->ntj != n_chunks) return DCDF_ERR_BAD_ARG;
    r->chunks.assign(chunks, chunks + n_chunks);
    for (size_t i = 0; i < n_chunks; i++)
        if (chunks[i] && chunks[i]->store && (r->keep.empty() || r->keep.back() != chunks[i]->store)) r->keep.push_back(chunks[i]->store);
    std::vector<ChunkRef> refs(n_chunks);
    for (size_t i = 0; i < n_chunks; i++) {
        const dcdf_chunk* h = chunks[i];
        if (!h) return DCDF_ERR_BAD_ARG;
        const uint32_t seg = (uint32_t)(i / ((size_t)r->nti * r->ntj)), ti = (uint32_t)(i / r->ntj % r->nti), tj = (uint32_t)(i % r->ntj);
        // every chunk must have the shape its place in the grid gives it
        if (h->instants != std::min(chunk_size, r->T - seg * chunk_size) || h->rows != std::min(tile, r->R - ti * tile) ||
            h->cols != std::min(tile, r->C - tj * tile))
            return DCDF_ERR_BAD_ARG;
        refs[i] = make_ref(h);
        r->all_wave = r->all_wave && wave_kernel_ok(h);
        r->all_node = r->all_node && node_kernel_ok(h);
        r->all_narrow = r->all_narrow && h->narrow32;
    }
    K2R_HIP(r->d_refs.alloc(n_chunks * sizeof(ChunkRef)));
    K2R_HIP(hipMemcpy(r->d_refs.p, refs.data(), n_chunks * sizeof(ChunkRef), hipMemcpyHostToDevice));
    std::vector<uint8_t> quirk(n_chunks * (size_t)chunk_size, 0);
    for (size_t i = 0; i < n_chunks; i++)
        for (size_t t = 0; t < chunks[i]->search_quirk.size() && t < chunk_size; t++) quirk[i * chunk_size + t] = chunks[i]->search_quirk[t];
    K2R_HIP(r->d_quirk.alloc(quirk.size()));
    K2R_HIP(hipMemcpy(r->d_quirk.p, quirk.data(), quirk.size(), hipMemcpyHostToDevice));
    *out = r.release();
    return DCDF_OK;
}
extern "C" void dcdf_raster_destroy(dcdf_raster* r) { delete r; }

// the pieces of one dataset-level cube: f(chunk id, local cube, raster origin of the chunk)
template <class F>
static void raster_pieces(const dcdf_raster* r, const dcdf_cube& c, F&& f) {
    for (uint32_t seg = c.start / r->cs; seg <= (c.end - 1) / r->cs; seg++)
        for (uint32_t ti = c.top / r->tile; ti <= (c.bottom - 1) / r->tile; ti++)
            for (uint32_t tj = c.left / r->tile; tj <= (c.right - 1) / r->tile; tj++) {
                const uint32_t t0 = seg * r->cs, r0 = ti * r->tile, c0 = tj * r->tile;
                const dcdf_cube l{std::max(c.start, t0) - t0, std::min(c.end, t0 + r->cs) - t0, std::max(c.top, r0) - r0,
                                  std::min(c.bottom, r0 + r->tile) - r0, std::max(c.left, c0) - c0, std::min(c.right, c0 + r->tile) - c0};
                f((uint32_t)(((uint64_t)seg * r->nti + ti) * r->ntj + tj), l, t0, r0, c0);
            }
}
// One thread per dataset-level cube: the wave items of its pieces (the loops of raster_pieces x window_items), written at
// item_base[q]; the host only counts them (a closed form per cube) -- 24 bytes per item need not cross PCIe.
struct RasterGeom {
    uint32_t T, R, C, tile, cs, nti, ntj, step;  // step: 64 (node-wise walk) or 32
};
__global__ void __launch_bounds__(256)
k_raster_expand(const dcdf_cube* __restrict__ cubes, const uint64_t* __restrict__ out_base, const uint32_t* __restrict__ item_base, uint32_t nq,
                RasterGeom g, WinItem* __restrict__ items) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    dcdf_cube c = cubes[q];
    if (c.start > c.end) { const uint32_t x = c.start; c.start = c.end; c.end = x; }  // helpers.rs:7-16 (norm_cube)
    if (c.top > c.bottom) { const uint32_t x = c.top; c.top = c.bottom; c.bottom = x; }
    if (c.left > c.right) { const uint32_t x = c.left; c.left = c.right; c.right = x; }
    const uint64_t wr = c.bottom - c.top, wc = c.right - c.left;
    if ((uint64_t)(c.end - c.start) * wr * wc == 0) return;
    WinItem* o = items + item_base[q];
    const uint64_t base = out_base[q];
    for (uint32_t seg = c.start / g.cs; seg <= (c.end - 1) / g.cs; seg++)
        for (uint32_t ti = c.top / g.tile; ti <= (c.bottom - 1) / g.tile; ti++)
            for (uint32_t tj = c.left / g.tile; tj <= (c.right - 1) / g.tile; tj++) {
                const uint32_t t0 = seg * g.cs, r0 = ti * g.tile, c0 = tj * g.tile;
                const uint32_t ls = max(c.start, t0) - t0, le = min(c.end, t0 + g.cs) - t0, lt = max(c.top, r0) - r0, lb = min(c.bottom, r0 + g.tile) - r0,
                               ll = max(c.left, c0) - c0, lr = min(c.right, c0 + g.tile) - c0;
                const uint32_t cid = (seg * g.nti + ti) * g.ntj + tj;
                const uint64_t at = base + ((uint64_t)(t0 + ls - c.start) * wr + (r0 + lt - c.top)) * wc + (c0 + ll - c.left);
                const uint32_t rs = g.step == 64 ? lt : (lt & ~31u), cs0 = g.step == 64 ? ll : (ll & ~31u);
                for (uint32_t t = ls; t < le; t++)
                    for (uint32_t rr = rs; rr < lb; rr += g.step)
                        for (uint32_t cc = cs0; cc < lr; cc += g.step) {
                            WinItem it;
                            it.chunk = cid;
                            it.inst = t;
                            it.top = (uint16_t)max(rr, lt);
                            it.bottom = (uint16_t)min(rr + g.step, lb);
                            it.left = (uint16_t)max(cc, ll);
                            it.right = (uint16_t)min(cc + g.step, lr);
                            it.out_sr = (uint32_t)wc;
                            it.out_off = at + (uint64_t)(t - ls) * wr * wc + (uint64_t)(it.top - lt) * wc + (it.left - ll);
                            *o++ = it;
                        }
            }
}
// wave items of one cube (the count of the loops above)
static uint64_t raster_item_count(const dcdf_raster* r, const dcdf_cube& c, uint32_t step) {
    auto along = [&](uint32_t a, uint32_t b, uint32_t unit) {  // sum over the tiles [a, b) meets of ceil(piece / step) (from the piece's start, or the 32-grid)
        uint64_t n = 0;
        for (uint32_t t = a / unit; t <= (b - 1) / unit; t++) {
            const uint32_t lo = std::max(a, t * unit) - t * unit, hi = std::min(b, t * unit + unit) - t * unit;
            const uint32_t from = step == 64 ? lo : (lo & ~31u);
            n += (hi - from + step - 1) / step;
        }
        return n;
    };
    return (uint64_t)(c.end - c.start) * along(c.top, c.bottom, r->tile) * along(c.left, c.right, r->tile);
}

extern "C" int dcdf_raster_fill_window_batch(const dcdf_raster* r, const dcdf_cube* cubes, size_t nq, void* out, int32_t out_dtype,
                                             int out_mem, const uint64_t* out_offset, float* kernel_ms) {
    if (!r || !cubes || !out || !out_offset || nq == 0 || nq > 0x7fffffffu) return DCDF_ERR_BAD_ARG;
    if (out_dtype != DCDF_I32 && out_dtype != DCDF_I64 && out_dtype != DCDF_F32 && out_dtype != DCDF_F64) return DCDF_ERR_BAD_ARG;
    if (out_mem != DCDF_MEM_HOST && out_mem != DCDF_MEM_DEVICE) return DCDF_ERR_BAD_ARG;
    if (!r->all_wave) return DCDF_ERR_UNSUPPORTED;  // arities beyond the wave walk (k * k > 64): use the per-chunk entry points
    const size_t es = (out_dtype == DCDF_I32 || out_dtype == DCDF_F32) ? 4 : 8;
    const bool to_dev = out_mem == DCDF_MEM_DEVICE;
    const uint32_t step = r->all_node ? 64u : 32u;
    // host: bounds, where each window goes, how many wave items it makes; device: the items themselves (k_raster_expand)
    std::vector<uint64_t> base(nq);
    std::vector<uint32_t> item_base(nq);
    uint64_t total = 0, n_items = 0;
    bool dense = true;
    for (size_t q = 0; q < nq; q++) {
        const dcdf_cube c = norm_cube(cubes[q]);
        if (c.end > r->T || c.bottom > r->R || c.right > r->C) return DCDF_ERR_BOUNDS;
        const uint64_t cells = (uint64_t)(c.end - c.start) * (c.bottom - c.top) * (c.right - c.left);
        dense = dense && out_offset[q] == total + out_offset[0];
        base[q] = to_dev ? out_offset[q] : total;
        total += cells;
        item_base[q] = (uint32_t)n_items;
        if (cells) n_items += raster_item_count(r, c, step);
        if (n_items > 0xfffffff0ull) return DCDF_ERR_CAPACITY;
    }
    if (n_items == 0) return DCDF_OK;
    DevBuf d_o, d_cubes, d_base, d_ibase, d_items;
    if (!to_dev) K2R_HIP(d_o.alloc_pooled(total * es));
    K2R_HIP(d_cubes.alloc(nq * sizeof(dcdf_cube)));
    K2R_HIP(d_base.alloc(nq * 8));
    K2R_HIP(d_ibase.alloc(nq * 4));
    K2R_HIP(d_items.alloc_pooled(n_items * sizeof(WinItem)));
    K2R_HIP(hipMemcpy(d_cubes.p, cubes, nq * sizeof(dcdf_cube), hipMemcpyHostToDevice));
    K2R_HIP(hipMemcpy(d_base.p, base.data(), nq * 8, hipMemcpyHostToDevice));
    K2R_HIP(hipMemcpy(d_ibase.p, item_base.data(), nq * 4, hipMemcpyHostToDevice));
    const RasterGeom g{r->T, r->R, r->C, r->tile, r->cs, r->nti, r->ntj, step};
    hipLaunchKernelGGL(k_raster_expand, dim3((uint32_t)((nq + 255) / 256)), dim3(256), 0, 0, d_cubes.as<dcdf_cube>(), d_base.as<uint64_t>(),
                       d_ibase.as<uint32_t>(), (uint32_t)nq, g, d_items.as<WinItem>());
    K2R_HIP(hipGetLastError());
    EventPair ev;
    K2R_HIP(ev.create());
    const int rc = launch_window_items_dev(r->d_refs, d_items.as<WinItem>(), (uint32_t)n_items, to_dev ? out : d_o.p, out_dtype, ev.e0, ev.e1,
                                           r->all_node, r->all_narrow);
    if (rc != DCDF_OK) return rc;
    if (!to_dev) {
        if (dense) {
            K2R_HIP(hipMemcpy((uint8_t*)out + out_offset[0] * es, d_o.p, total * es, hipMemcpyDeviceToHost));
        } else {
            std::vector<uint8_t> tmp(total * es);
            K2R_HIP(hipMemcpy(tmp.data(), d_o.p, total * es, hipMemcpyDeviceToHost));
            uint64_t run = 0;
            for (size_t q = 0; q < nq; q++) {
                const dcdf_cube c = norm_cube(cubes[q]);
                const uint64_t cells = (uint64_t)(c.end - c.start) * (c.bottom - c.top) * (c.right - c.left);
                if (cells) std::memcpy((uint8_t*)out + out_offset[q] * es, tmp.data() + run * es, cells * es);
                run += cells;
            }
        }
    }
    float ms = 0.f;
    K2R_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
    if (kernel_ms) *kernel_ms = ms;
    return DCDF_OK;
}
// ---- search of dataset-level cubes with everything but a count per cube on the device ---------------------------------
// One thread per cube writes what search_impl builds on the host: a WinQuery per chunk-level piece (with the chunk's origin
// for the emit kernel), a SearchItem per (piece, instant), a WinItem + SearchExtra per <= 64 x 64 part of it.
__global__ void __launch_bounds__(256)
k_raster_search_expand(const dcdf_cube* __restrict__ cubes, const int64_t* __restrict__ lower, const int64_t* __restrict__ upper,
                       const uint32_t* __restrict__ sb, const uint32_t* __restrict__ ib, const uint32_t* __restrict__ wb, uint32_t nq, RasterGeom g,
                       const uint8_t* __restrict__ quirk, WinQuery* __restrict__ qs, SearchItem* __restrict__ items, WinItem* __restrict__ witems,
                       SearchExtra* __restrict__ sx) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    dcdf_cube c = cubes[q];
    if (c.start > c.end) { const uint32_t x = c.start; c.start = c.end; c.end = x; }
    if (c.top > c.bottom) { const uint32_t x = c.top; c.top = c.bottom; c.bottom = x; }
    if (c.left > c.right) { const uint32_t x = c.left; c.left = c.right; c.right = x; }
    if ((uint64_t)(c.end - c.start) * (c.bottom - c.top) * (c.right - c.left) == 0) return;
    const int64_t lo = min(lower[q], upper[q]), hi = max(lower[q], upper[q]);  // helpers.rs:7-16 via chunk.rs:214
    uint32_t s = sb[q], it = ib[q], w = wb[q];
    for (uint32_t seg = c.start / g.cs; seg <= (c.end - 1) / g.cs; seg++)
        for (uint32_t ti = c.top / g.tile; ti <= (c.bottom - 1) / g.tile; ti++)
            for (uint32_t tj = c.left / g.tile; tj <= (c.right - 1) / g.tile; tj++) {
                const uint32_t t0 = seg * g.cs, r0 = ti * g.tile, c0 = tj * g.tile;
                const uint32_t ls = max(c.start, t0) - t0, le = min(c.end, t0 + g.cs) - t0, lt = max(c.top, r0) - r0, lb = min(c.bottom, r0 + g.tile) - r0,
                               ll = max(c.left, c0) - c0, lr = min(c.right, c0 + g.tile) - c0;
                const uint32_t cid = (seg * g.nti + ti) * g.ntj + tj;
                WinQuery Q;
                Q.chunk = cid;
                Q.start = ls; Q.end = le; Q.top = lt; Q.bottom = lb; Q.left = ll; Q.right = lr;
                Q._pad = t0;  // the chunk's origin inside the raster, added to every triple by k_search_emit
                Q.lower = lo;
                Q.upper = hi;
                Q.out_off = (uint64_t)r0 | (uint64_t)c0 << 32;
                qs[s] = Q;
                const uint32_t ncb = (lr - ll + 63u) >> 6;
                for (uint32_t t = ls; t < le; t++) {
                    items[it++] = SearchItem{s, t, 0, w, ncb};
                    const uint32_t qk = quirk[(size_t)cid * g.cs + t];
                    for (uint32_t rr = lt; rr < lb; rr += 64)
                        for (uint32_t cc = ll; cc < lr; cc += 64) {
                            WinItem wi;
                            wi.chunk = cid;
                            wi.inst = t;
                            wi.top = (uint16_t)rr;
                            wi.bottom = (uint16_t)min(rr + 64, lb);
                            wi.left = (uint16_t)cc;
                            wi.right = (uint16_t)min(cc + 64, lr);
                            wi.out_sr = 0;
                            wi.out_off = 0;
                            witems[w] = wi;
                            sx[w] = SearchExtra{lo, hi, qk, 0u};
                            w++;
                        }
                }
                s++;
            }
}
// exclusive prefix sum of n uint32 counts into uint64 offsets: block sums, their scan by one block, the offsets
constexpr uint32_t kScanPer = 2048;  // elements per 256-thread block
__global__ void __launch_bounds__(256) k_scan_sums(const uint32_t* __restrict__ v, uint32_t n, uint64_t* __restrict__ sums) {
    __shared__ uint64_t part[256];
    const uint32_t b0 = blockIdx.x * kScanPer;
    uint64_t a = 0;
    for (uint32_t i = threadIdx.x; i < kScanPer && b0 + i < n; i += 256) a += v[b0 + i];
    part[threadIdx.x] = a;
    __syncthreads();
    for (uint32_t st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[blockIdx.x] = part[0];
}
__global__ void __launch_bounds__(256) k_scan_top(uint64_t* __restrict__ sums, uint32_t nb, uint64_t* __restrict__ total) {
    // (one block; nb is small: n / 2048) sums[b] <- sum of the blocks before b
    __shared__ uint64_t part[256];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nb; b0 += 256) {
        const uint32_t i = b0 + threadIdx.x;
        const uint64_t x = i < nb ? sums[i] : 0;
        part[threadIdx.x] = x;
        __syncthreads();
        for (uint32_t st = 1; st < 256; st <<= 1) {  // inclusive scan (Hillis-Steele)
            const uint64_t y = threadIdx.x >= st ? part[threadIdx.x - st] : 0;
            __syncthreads();
            part[threadIdx.x] += y;
            __syncthreads();
        }
        if (i < nb) sums[i] = carry + part[threadIdx.x] - x;
        __syncthreads();
        if (threadIdx.x == 255) carry += part[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ void __launch_bounds__(256) k_scan_apply(const uint32_t* __restrict__ v, uint32_t n, const uint64_t* __restrict__ sums,
                                                    uint64_t* __restrict__ offs) {
    // thread t of the block owns 8 consecutive elements: its prefix inside the block by a scan of the threads' sums
    __shared__ uint64_t part[256];
    const uint32_t b0 = blockIdx.x * kScanPer + threadIdx.x * 8;
    uint32_t x[8];
    uint64_t a = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        x[j] = b0 + j < n ? v[b0 + j] : 0u;
        a += x[j];
    }
    part[threadIdx.x] = a;
    __syncthreads();
    for (uint32_t st = 1; st < 256; st <<= 1) {
        const uint64_t y = threadIdx.x >= st ? part[threadIdx.x - st] : 0;
        __syncthreads();
        part[threadIdx.x] += y;
        __syncthreads();
    }
    uint64_t run = sums[blockIdx.x] + part[threadIdx.x] - a;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (b0 + j < n) offs[b0 + j] = run;
        run += x[j];
    }
}
// per cube: where its triples begin and how many there are (its items are consecutive)
__global__ void __launch_bounds__(256) k_raster_query_counts(const uint32_t* __restrict__ ib, uint32_t nq, uint32_t ni, const uint64_t* __restrict__ offs,
                                                             const uint64_t* __restrict__ total, uint64_t* __restrict__ counts,
                                                             uint64_t* __restrict__ offsets) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const uint32_t a = ib[q], b = ib[q + 1];
    const uint64_t oa = a < ni ? offs[a] : *total, ob = b < ni ? offs[b] : *total;
    offsets[q] = oa;
    counts[q] = ob - oa;
}

static int raster_search_device(const dcdf_raster* r, const dcdf_cube* cubes, const int64_t* lower, const int64_t* upper, size_t nq, uint32_t* out,
                                size_t cap, int out_mem, uint64_t* counts, uint64_t* offsets, float* kernel_ms) {
    std::vector<uint32_t> sb(nq + 1), ib(nq + 1), wb(nq + 1);
    uint64_t ns = 0, ni = 0, nw = 0;
    for (size_t q = 0; q < nq; q++) {
        const dcdf_cube c = norm_cube(cubes[q]);
        if (c.end > r->T || c.bottom > r->R || c.right > r->C) return DCDF_ERR_BOUNDS;
        sb[q] = (uint32_t)ns; ib[q] = (uint32_t)ni; wb[q] = (uint32_t)nw;
        if ((uint64_t)(c.end - c.start) * (c.bottom - c.top) * (c.right - c.left) == 0) continue;
        const uint64_t tiles = (uint64_t)((c.bottom - 1) / r->tile - c.top / r->tile + 1) * ((c.right - 1) / r->tile - c.left / r->tile + 1);
        ns += ((c.end - 1) / r->cs - c.start / r->cs + 1) * tiles;
        ni += (uint64_t)(c.end - c.start) * tiles;
        nw += raster_item_count(r, c, 64);
        if (nw + 4096 > 0xffffff00ull) return DCDF_ERR_CAPACITY;
    }
    sb[nq] = (uint32_t)ns; ib[nq] = (uint32_t)ni; wb[nq] = (uint32_t)nw;
    for (size_t q = 0; q < nq; q++) counts[q] = offsets[q] = 0;
    if (ni == 0) {
        if (kernel_ms) *kernel_ms = 0.f;
        return DCDF_OK;
    }
    DevBuf d_cubes, d_lo, d_hi, d_sb, d_ib, d_wb, d_qs, d_items, d_witems, d_sx, d_wbits, d_counts, d_sums, d_total, d_offs, d_qc, d_qo, d_out;
    K2R_HIP(d_cubes.alloc(nq * sizeof(dcdf_cube)));
    K2R_HIP(d_lo.alloc(nq * 8));
    K2R_HIP(d_hi.alloc(nq * 8));
    K2R_HIP(d_sb.alloc((nq + 1) * 4));
    K2R_HIP(d_ib.alloc((nq + 1) * 4));
    K2R_HIP(d_wb.alloc((nq + 1) * 4));
    K2R_HIP(hipMemcpy(d_cubes.p, cubes, nq * sizeof(dcdf_cube), hipMemcpyHostToDevice));
    K2R_HIP(hipMemcpy(d_lo.p, lower, nq * 8, hipMemcpyHostToDevice));
    K2R_HIP(hipMemcpy(d_hi.p, upper, nq * 8, hipMemcpyHostToDevice));
    K2R_HIP(hipMemcpy(d_sb.p, sb.data(), (nq + 1) * 4, hipMemcpyHostToDevice));
    K2R_HIP(hipMemcpy(d_ib.p, ib.data(), (nq + 1) * 4, hipMemcpyHostToDevice));
    K2R_HIP(hipMemcpy(d_wb.p, wb.data(), (nq + 1) * 4, hipMemcpyHostToDevice));
    K2R_HIP(d_qs.alloc_pooled(ns * sizeof(WinQuery)));
    K2R_HIP(d_items.alloc_pooled(ni * sizeof(SearchItem)));
    K2R_HIP(d_witems.alloc_pooled(nw * sizeof(WinItem)));
    K2R_HIP(d_sx.alloc_pooled(nw * sizeof(SearchExtra)));
    K2R_HIP(d_wbits.alloc_pooled(nw * 512));  // (every word is written by the walk: nothing to clear)
    K2R_HIP(d_counts.alloc(ni * 4));
    K2R_HIP(d_offs.alloc(ni * 8));
    const uint32_t nb = (uint32_t)((ni + kScanPer - 1) / kScanPer);
    K2R_HIP(d_sums.alloc((size_t)nb * 8));
    K2R_HIP(d_total.alloc(8));
    K2R_HIP(d_qc.alloc(nq * 8));
    K2R_HIP(d_qo.alloc(nq * 8));
    EventPair ev;
    K2R_HIP(ev.create());
    const RasterGeom g{r->T, r->R, r->C, r->tile, r->cs, r->nti, r->ntj, 64u};
    const uint32_t nw32 = (uint32_t)nw, ni32 = (uint32_t)ni, nq32 = (uint32_t)nq;
    hipLaunchKernelGGL(k_raster_search_expand, dim3((nq32 + 255) / 256), dim3(256), 0, 0, d_cubes.as<dcdf_cube>(), d_lo.as<int64_t>(), d_hi.as<int64_t>(),
                       d_sb.as<uint32_t>(), d_ib.as<uint32_t>(), d_wb.as<uint32_t>(), nq32, g, r->d_quirk.as<uint8_t>(), d_qs.as<WinQuery>(),
                       d_items.as<SearchItem>(), d_witems.as<WinItem>(), d_sx.as<SearchExtra>());
    K2R_HIP(hipEventRecord(ev.e0, 0));
    if (r->all_narrow)
        hipLaunchKernelGGL((k_window_wave2<4, false, true, int32_t>), dim3(std::min<uint32_t>((nw32 + 3) / 4, 256u * 16u)), dim3(256), 0, 0,
                           r->d_refs.as<ChunkRef>(), d_witems.as<WinItem>(), nw32, d_wbits.p, (int32_t)DCDF_I64, d_sx.as<SearchExtra>());
    else
        hipLaunchKernelGGL((k_window_wave2<3, false, true>), dim3(std::min<uint32_t>((nw32 + 3) / 4, 256u * 16u)), dim3(256), 0, 0,
                           r->d_refs.as<ChunkRef>(), d_witems.as<WinItem>(), nw32, d_wbits.p, (int32_t)DCDF_I64, d_sx.as<SearchExtra>());
    hipLaunchKernelGGL(k_search_count, dim3((ni32 + 63) / 64), dim3(64), 0, 0, d_wbits.as<uint32_t>(), d_items.as<SearchItem>(), d_qs.as<WinQuery>(), ni32,
                       d_counts.as<uint32_t>());
    hipLaunchKernelGGL(k_scan_sums, dim3(nb), dim3(256), 0, 0, d_counts.as<uint32_t>(), ni32, d_sums.as<uint64_t>());
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(256), 0, 0, d_sums.as<uint64_t>(), nb, d_total.as<uint64_t>());
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(256), 0, 0, d_counts.as<uint32_t>(), ni32, d_sums.as<uint64_t>(), d_offs.as<uint64_t>());
    hipLaunchKernelGGL(k_raster_query_counts, dim3((nq32 + 255) / 256), dim3(256), 0, 0, d_ib.as<uint32_t>(), nq32, ni32, d_offs.as<uint64_t>(),
                       d_total.as<uint64_t>(), d_qc.as<uint64_t>(), d_qo.as<uint64_t>());
    K2R_HIP(hipGetLastError());
    uint64_t total = 0;
    K2R_HIP(hipMemcpy(&total, d_total.p, 8, hipMemcpyDeviceToHost));
    K2R_HIP(hipMemcpy(counts, d_qc.p, nq * 8, hipMemcpyDeviceToHost));
    K2R_HIP(hipMemcpy(offsets, d_qo.p, nq * 8, hipMemcpyDeviceToHost));
    if (total > cap) return DCDF_ERR_CAPACITY;
    const bool to_dev = out_mem == DCDF_MEM_DEVICE;
    if (total > 0) {
        if (!to_dev) K2R_HIP(d_out.alloc_pooled(total * 12));
        hipLaunchKernelGGL(k_search_emit, dim3((ni32 + 63) / 64), dim3(64), 0, 0, d_qs.as<WinQuery>(), d_items.as<SearchItem>(), ni32,
                           (const uint32_t*)nullptr, d_wbits.as<uint32_t>(), d_offs.as<uint64_t>(), to_dev ? out : d_out.as<uint32_t>());
    }
    K2R_HIP(hipEventRecord(ev.e1, 0));
    K2R_HIP(hipGetLastError());
    if (total > 0 && !to_dev) K2R_HIP(hipMemcpy(out, d_out.p, total * 12, hipMemcpyDeviceToHost));
    else K2R_HIP(hipDeviceSynchronize());
    float ms = 0.f;
    K2R_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
    if (kernel_ms) *kernel_ms = ms;
    return DCDF_OK;
}

extern "C" int dcdf_raster_search_batch(const dcdf_raster* r, const dcdf_cube* cubes, const int64_t* lower, const int64_t* upper, size_t nq,
                                        uint32_t* out, size_t cap, int out_mem, uint64_t* counts, uint64_t* offsets, float* kernel_ms) {
    if (!r || !cubes || !lower || !upper || !counts || !offsets || nq == 0 || (!out && cap)) return DCDF_ERR_BAD_ARG;
    if (out_mem != DCDF_MEM_HOST && out_mem != DCDF_MEM_DEVICE) return DCDF_ERR_BAD_ARG;
    // k = 2 chunks: pieces, items, counts, offsets and triples all stay on the device; other arities take the host-built form below
    if (r->all_node && nq <= 0x7fffffffu && !std::getenv("K2R_SEARCH_DFS") && !std::getenv("K2R_RASTER_HOST"))
        return raster_search_device(r, cubes, lower, upper, nq, out, cap, out_mem, counts, offsets, kernel_ms);
    std::vector<dcdf_chunk*> sch;
    std::vector<dcdf_cube> scube;
    std::vector<int64_t> slo, shi;
    std::vector<uint32_t> sorg, scid, first(nq + 1, 0);
    sch.reserve(2 * nq); scube.reserve(2 * nq); slo.reserve(2 * nq); shi.reserve(2 * nq); sorg.reserve(6 * nq); scid.reserve(2 * nq);
    for (size_t q = 0; q < nq; q++) {
        const dcdf_cube c = norm_cube(cubes[q]);
        if (c.end > r->T || c.bottom > r->R || c.right > r->C) return DCDF_ERR_BOUNDS;
        if ((uint64_t)(c.end - c.start) * (c.bottom - c.top) * (c.right - c.left) != 0)
            raster_pieces(r, c, [&](uint32_t cid, const dcdf_cube& l, uint32_t t0, uint32_t r0, uint32_t c0) {
                sch.push_back(r->chunks[cid]);
                scid.push_back(cid);
                scube.push_back(l);
                slo.push_back(lower[q]);
                shi.push_back(upper[q]);
                sorg.push_back(t0);
                sorg.push_back(r0);
                sorg.push_back(c0);
            });
        first[q + 1] = (uint32_t)sch.size();
    }
    for (size_t q = 0; q < nq; q++) counts[q] = offsets[q] = 0;
    if (sch.empty()) return DCDF_OK;
    std::vector<uint64_t> scnt(sch.size()), soff(sch.size());
    size_t total = 0;
    // the pieces of one query follow each other (segments, then tile rows, then tile columns) and search_impl emits in
    // query order, so a query's triples are contiguous; each is moved to raster coordinates as it is written
    const SearchCtx ctx{&r->d_refs, scid.data(), sorg.data(), r->all_node, r->all_narrow, r->all_wave};
    const int rc = search_impl(sch.data(), scube.data(), slo.data(), shi.data(), sch.size(), out, cap, scnt.data(), soff.data(), &total, kernel_ms,
                               out_mem, &ctx);
    if (rc != DCDF_OK) return rc;
    for (size_t q = 0; q < nq; q++) {
        offsets[q] = first[q] < sch.size() ? soff[first[q]] : total;
        for (uint32_t k = first[q]; k < first[q + 1]; k++) counts[q] += scnt[k];
    }
    return DCDF_OK;
}
