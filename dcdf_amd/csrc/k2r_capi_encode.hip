// k2r_capi_encode.hip -- C ABI, encode side (include/dcdf_k2r.h).  Host code only.
//
// dcdf_encoder = a device-resident encode session: tile descriptors, per-tile output slots,
// per-class work queues.  dcdf_chunk_build_batch = upload + session + fetch.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <sched.h>

#include <cstring>
#include <queue>

#include "k2r_launch.h"
#include "k2r_runtime.h"

using namespace k2r;

namespace k2r {

Runtime& Runtime::get() {
    static Runtime rt;
    static std::once_flag once;
    std::call_once(once, [] {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return;
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) != hipSuccess) return;
        rt.device = dev;
        rt.cus = p.multiProcessorCount;
        rt.name = std::string(p.gcnArchName) + " " + p.name + ", " + std::to_string(p.multiProcessorCount) + " CUs";
        rt.ok = true;
    });
    return rt;
}

static uint32_t sidelen_log2(uint32_t rows, uint32_t cols) {  // snapshot.rs:118-119 for k = 2
    uint32_t m = std::max(rows, cols);
    uint32_t lg = 0;
    while ((1u << lg) < m) lg++;
    return lg;
}
static size_t elem_size(int dtype) { return (dtype == DCDF_I32 || dtype == DCDF_F32) ? 4 : 8; }

// ---- speculative parts: which tiles of a class to encode as several work items ---------------------------------------------
// A chunk is the unit of work (its instants are sequential) and a workgroup owns a CU, so when a GPU holds only a few
// chunks per CU the end of the work queue leaves CUs idle: 384 chunks on 256 CUs take two chunk-times, not 1.5, and whatever
// the count, the last chunk popped runs a whole chunk-time after the queue is empty.  Encoding the chunks at the END of the
// (longest-first) queue in parts of decreasing length (T/2, T/4, T/8, T/8: guided self-scheduling) fills that tail with
// small items.  A part costs its instants plus a little (the wait for the first part's snapshot copy, the splice), and a
// failed speculation costs a whole re-encode, so only tiles of the last rounds are split, and only when the queue is short:
// n <= 4 * wgs.  K2R_SPLIT=0 disables, =all forces every tile (tests), =tail forces the last two rounds whatever n.
static size_t plan_split(const std::vector<uint32_t>& inst, uint32_t wgs) {
    const size_t n = inst.size();
    const char* env = std::getenv("K2R_SPLIT");
    auto splittable = [&](size_t ns) {  // the last ns tiles, shrunk to those with at least 4 instants
        while (ns > 0 && inst[n - ns] < 4) ns--;
        size_t ok = 0;
        for (size_t q = n - ns; q < n; q++) ok += inst[q] >= 4;
        return ok == ns ? ns : (size_t)0;
    };
    if (env && std::strcmp(env, "0") == 0) return 0;
    if (env && std::strcmp(env, "all") == 0) return splittable(n);
    const size_t tail = std::min<size_t>(n, 2 * (size_t)wgs);
    if (env && std::strcmp(env, "tail") == 0) return splittable(tail);
    if (n > 4 * (size_t)wgs) return 0;
    return splittable(tail);
}
// the instants at which the parts of a T-instant tile begin (after 0): T/2, then half of what is left, ... (<= 4 parts)
static std::vector<uint32_t> part_bounds(uint32_t T) {
    std::vector<uint32_t> b;
    const char* env = std::getenv("K2R_PARTS");
    const uint32_t maxp = env ? (uint32_t)std::max(2, std::min(8, std::atoi(env))) : 8u;  // (8 parts: 0.934 of linear on the N = 8 share, 4: 0.919; profiles/r04r)
    uint32_t at = 0, left = T;
    while (b.size() + 2 <= maxp && left >= 4) {
        at += (left + 1) / 2;
        left -= (left + 1) / 2;
        b.push_back(at);
    }
    return b;
}

// Splices the parts of a split tile: items[first[j]] = the tile (first part, in its own slot), the rest = its continuations
// in order.  Valid when the first part holds exactly one block and no middle part opened one (what every continuation
// assumed): the bytes are appended, block 0's count byte (chunk offset 6, block.rs:89) and n_blocks (chunk.rs:238) are
// patched, the counters added.  Otherwise ST_RESPLIT (the host re-encodes the tile whole); an error of any part is the
// tile's error.
constexpr int kMaxParts = 8;
// COPY = false (inside dcdf_encoder_run, timed): the checks, the totals and the two header patches -- the continuations' bytes
// stay in their own slots.  COPY = true (materialize(), the first time somebody asks for the bytes: gather, download, fetch, the
// device pointer, hashing): the bytes are appended.  The copy is 250 MB per launch for the N = 8 share of configs[3], 7 % of
// that launch, and a gather copies every byte again anyway.  TileResult::_pad2 of the first part keeps its own length between
// the two.
template <bool COPY>
__global__ void __launch_bounds__(1024) k_stitch(const uint32_t* __restrict__ first, const uint32_t* __restrict__ items,
                                                  const TileArgs* __restrict__ args, TileResult* __restrict__ res) {
    __shared__ int32_t s_st;
    __shared__ uint64_t s_at[kMaxParts + 1];
    const uint32_t i0 = first[blockIdx.x], np = first[blockIdx.x + 1] - i0;
    const uint32_t a = items[i0];
    if (threadIdx.x == 0) {
        int32_t st = ST_OK;
        uint64_t total = COPY ? (uint64_t)res[a]._pad2 : res[a].len;
        if (res[a].status != ST_OK) st = res[a].status;
        else if (!COPY && res[a].snapshots != 1) st = ST_RESPLIT;
        for (uint32_t p = 1; p < np && st == ST_OK; p++) {
            const uint32_t b = items[i0 + p];
            if (res[b].status != ST_OK) st = res[b].status;
            else if (!COPY && p + 1 < np && res[b].snapshots != 0) st = ST_RESPLIT;
            s_at[p] = total;
            total += res[b].len;
        }
        if (st == ST_OK && total > args[a].out_cap) st = ST_OUT_CAPACITY;
        s_at[np] = total;
        s_st = st;
    }
    __syncthreads();
    const int32_t st = s_st;
    if (COPY && st == ST_OK) {
        // 16 bytes per thread and step (a source slot is 256-byte aligned; the destination is wherever the previous part ended:
        // global memory takes unaligned vector stores), then the tail
        typedef uint32_t u4 __attribute__((ext_vector_type(4), aligned(1)));
        typedef uint32_t u4a __attribute__((ext_vector_type(4)));
        for (uint32_t p = 1; p < np; p++) {
            const uint32_t b = items[i0 + p];
            uint8_t* dst = args[a].out + s_at[p];
            const uint8_t* src = args[b].out;
            const uint64_t lb = s_at[p + 1] - s_at[p], nv = lb / 16;
            for (uint64_t i = threadIdx.x; i < nv; i += blockDim.x) *(u4*)(dst + 16 * i) = *(const u4a*)(src + 16 * i);
            for (uint64_t i = 16 * nv + threadIdx.x; i < lb; i += blockDim.x) dst[i] = src[i];
        }
    }
    if (!COPY && threadIdx.x == 0) {
        if (st == ST_OK) {
            const uint32_t last = items[i0 + np - 1];
            args[a].out[6] = (uint8_t)res[last].carry_count;
            store_be32(args[a].out + 2, 1u + res[last].snapshots);
            res[a]._pad2 = (uint32_t)res[a].len;  // (a slot holds far less than 4 GB)
            res[a].len = s_at[np];
            for (uint32_t p = 1; p < np; p++) {
                const uint32_t b = items[i0 + p];
                res[a].snapshots += res[b].snapshots;
                res[a].logs += res[b].logs;
                res[a].stash_logs += res[b].stash_logs;
            }
        } else {
            res[a].status = st;
            res[a].len = 0;
        }
    }
}

// Upper bound of one serialized instant (all values 4 bytes): used when the default slot overflows.
static uint64_t worst_instant_bytes(uint32_t lg) {
    const uint64_t maxv = ((1ull << (2 * (lg + 1))) - 1) / 3, maxt = ((1ull << (2 * lg)) - 1) / 3;
    auto bm = [](uint64_t n) { return 8 + 4 * (n / 128) + 4 * ((n + 31) / 32); };
    return 13 + 2 * bm(maxt) + (1 + 4 * (bm(maxv) + maxv)) + (1 + 4 * (bm(maxt) + maxt));
}

}  // namespace k2r

struct dcdf_encoder {
    std::vector<dcdf_tile_desc> desc;
    std::vector<TileArgs> args;        // host mirror of the device array
    std::vector<TileResult> results;   // host copy after run
    std::vector<int32_t> pre_status;   // host-side validation result per tile
    std::vector<EncClass> classes;
    std::vector<std::vector<uint32_t>> class_tiles;
    std::vector<uint64_t> slot_off, slot_cap;
    DevBuf d_args, d_results, d_out, d_minmax, d_order, d_queue, d_lists;
    std::vector<size_t> order_off;     // per class offset into d_order
    std::vector<size_t> lists_off;     // per class offset (u64 words) into d_lists
    std::vector<uint32_t> grid;
    std::vector<std::unique_ptr<DevBuf>> retry_slots;  // bigger slots for tiles that overflowed
    // Speculative parts (k2r_encode.h): tiles encoded as several work items, the tile's own TileArgs restricted to the first
    // instants and extra TileArgs at indices >= n for the rest, spliced by k_stitch inside the timed region.
    std::vector<std::vector<uint32_t>> class_items;  // per class: the launch order (tile indices and extra indices >= n)
    std::vector<std::vector<uint32_t>> split;        // per split tile: {tile, continuation items in order}
    bool spliced = true;                             // false between a run and the first request for a split tile's bytes
    std::vector<uint32_t> splice_first, splice_items;  // the split tiles whose parts checked out in the last run (k_stitch<false>):
                                                       // only those are appended by materialize() -- a tile that was re-encoded
                                                       // whole afterwards (retry, universal kernel) already has all its bytes
    DevBuf d_part_first, d_part_items, d_out_b, d_flags, d_shared;
    // tiles outside the fused kernel's contract (k != 2, sidelen < 8 or > 256): encoded by the universal kernel
    // (k2r_generic.hip); key = k << 8 | H
    std::vector<std::pair<uint32_t, std::vector<uint32_t>>> generic_groups;
    std::vector<uint8_t> is_generic;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint64_t minmax_total = 0;
    std::vector<uint64_t> minmax_off;
    ~dcdf_encoder() {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream) StreamPool::get().give(false, stream);
    }
};

// depth and sidelen of a tile's tree, by the reference's own f64 formula (k2r_runtime.h ref_levels)
static uint32_t depth_for(uint32_t rows, uint32_t cols, uint32_t k, uint64_t* sidelen) {
    const uint64_t m = std::max(rows, cols);
    *sidelen = ref_sidelen(m, k);
    return ref_levels(m, k);
}
constexpr uint64_t kGenericMaxSidelen = 1024;  // universal kernel: level arrays in HBM scratch, ~77 B per node

// returns DCDF_OK and either a fused-kernel class (*generic_key == 0) or the universal kernel's key (k << 8 | H)
static int validate_tile(const dcdf_tile_desc& t, int k, EncClass* cls, uint32_t* generic_key) {
    *generic_key = 0;
    if (!t.base || t.instants == 0 || t.rows == 0 || t.cols == 0) return DCDF_ERR_BAD_ARG;
    if (t.dtype != DCDF_I32 && t.dtype != DCDF_I64 && t.dtype != DCDF_F32 && t.dtype != DCDF_F64) return DCDF_ERR_BAD_ARG;
    if ((t.dtype == DCDF_F32 || t.dtype == DCDF_F64) && t.fractional_bits > 62) return DCDF_ERR_BAD_ARG;
    if (k < 2 || k > 16) return DCDF_ERR_BAD_ARG;
    const uint32_t lg = sidelen_log2(t.rows, t.cols);
    if (k != 2 || lg < 4 || lg > 8) {  // outside the fused kernel (k = 2, sidelen 16 .. 256): any k, sidelen 1 .. 1024
        uint64_t side;
        const uint32_t H = depth_for(t.rows, t.cols, (uint32_t)k, &side);
        if (side > kGenericMaxSidelen) return DCDF_ERR_UNSUPPORTED;
        *generic_key = ((uint32_t)k << 8) | H;
        return DCDF_OK;
    }
    const uint32_t S = 1u << lg;
    cls->log2s = (int)lg;
    cls->padded = t.rows != S || t.cols != S;
    // 16-byte row loads: unit column stride, rows and instants 16-byte aligned, 32-bit byte offsets inside an instant
    const uint64_t esz = (t.dtype == DCDF_I64 || t.dtype == DCDF_F64) ? 8 : 4;
    const int64_t al = (int64_t)(16 / esz);
    const bool rows16 = !cls->padded && t.stride_c == 1 && (t.stride_r % al) == 0 && t.stride_r > 0 && (t.stride_t % al) == 0 &&
                        ((uintptr_t)t.base % 16) == 0 &&
                        ((uint64_t)(t.rows - 1) * (uint64_t)t.stride_r + t.cols) * esz < (1ull << 31);
    cls->vec = !rows16 ? 0 : (t.dtype == DCDF_I32 ? 1 : (t.dtype == DCDF_F32 ? 2 : (t.dtype == DCDF_I64 ? 3 : 4)));
    return DCDF_OK;
}

extern "C" int dcdf_encoder_create(const dcdf_tile_desc* tiles, size_t n, int k, size_t out_cap_per_tile,
                                   dcdf_encoder** enc_out) {
    if (!tiles || !enc_out || n == 0 || n > 0x7fffffffu) return DCDF_ERR_BAD_ARG;
    Runtime& rt = Runtime::get();
    if (!rt.ok) return DCDF_ERR_NO_DEVICE;
    std::unique_ptr<dcdf_encoder> e(new (std::nothrow) dcdf_encoder());
    if (!e) return DCDF_ERR_NOMEM;
    e->desc.assign(tiles, tiles + n);
    e->args.resize(n);
    e->results.resize(n);
    e->pre_status.resize(n);
    e->slot_off.resize(n);
    e->slot_cap.resize(n);
    e->minmax_off.resize(n);
    e->is_generic.assign(n, 0);
    uint64_t out_total = 0, mm_total = 0;
    for (size_t i = 0; i < n; i++) {
        const dcdf_tile_desc& t = tiles[i];
        EncClass cls{};
        uint32_t gkey = 0;
        const int st = validate_tile(t, k, &cls, &gkey);
        e->pre_status[i] = st;
        e->results[i] = TileResult{};
        if (st != DCDF_OK) continue;
        if (gkey) {
            e->is_generic[i] = 1;
            size_t gi = 0;
            for (; gi < e->generic_groups.size(); gi++)
                if (e->generic_groups[gi].first == gkey) break;
            if (gi == e->generic_groups.size()) e->generic_groups.emplace_back(gkey, std::vector<uint32_t>());
            e->generic_groups[gi].second.push_back((uint32_t)i);
        } else {
            size_t ci = 0;
            for (; ci < e->classes.size(); ci++)
                if (e->classes[ci] == cls) break;
            if (ci == e->classes.size()) {
                e->classes.push_back(cls);
                e->class_tiles.emplace_back();
            }
            e->class_tiles[ci].push_back((uint32_t)i);
        }
        uint64_t cap = out_cap_per_tile ? out_cap_per_tile : (uint64_t)t.instants * t.rows * t.cols * 4 + 4096;
        cap = (cap + 255) & ~255ull;
        e->slot_off[i] = out_total;
        e->slot_cap[i] = cap;
        out_total += cap;
        e->minmax_off[i] = mm_total;
        mm_total += 2ull * t.instants;
    }
    e->minmax_total = mm_total;
    K2R_HIP(StreamPool::get().take(false, &e->stream));
    K2R_HIP(hipEventCreate(&e->ev0));
    K2R_HIP(hipEventCreate(&e->ev1));
    K2R_HIP(e->d_out.alloc_pooled(out_total));
    K2R_HIP(e->d_minmax.alloc(mm_total * 8));
    K2R_HIP(e->d_args.alloc(n * sizeof(TileArgs)));
    K2R_HIP(e->d_results.alloc(n * sizeof(TileResult)));
    // diagnostic knob (A/B runs): cap the LDS words the log stash may use; 1 = always take the re-reading passes
    uint32_t stash_words = 0;
    if (const char* sw = std::getenv("K2R_STASH_WORDS")) stash_words = (uint32_t)std::atoi(sw);
    for (size_t i = 0; i < n; i++) {
        const dcdf_tile_desc& t = tiles[i];
        TileArgs a{};
        a.stash_words = stash_words;
        a.base = t.base;
        a.st = t.stride_t; a.sr = t.stride_r; a.sc = t.stride_c;
        a.instants = t.instants; a.rows = t.rows; a.cols = t.cols;
        a.dtype = t.dtype;
        a.fbits = (t.dtype == DCDF_F32 || t.dtype == DCDF_F64) ? t.fractional_bits : 0;  // mmbuffer.rs:397-403
        a.round = t.round;
        a.out = e->d_out.as<uint8_t>() + e->slot_off[i];
        a.out_cap = e->slot_cap[i];
        a.minmax = e->d_minmax.as<int64_t>() + e->minmax_off[i];
        e->args[i] = a;
    }
    // per-class launch geometry; which tiles are encoded in speculative parts (plan_split)
    size_t order_total = 0, lists_total = 0;
    uint64_t out_b_total = 0, shared_total = 0;
    std::vector<uint64_t> out_b_off, shared_off;
    e->class_items.resize(e->classes.size());
    for (size_t ci = 0; ci < e->classes.size(); ci++) {
        const EncClass& c = e->classes[ci];
        auto& v = e->class_tiles[ci];
        // longest chunks first: the tail of the work queue is then made of the short ones
        std::stable_sort(v.begin(), v.end(), [&](uint32_t a, uint32_t b) { return tiles[a].instants > tiles[b].instants; });
        const int per_cu = encode_blocks_per_cu(c);
        uint32_t wgs = (uint32_t)((uint64_t)rt.cus * (uint64_t)per_cu);
        if (const char* mw = std::getenv("K2R_MAX_WGS")) wgs = std::min<uint32_t>(wgs, (uint32_t)std::max(1, std::atoi(mw)));  // diagnostics
        std::vector<uint32_t> inst(v.size());
        for (size_t q = 0; q < v.size(); q++) inst[q] = tiles[v[q]].instants;
        const size_t ns = plan_split(inst, std::max(1u, wgs));  // the LAST ns tiles of v are split
        struct Item { uint32_t idx, cost, cont; };
        std::vector<Item> items;
        const uint64_t cmp_bytes = ((2ull << (2 * c.log2s)) + 255) & ~255ull;  // 2 bytes per cell of the padded tile
        for (size_t q = 0; q < v.size(); q++) {
            const uint32_t ti = v[q], T = tiles[ti].instants;
            if (q + ns < v.size()) {
                items.push_back({ti, T, 0});
                continue;
            }
            const std::vector<uint32_t> at = part_bounds(T);
            std::vector<uint32_t> parts{ti};
            e->args[ti].inst_end = at[0];
            items.push_back({ti, at[0], 0});
            uint32_t prev_cost = at[0];
            shared_off.push_back(shared_total);
            shared_total += cmp_bytes;
            for (size_t p = 0; p < at.size(); p++) {
                const uint32_t m0 = at[p], m1 = p + 1 < at.size() ? at[p + 1] : T, bi = (uint32_t)e->args.size();
                TileArgs b = e->args[ti];
                b.inst_begin = m0;
                b.inst_end = m1;
                b.out_cap = ((e->slot_cap[ti] * (m1 - m0) + T - 1) / T + 4096 + 255) & ~255ull;
                out_b_off.push_back(out_b_total);
                out_b_total += b.out_cap;
                e->args.push_back(b);
                parts.push_back(bi);
                prev_cost = std::min(prev_cost, m1 - m0);  // (never sorted before the part it follows)
                items.push_back({bi, prev_cost, 1});
            }
            e->split.push_back(std::move(parts));
        }
        // longest first; at equal length a first part before any continuation -- with first parts at least as long as their
        // continuations, every continuation is popped after the part it waits for (k2r_encode.h)
        std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.cost != b.cost ? a.cost > b.cost : a.cont < b.cont; });
        for (const Item& it : items) e->class_items[ci].push_back(it.idx);
        const uint32_t g = (uint32_t)std::min<uint64_t>(items.size(), wgs);
        e->grid.push_back(std::max(1u, g));
        e->order_off.push_back(order_total);
        order_total += items.size();
        e->lists_off.push_back(lists_total);
        lists_total += (size_t)e->grid.back() * encode_list_words(c);
    }
    if (!e->split.empty()) {
        K2R_HIP(e->d_out_b.alloc_pooled(out_b_total));
        K2R_HIP(e->d_shared.alloc_pooled(shared_total));
        K2R_HIP(e->d_flags.alloc(e->split.size() * 16));
        std::vector<uint32_t> first{0}, flat;
        size_t nb = 0;
        for (size_t j = 0; j < e->split.size(); j++) {
            const auto& parts = e->split[j];
            for (size_t p = 0; p < parts.size(); p++) {
                TileArgs& a = e->args[parts[p]];
                a.shared_flag = e->d_flags.as<uint32_t>() + 4 * j;
                a.shared_cmp = (uint32_t*)(e->d_shared.as<uint8_t>() + shared_off[j]);
                if (p > 0) a.out = e->d_out_b.as<uint8_t>() + out_b_off[nb++];
                flat.push_back(parts[p]);
            }
            first.push_back((uint32_t)flat.size());
        }
        K2R_HIP(e->d_part_first.alloc(first.size() * 4));
        K2R_HIP(e->d_part_items.alloc(flat.size() * 4));
        K2R_HIP(hipMemcpy(e->d_part_first.p, first.data(), first.size() * 4, hipMemcpyHostToDevice));
        K2R_HIP(hipMemcpy(e->d_part_items.p, flat.data(), flat.size() * 4, hipMemcpyHostToDevice));
        // the item tables grew
        e->d_args.release();
        e->d_results.release();
        K2R_HIP(e->d_args.alloc(e->args.size() * sizeof(TileArgs)));
        K2R_HIP(e->d_results.alloc(e->args.size() * sizeof(TileResult)));
    }
    K2R_HIP(e->d_order.alloc(std::max<size_t>(order_total, 1) * 4));
    K2R_HIP(e->d_queue.alloc(std::max<size_t>(e->classes.size(), 1) * 4));
    K2R_HIP(e->d_lists.alloc_pooled(std::max<size_t>(lists_total, 1) * 8));
    for (size_t ci = 0; ci < e->classes.size(); ci++)
        K2R_HIP(hipMemcpy(e->d_order.as<uint32_t>() + e->order_off[ci], e->class_items[ci].data(), e->class_items[ci].size() * 4,
                          hipMemcpyHostToDevice));
    K2R_HIP(hipMemcpy(e->d_args.p, e->args.data(), e->args.size() * sizeof(TileArgs), hipMemcpyHostToDevice));
    *enc_out = e.release();
    return DCDF_OK;
}

static int run_classes(dcdf_encoder* e, const std::vector<std::vector<uint32_t>>* subset, float* kernel_ms) {
    // subset == nullptr: all tiles of every class (order already on the device)
    K2R_HIP(hipMemsetAsync(e->d_queue.p, 0, e->d_queue.bytes, e->stream));
    if (!subset && !e->split.empty()) K2R_HIP(hipMemsetAsync(e->d_flags.p, 0, e->d_flags.bytes, e->stream));
    K2R_HIP(hipEventRecord(e->ev0, e->stream));
    for (size_t ci = 0; ci < e->classes.size(); ci++) {
        uint32_t nt = (uint32_t)e->class_items[ci].size();
        if (subset) {
            nt = (uint32_t)(*subset)[ci].size();
            if (nt == 0) continue;
            K2R_HIP(hipMemcpyAsync(e->d_order.as<uint32_t>() + e->order_off[ci], (*subset)[ci].data(), nt * 4ull,
                                   hipMemcpyHostToDevice, e->stream));
        }
        EncodeLaunch L{};
        L.tiles = e->d_args.as<TileArgs>();
        L.results = e->d_results.as<TileResult>();
        L.order = e->d_order.as<uint32_t>() + e->order_off[ci];
        L.n = nt;
        L.queue = e->d_queue.as<uint32_t>() + ci;
        L.lists = e->d_lists.as<uint64_t>() + e->lists_off[ci];
        L.grid = std::min(e->grid[ci], std::max(1u, nt));
        K2R_HIP(launch_encode(e->classes[ci], L, e->stream));
    }
    if (!subset && !e->split.empty()) {  // check and account the speculative parts (inside the timed region); bytes: materialize()
        hipLaunchKernelGGL(k2r::k_stitch<false>, dim3((uint32_t)e->split.size()), dim3(1024), 0, e->stream, e->d_part_first.as<uint32_t>(),
                           e->d_part_items.as<uint32_t>(), e->d_args.as<TileArgs>(), e->d_results.as<TileResult>());
        e->spliced = false;
    }
    K2R_HIP(hipEventRecord(e->ev1, e->stream));
    K2R_HIP(hipStreamSynchronize(e->stream));
    if (kernel_ms) K2R_HIP(hipEventElapsedTime(kernel_ms, e->ev0, e->ev1));
    return DCDF_OK;
}

// The universal kernel over `tiles` (all of one (k, H)); tiles whose slot is too small are re-run into exact-size slots.
static int run_generic(dcdf_encoder* e, uint32_t key, const std::vector<uint32_t>& tiles, float* ms_acc) {
    if (tiles.empty()) return DCDF_OK;
    Runtime& rt = Runtime::get();
    const uint32_t k = key >> 8, H = key & 0xffu;
    const uint64_t per_wg = (generic_scratch_bytes(k, H) + 255) & ~255ull;
    const uint64_t budget = 8ull << 30;
    const size_t n = e->desc.size();
    std::vector<uint32_t> todo(tiles);
    for (int pass = 0; pass < 2 && !todo.empty(); pass++) {
        const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)todo.size(), (uint64_t)rt.cus, budget / per_wg}));
        DevBuf d_scratch, d_order, d_queue;
        K2R_HIP(d_scratch.alloc((size_t)grid * per_wg));
        K2R_HIP(d_order.alloc(todo.size() * 4));
        K2R_HIP(d_queue.alloc(4));
        K2R_HIP(hipMemcpyAsync(d_order.p, todo.data(), todo.size() * 4, hipMemcpyHostToDevice, e->stream));
        K2R_HIP(hipMemsetAsync(d_queue.p, 0, 4, e->stream));
        EncodeLaunch L{};
        L.tiles = e->d_args.as<TileArgs>();
        L.results = e->d_results.as<TileResult>();
        L.order = d_order.as<uint32_t>();
        L.n = (uint32_t)todo.size();
        L.queue = d_queue.as<uint32_t>();
        L.lists = nullptr;
        L.grid = grid;
        K2R_HIP(hipEventRecord(e->ev0, e->stream));
        K2R_HIP(launch_encode_generic(L, d_scratch.as<uint8_t>(), per_wg, k, H, e->stream));
        K2R_HIP(hipEventRecord(e->ev1, e->stream));
        K2R_HIP(hipStreamSynchronize(e->stream));
        float ms = 0.f;
        K2R_HIP(hipEventElapsedTime(&ms, e->ev0, e->ev1));
        if (ms_acc) *ms_acc += ms;
        K2R_HIP(hipMemcpy(e->results.data(), e->d_results.p, n * sizeof(TileResult), hipMemcpyDeviceToHost));
        std::vector<uint32_t> again;
        for (uint32_t ti : todo)
            if (e->results[ti].status == ST_OUT_CAPACITY && pass == 0) {  // res.len = the bytes it needs
                const uint64_t cap = (e->results[ti].len + 255) & ~255ull;
                std::unique_ptr<DevBuf> b(new DevBuf());
                K2R_HIP(b->alloc(cap));
                e->args[ti].out = b->as<uint8_t>();
                e->args[ti].out_cap = cap;
                e->retry_slots.push_back(std::move(b));
                K2R_HIP(hipMemcpy(e->d_args.as<TileArgs>() + ti, &e->args[ti], sizeof(TileArgs), hipMemcpyHostToDevice));
                again.push_back(ti);
            }
        todo.swap(again);
    }
    return DCDF_OK;
}

extern "C" int dcdf_encoder_run(dcdf_encoder* e, float* kernel_ms) {
    if (!e) return DCDF_ERR_BAD_ARG;
    const size_t n = e->desc.size();
    int rc = run_classes(e, nullptr, kernel_ms);
    if (rc != DCDF_OK) return rc;
    K2R_HIP(hipMemcpy(e->results.data(), e->d_results.p, n * sizeof(TileResult), hipMemcpyDeviceToHost));
    e->splice_first.assign(1, 0u);
    e->splice_items.clear();
    for (const auto& parts : e->split)
        if (e->results[parts[0]].status == ST_OK) {
            for (uint32_t it : parts) e->splice_items.push_back(it);
            e->splice_first.push_back((uint32_t)e->splice_items.size());
        }
    if (std::getenv("K2R_PROFILE_PRINT")) {
        uint64_t acc[NPROF] = {0};
        for (size_t i = 0; i < n; i++)
            for (int k = 0; k < NPROF; k++) acc[k] += e->results[i].prof[k];
        uint64_t tot = 0;
        for (int k = 0; k < NPROF; k++) tot += acc[k];
        static const char* names[NPROF] = {"p1 load+analysis", "p2 top", "p3 own/top nodes", "reduce4", "clear+hdr",
                                           "emit C snapshot", "emit C/Q log", "T/eqB bitmaps", "Lmax dac | V0,M0 bitmaps", "Lmin dac | zero bitmaps",
                                           "emit pass A", "emit pass B/I", "sizes+heuristic", "winner scan", "byte-1 pass A", "fast: phase 1",
                                           "fast: wait p1", "fast: top+plan", "fast: copy-out", "fast: bitmaps"};
        for (int k = 0; k < NPROF; k++)
            std::fprintf(stderr, "k2r-prof %-18s %14llu cyc %5.1f%%\n", names[k], (unsigned long long)acc[k],
                         tot ? 100.0 * (double)acc[k] / (double)tot : 0.0);
#ifdef K2R_PROFILE
        // Per WAVE (k2r_exec.h): cycles of a wave's own work and cycles it was parked at barriers, per phase, summed over all
        // tiles.  "max" is the busiest wave's work -- what the phase costs the workgroup --, "mean" the average wave's; the
        // difference is barrier wait caused by imbalance, and "wait" is what the average wave actually spent parked.
        {
            static uint64_t w[16][NPROF][2];
            std::memset(w, 0, sizeof(w));
            for (size_t i = 0; i < n; i++)
                for (int v = 0; v < 16; v++)
                    for (int k = 0; k < NPROF; k++) {
                        w[v][k][0] += e->results[i].pw[v][k][0];
                        w[v][k][1] += e->results[i].pw[v][k][1];
                    }
            int nw = 0;
            for (int v = 0; v < 16; v++) {
                uint64_t t = 0;
                for (int k = 0; k < NPROF; k++) t += w[v][k][0] + w[v][k][1];
                if (t) nw = v + 1;
            }
            uint64_t insts = 0;
            for (size_t i = 0; i < n; i++) insts += e->results[i].snapshots + e->results[i].logs;
            double all = 0;
            for (int k = 0; k < NPROF; k++)
                for (int v = 0; v < nw; v++) all += (double)(w[v][k][0] + w[v][k][1]);
            all /= nw ? nw : 1;
            std::fprintf(stderr, "k2r-pw %d waves, %llu chunk-instants, %.0f cycles per chunk-instant and wave (work + wait)\n", nw,
                         (unsigned long long)insts, insts ? all / (double)insts : 0.0);
            std::fprintf(stderr, "k2r-pw %-24s %9s %9s %9s %9s %7s   (cycles per chunk-instant)\n", "phase", "work-mean", "work-max", "work-w0",
                         "wait-mean", "share");
            for (int k = 0; k < NPROF; k++) {
                double mean = 0, mx = 0, wait = 0;
                for (int v = 0; v < nw; v++) {
                    mean += (double)w[v][k][0];
                    wait += (double)w[v][k][1];
                    mx = std::max(mx, (double)w[v][k][0]);
                }
                if (mean + wait == 0) continue;
                mean /= nw;
                wait /= nw;
                const double d = insts ? (double)insts : 1.0;
                std::fprintf(stderr, "k2r-pw %-24s %9.0f %9.0f %9.0f %9.0f %6.1f%%\n", names[k], mean / d, mx / d, (double)w[0][k][0] / d, wait / d,
                             all ? 100.0 * (mean + wait) / all : 0.0);
            }
        }
#endif
        // per-tile totals: how uneven are the chunks?  (cycles per instant, logs that did not come from the stash)
        std::vector<double> per;
        uint64_t logs = 0, stash = 0;
        for (size_t i = 0; i < n; i++) {
            uint64_t t = 0;
            for (int k = 0; k < NPROF; k++) t += e->results[i].prof[k];
            const uint32_t ni = e->results[i].snapshots + e->results[i].logs;
            if (ni) per.push_back((double)t / ni);
            logs += e->results[i].logs;
            stash += e->results[i].stash_logs;
        }
        std::sort(per.begin(), per.end());
        if (!per.empty()) {
            double sum = 0;
            for (double x : per) sum += x;
            std::fprintf(stderr, "k2r-prof cycles/instant per tile: min %.0f p50 %.0f mean %.0f p90 %.0f p99 %.0f max %.0f | logs %llu from stash %llu\n",
                         per.front(), per[per.size() / 2], sum / per.size(), per[per.size() * 9 / 10], per[per.size() * 99 / 100],
                         per.back(), (unsigned long long)logs, (unsigned long long)stash);
        }
    }
    for (size_t i = 0; i < n; i++)
        if (e->results[i].status == ST_INTERNAL) {
            const uint32_t* d = e->results[i].dbg;
            std::fprintf(stderr, "dcdf_k2r: internal guard tripped on tile %zu: guard bitmask=0x%08x (bit = kGuard* in k2r_encode.h)\n",
                         i, d[0]);
        }
    // Tiles whose slot was too small: re-encode them alone into worst-case slots (rare; not timed).  Tiles whose
    // speculative parts did not splice: re-encode them whole (which may then turn out too big for the slot: second pass).
    for (int pass = 0; pass < 2; pass++) {
    std::vector<std::vector<uint32_t>> again(e->classes.size());
    bool any = false, unsplit = false;
    for (size_t ci = 0; ci < e->classes.size(); ci++)
        for (uint32_t ti : e->class_tiles[ci])
            if (e->results[ti].status == ST_OUT_CAPACITY || e->results[ti].status == ST_RESPLIT) {
                if (e->results[ti].status == ST_OUT_CAPACITY) {
                    const dcdf_tile_desc& t = e->desc[ti];
                    const uint64_t cap = 6 + (uint64_t)t.instants * (1 + worst_instant_bytes((uint32_t)e->classes[ci].log2s));
                    std::unique_ptr<DevBuf> b(new DevBuf());
                    K2R_HIP(b->alloc(cap));
                    e->args[ti].out = b->as<uint8_t>();
                    e->args[ti].out_cap = cap;
                    e->retry_slots.push_back(std::move(b));
                }
                // (a tile that had been split is encoded whole this time -- and from now on: a block boundary in its first
                // half, or a slot too small, would come back on every run)
                const bool was_split = e->args[ti].inst_end != 0;
                e->args[ti].inst_end = 0;
                e->args[ti].shared_flag = nullptr;
                e->args[ti].shared_cmp = nullptr;
                unsplit = unsplit || was_split;
                K2R_HIP(hipMemcpy(e->d_args.as<TileArgs>() + ti, &e->args[ti], sizeof(TileArgs), hipMemcpyHostToDevice));
                again[ci].push_back(ti);
                any = true;
            }
    if (any) {
        rc = run_classes(e, &again, nullptr);
        if (rc != DCDF_OK) return rc;
        K2R_HIP(hipMemcpy(e->results.data(), e->d_results.p, n * sizeof(TileResult), hipMemcpyDeviceToHost));
        if (unsplit) {  // drop the parts (extra items) of the tiles that are whole from now on
            std::vector<std::vector<uint32_t>> keep_split;
            std::vector<uint8_t> gone(e->args.size(), 0);
            std::vector<uint32_t> first{0}, flat;
            for (auto& parts : e->split) {
                if (e->args[parts[0]].inst_end == 0) {
                    for (size_t p = 1; p < parts.size(); p++) gone[parts[p]] = 1;
                    continue;
                }
                for (uint32_t it : parts) flat.push_back(it);
                first.push_back((uint32_t)flat.size());
                keep_split.push_back(std::move(parts));
            }
            e->split.swap(keep_split);
            if (!flat.empty()) {
                K2R_HIP(hipMemcpy(e->d_part_first.p, first.data(), first.size() * 4, hipMemcpyHostToDevice));
                K2R_HIP(hipMemcpy(e->d_part_items.p, flat.data(), flat.size() * 4, hipMemcpyHostToDevice));
            }
            for (auto& items : e->class_items) {
                std::vector<uint32_t> keep;
                for (uint32_t it : items)
                    if (!gone[it]) keep.push_back(it);
                items.swap(keep);
            }
        }
        // restore the full order lists for a later run()
        for (size_t ci = 0; ci < e->classes.size(); ci++)
            K2R_HIP(hipMemcpy(e->d_order.as<uint32_t>() + e->order_off[ci], e->class_items[ci].data(),
                              e->class_items[ci].size() * 4, hipMemcpyHostToDevice));
    }
    if (!any) break;
    }
    // The universal kernel: tiles outside the fused kernel's shapes, then the tiles the fused kernel declined at run time
    // (a stored value beyond its 2^30 contract: ST_UNSUPPORTED) -- same slot, retried with an exact-size one if need be.
    float gen_ms = 0.f;
    for (const auto& g : e->generic_groups) {
        rc = run_generic(e, g.first, g.second, &gen_ms);
        if (rc != DCDF_OK) return rc;
    }
    std::vector<std::pair<uint32_t, std::vector<uint32_t>>> declined;
    for (size_t ci = 0; ci < e->classes.size(); ci++)
        for (uint32_t ti : e->class_tiles[ci])
            if (e->results[ti].status == ST_UNSUPPORTED) {
                const uint32_t key = (2u << 8) | (uint32_t)e->classes[ci].log2s;
                size_t gi = 0;
                for (; gi < declined.size(); gi++)
                    if (declined[gi].first == key) break;
                if (gi == declined.size()) declined.emplace_back(key, std::vector<uint32_t>());
                declined[gi].second.push_back(ti);
            }
    for (const auto& g : declined) {
        rc = run_generic(e, g.first, g.second, &gen_ms);
        if (rc != DCDF_OK) return rc;
    }
    if (kernel_ms) *kernel_ms += gen_ms;
    return DCDF_OK;
}

// The bytes of the tiles that were encoded in parts, made contiguous in the first part's slot (k_stitch<true>): once per run,
// the first time they are asked for.
static int materialize(dcdf_encoder* e) {
    if (e->spliced || e->splice_items.empty()) {
        e->spliced = true;
        return DCDF_OK;
    }
    DevBuf d_first, d_items;
    K2R_HIP(d_first.alloc(e->splice_first.size() * 4));
    K2R_HIP(d_items.alloc(e->splice_items.size() * 4));
    K2R_HIP(hipMemcpyAsync(d_first.p, e->splice_first.data(), e->splice_first.size() * 4, hipMemcpyHostToDevice, e->stream));
    K2R_HIP(hipMemcpyAsync(d_items.p, e->splice_items.data(), e->splice_items.size() * 4, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k2r::k_stitch<true>, dim3((uint32_t)e->splice_first.size() - 1), dim3(1024), 0, e->stream, d_first.as<uint32_t>(),
                       d_items.as<uint32_t>(), e->d_args.as<TileArgs>(), e->d_results.as<TileResult>());
    K2R_HIP(hipGetLastError());
    K2R_HIP(hipStreamSynchronize(e->stream));
    e->spliced = true;
    return DCDF_OK;
}

extern "C" int dcdf_encoder_result(dcdf_encoder* e, size_t i, int32_t* status, uint64_t* len, uint32_t* snapshots,
                                   uint32_t* logs, const uint8_t** device_bytes) {
    if (!e || i >= e->desc.size()) return DCDF_ERR_BAD_ARG;
    if (device_bytes) {
        const int mrc = materialize(e);
        if (mrc != DCDF_OK) return mrc;
    }
    const bool pre_ok = e->pre_status[i] == DCDF_OK;
    const TileResult& r = e->results[i];
    const int st = pre_ok ? map_status(r.status) : e->pre_status[i];
    if (status) *status = st;
    if (len) *len = st == DCDF_OK ? r.len : 0;
    if (snapshots) *snapshots = st == DCDF_OK ? r.snapshots : 0;
    if (logs) *logs = st == DCDF_OK ? r.logs : 0;
    if (device_bytes) *device_bytes = st == DCDF_OK ? e->args[i].out : nullptr;
    return DCDF_OK;
}

extern "C" int dcdf_encoder_fetch(dcdf_encoder* e, size_t i, uint8_t* dst, size_t cap) {
    if (!e || i >= e->desc.size() || !dst) return DCDF_ERR_BAD_ARG;
    if (e->pre_status[i] != DCDF_OK) return e->pre_status[i];
    const TileResult& r = e->results[i];
    if (r.status != ST_OK) return map_status(r.status);
    if (cap < r.len) return DCDF_ERR_CAPACITY;
    {
        const int mrc = materialize(e);
        if (mrc != DCDF_OK) return mrc;
    }
    K2R_HIP(hipMemcpy(dst, e->args[i].out, r.len, hipMemcpyDeviceToHost));
    return DCDF_OK;
}

namespace k2r {
hipError_t launch_object_sha256(const TileArgs* tiles, const TileResult* results, uint32_t n, uint8_t* digests, hipStream_t stream);
}

extern "C" int dcdf_encoder_object_sha256(dcdf_encoder* e, uint8_t* digests, float* kernel_ms) {
    if (!e || !digests) return DCDF_ERR_BAD_ARG;
    {
        const int mrc = materialize(e);
        if (mrc != DCDF_OK) return mrc;
    }
    const size_t n = e->desc.size();
    // the statuses the kernel may trust: tiles rejected on the host never reached the device
    std::vector<TileResult> res(e->results);
    for (size_t i = 0; i < n; i++)
        if (e->pre_status[i] != DCDF_OK) {
            res[i].status = ST_BAD_ARG;
            res[i].len = 0;
        }
    DevBuf d_res, d_dig;
    K2R_HIP(d_res.alloc(n * sizeof(TileResult)));
    K2R_HIP(d_dig.alloc(n * 32));
    K2R_HIP(hipMemcpy(d_res.p, res.data(), n * sizeof(TileResult), hipMemcpyHostToDevice));
    K2R_HIP(hipEventRecord(e->ev0, e->stream));
    K2R_HIP(launch_object_sha256(e->d_args.as<TileArgs>(), d_res.as<TileResult>(), (uint32_t)n, d_dig.as<uint8_t>(), e->stream));
    K2R_HIP(hipEventRecord(e->ev1, e->stream));
    K2R_HIP(hipStreamSynchronize(e->stream));
    if (kernel_ms) K2R_HIP(hipEventElapsedTime(kernel_ms, e->ev0, e->ev1));
    K2R_HIP(hipMemcpy(digests, d_dig.p, n * 32, hipMemcpyDeviceToHost));
    return DCDF_OK;
}

// ---- host-side gather of the encoded buffers ("{offsets[], bytes[]} per GPU plus the per-(chunk,instant)
// (min,max) pairs") ------------------------------------------------------------------------------------
// The tiles' slots are packed on the device (16-byte aligned starts, 16-byte vector copies) and come back in ONE copy.
namespace k2r {
struct PackItem {
    const uint8_t* src;
    uint64_t len;      // bytes
    uint64_t dst_off;  // multiple of 16
};
__global__ void __launch_bounds__(256) k_pack(const PackItem* __restrict__ items, uint32_t n, uint8_t* __restrict__ dst) {
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const PackItem it = items[i];
        const uint4* s = (const uint4*)it.src;  // slots are 256-byte aligned and at least len rounded up to 256 long
        uint4* d = (uint4*)(dst + it.dst_off);
        const uint64_t nv = (it.len + 15) / 16;
        for (uint64_t v = threadIdx.x; v < nv; v += blockDim.x) d[v] = s[v];
    }
}
}  // namespace k2r

extern "C" int dcdf_encoder_gather_size(dcdf_encoder* e, uint64_t* packed_bytes, uint64_t* minmax_words) {
    if (!e || !packed_bytes) return DCDF_ERR_BAD_ARG;
    uint64_t tot = 0;
    for (size_t i = 0; i < e->desc.size(); i++)
        if (e->pre_status[i] == DCDF_OK && e->results[i].status == ST_OK) tot += (e->results[i].len + 15) & ~15ull;
    *packed_bytes = tot;
    if (minmax_words) *minmax_words = e->minmax_total;
    return DCDF_OK;
}

extern "C" int dcdf_encoder_gather(dcdf_encoder* e, uint8_t* dst, size_t cap, uint64_t* offsets, uint64_t* lens, int64_t* minmax) {
    if (!e || !dst || !offsets || !lens) return DCDF_ERR_BAD_ARG;
    {
        const int mrc = materialize(e);
        if (mrc != DCDF_OK) return mrc;
    }
    const size_t n = e->desc.size();
    std::vector<PackItem> items;
    uint64_t tot = 0;
    for (size_t i = 0; i < n; i++) {
        const bool ok = e->pre_status[i] == DCDF_OK && e->results[i].status == ST_OK;
        offsets[i] = tot;
        lens[i] = ok ? e->results[i].len : 0;
        if (!ok || lens[i] == 0) continue;
        items.push_back(PackItem{e->args[i].out, lens[i], tot});
        tot += (lens[i] + 15) & ~15ull;
    }
    if (tot > cap) return DCDF_ERR_CAPACITY;
    if (!items.empty()) {
        // on a stream of its own: the gather only reads what a finished run left behind, so it may overlap other readers of the
        // session (the superchunk assembly hashes the objects on the session's stream meanwhile)
        struct OwnStream {
            hipStream_t s = nullptr;
            ~OwnStream() {
                if (s) StreamPool::get().give(true, s);
            }
        } own;
        K2R_HIP(StreamPool::get().take(true, &own.s));
        DevBuf d_items, d_packed;
        struct Settle {  // (declared behind the buffers: the stream is idle before an early return hands them back to the pool)
            hipStream_t s;
            ~Settle() { (void)hipStreamSynchronize(s); }
        } settle{own.s};
        K2R_HIP(d_items.alloc(items.size() * sizeof(PackItem)));
        K2R_HIP(d_packed.alloc_pooled(tot));
        K2R_HIP(hipMemcpyAsync(d_items.p, items.data(), items.size() * sizeof(PackItem), hipMemcpyHostToDevice, own.s));
        const uint32_t grid = (uint32_t)std::min<size_t>(items.size(), 8192);
        hipLaunchKernelGGL(k_pack, dim3(grid), dim3(256), 0, own.s, d_items.as<PackItem>(), (uint32_t)items.size(),
                           d_packed.as<uint8_t>());
        K2R_HIP(hipGetLastError());
        K2R_HIP(hipMemcpyAsync(dst, d_packed.p, tot, hipMemcpyDeviceToHost, own.s));
        K2R_HIP(hipStreamSynchronize(own.s));
    }
    if (minmax && e->minmax_total) K2R_HIP(hipMemcpy(minmax, e->d_minmax.p, e->minmax_total * 8, hipMemcpyDeviceToHost));
    return DCDF_OK;
}

extern "C" uint64_t dcdf_encoder_total_bytes(dcdf_encoder* e) {
    uint64_t s = 0;
    if (!e) return 0;
    for (size_t i = 0; i < e->desc.size(); i++)
        if (e->pre_status[i] == DCDF_OK && e->results[i].status == ST_OK) s += e->results[i].len;
    return s;
}

extern "C" void dcdf_encoder_destroy(dcdf_encoder* e) { delete e; }

// ---- host-buffer convenience: Chunk::build for a batch ------------------------------------------------
extern "C" void dcdf_free_encoded(dcdf_encoded* out, size_t n) {
    if (!out) return;
    for (size_t i = 0; i < n; i++) {
        std::free(out[i].bytes);
        std::free(out[i].minmax);
    }
    std::free(out);
}

// Host-side transfer machinery of the host-buffer entry point: two pinned staging buffers filled by a few packing
// threads while the previous one is in flight on the copy engine (uploads), and drained the same way (downloads).
namespace {

constexpr size_t kPinBytes = 64u << 20;
constexpr int kDownSlots = 4;                            // downloads see the two buffers as four slots ...
constexpr size_t kSlotBytes = 2 * kPinBytes / kDownSlots;  // ... of 32 MB, each with its own event

// Host threads for packing, unpacking and hashing: one and a half per CPU this process may use (they wait on page faults and on
// the copy engine as much as they compute) -- the cgroup's CPU quota where there is one, else the affinity mask -- 32 at most.
// K2R_HOST_THREADS overrides.
int host_threads() {
    static const int n = [] {
        if (const char* e = std::getenv("K2R_HOST_THREADS")) {
            const int v = std::atoi(e);
            if (v >= 1) return std::min(v, 64);
        }
        int c = 0;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) c = CPU_COUNT(&set);
        if (c <= 0) c = (int)std::thread::hardware_concurrency();
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" or "max <period>"
            long long quota = 0, period = 0;
            if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
                c = std::min<int>(c, (int)std::max<long long>(1, (quota + period - 1) / period));
            std::fclose(f);
        }
        return std::max(1, std::min(c + c / 2, 32));
    }();
    return n;
}

struct PinRing {
    void* buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipEvent_t slot_ev[kDownSlots] = {nullptr, nullptr, nullptr, nullptr};
    bool busy[2] = {false, false};
    hipStream_t stream = nullptr;
    bool ok = false;
    std::mutex mu;  // one host-buffer batch at a time uses the ring
    bool init() {
        if (ok) return true;
        for (int i = 0; i < 2; i++) {
            // page-locked if the system allows it; otherwise ordinary memory (the copies then stage inside the runtime)
            if (!buf[i] && hipHostMalloc(&buf[i], kPinBytes, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                buf[i] = std::malloc(kPinBytes);
                if (!buf[i]) return false;
            }
            if (!ev[i] && hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return false;
        }
        for (int i = 0; i < kDownSlots; i++)
            if (!slot_ev[i] && hipEventCreateWithFlags(&slot_ev[i], hipEventDisableTiming) != hipSuccess) return false;
        if (!stream && hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return false;
        ok = true;
        return true;
    }
    bool wait(int b) {
        if (busy[b] && hipEventSynchronize(ev[b]) != hipSuccess) return false;
        busy[b] = false;
        return true;
    }
    uint8_t* slot(int s) const { return (uint8_t*)buf[s / (kDownSlots / 2)] + (size_t)(s % (kDownSlots / 2)) * kSlotBytes; }
};
PinRing& pin_ring() {
    static PinRing r;
    return r;
}

// runs f(0..n) on up to host_threads() threads
template <class F>
void parallel_for(size_t n, F&& f) {
    const size_t nt = std::min<size_t>((size_t)host_threads(), n);
    if (nt <= 1) {
        for (size_t i = 0; i < n; i++) f(i);
        return;
    }
    std::vector<std::thread> th;
    std::atomic<size_t> next{0};
    for (size_t t = 0; t < nt; t++)
        th.emplace_back([&] {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= n) break;
                f(i);
            }
        });
    for (auto& t : th) t.join();
}

// dense copy of a (possibly strided) host tile (mmbuffer.rs:517-522 tile slices)
void pack_tile(const dcdf_tile_desc& t, uint8_t* d) {
    const size_t es = elem_size(t.dtype);
    const uint8_t* src = (const uint8_t*)t.base;
    const bool contiguous = t.stride_c == 1 && t.stride_r == (int64_t)t.cols && t.stride_t == (int64_t)t.rows * t.cols;
    if (contiguous) {
        std::memcpy(d, src, (size_t)t.instants * t.rows * t.cols * es);
        return;
    }
    for (uint32_t a = 0; a < t.instants; a++)
        for (uint32_t r = 0; r < t.rows; r++) {
            const uint8_t* row = src + ((int64_t)a * t.stride_t + (int64_t)r * t.stride_r) * (int64_t)es;
            if (t.stride_c == 1) {
                std::memcpy(d, row, (size_t)t.cols * es);
                d += (size_t)t.cols * es;
            } else {
                for (uint32_t c = 0; c < t.cols; c++, d += es) std::memcpy(d, row + (int64_t)c * t.stride_c * (int64_t)es, es);
            }
        }
}

}  // namespace

static int build_group(const dcdf_tile_desc* tiles, size_t n, int k, int mem, dcdf_encoded* out) {
    // tiles[0..n) fit the staging budget; host tiles are packed contiguously and uploaded
    std::vector<dcdf_tile_desc> dev(tiles, tiles + n);
    DevBuf stage;
    PinRing& ring = pin_ring();
    std::unique_lock<std::mutex> ring_lock(ring.mu, std::defer_lock);
    if (mem == DCDF_MEM_HOST) {
        ring_lock.lock();
        if (!ring.init()) return DCDF_ERR_NOMEM;
        uint64_t total = 0;
        std::vector<uint64_t> off(n), bytes(n, 0);
        for (size_t i = 0; i < n; i++) {
            off[i] = total;
            const dcdf_tile_desc& t = tiles[i];
            if (!t.base || t.instants == 0 || t.rows == 0 || t.cols == 0) continue;
            if (t.dtype != DCDF_I32 && t.dtype != DCDF_I64 && t.dtype != DCDF_F32 && t.dtype != DCDF_F64) continue;
            bytes[i] = (uint64_t)t.instants * t.rows * t.cols * elem_size(t.dtype);
            total += (bytes[i] + 255) & ~255ull;
        }
        K2R_HIP(stage.alloc(total));
        // fills: runs of tiles whose staged range fits one pinned buffer (a tile larger than that goes alone, in pieces)
        size_t i = 0;
        int fill = 0;
        while (i < n) {
            if (bytes[i] == 0) {
                i++;
                continue;
            }
            const int b = fill & 1;
            if (!ring.wait(b)) return DCDF_ERR_NO_DEVICE;
            if (bytes[i] > kPinBytes) {  // oversize tile: dense temporary, then piecewise through the ring
                std::vector<uint8_t> tmp(bytes[i]);
                pack_tile(tiles[i], tmp.data());
                for (uint64_t o = 0; o < bytes[i]; o += kPinBytes) {
                    const int bb = fill & 1;
                    if (!ring.wait(bb)) return DCDF_ERR_NO_DEVICE;
                    const size_t len = (size_t)std::min<uint64_t>(kPinBytes, bytes[i] - o);
                    std::memcpy(ring.buf[bb], tmp.data() + o, len);
                    K2R_HIP(hipMemcpyAsync(stage.as<uint8_t>() + off[i] + o, ring.buf[bb], len, hipMemcpyHostToDevice, ring.stream));
                    K2R_HIP(hipEventRecord(ring.ev[bb], ring.stream));
                    ring.busy[bb] = true;
                    fill++;
                }
                i++;
                continue;
            }
            size_t j = i;
            const uint64_t base_off = off[i];
            while (j < n && (bytes[j] == 0 || off[j] + bytes[j] - base_off <= kPinBytes)) j++;
            uint8_t* pb = (uint8_t*)ring.buf[b];
            parallel_for(j - i, [&](size_t q) {
                const size_t ti = i + q;
                if (bytes[ti]) pack_tile(tiles[ti], pb + (off[ti] - base_off));
            });
            const uint64_t len = (j < n ? off[j] : total) - base_off;
            K2R_HIP(hipMemcpyAsync(stage.as<uint8_t>() + base_off, pb, std::min<uint64_t>(len, kPinBytes), hipMemcpyHostToDevice, ring.stream));
            K2R_HIP(hipEventRecord(ring.ev[b], ring.stream));
            ring.busy[b] = true;
            fill++;
            i = j;
        }
        if (!ring.wait(0) || !ring.wait(1)) return DCDF_ERR_NO_DEVICE;
        for (size_t q = 0; q < n; q++) {
            if (!bytes[q]) continue;
            const dcdf_tile_desc& t = tiles[q];
            dev[q].base = stage.as<uint8_t>() + off[q];
            dev[q].stride_c = 1;
            dev[q].stride_r = t.cols;
            dev[q].stride_t = (int64_t)t.rows * t.cols;
        }
    }
    dcdf_encoder* e = nullptr;
    int rc = dcdf_encoder_create(dev.data(), n, k, 0, &e);
    if (rc != DCDF_OK) return rc;
    std::unique_ptr<dcdf_encoder> guard(e);
    rc = dcdf_encoder_run(e, nullptr);
    if (rc != DCDF_OK) return rc;
    std::vector<int64_t> mm(e->minmax_total);
    if (e->minmax_total) K2R_HIP(hipMemcpy(mm.data(), e->d_minmax.p, e->minmax_total * 8, hipMemcpyDeviceToHost));
    std::vector<uint64_t> lens(n, 0);
    for (size_t i = 0; i < n; i++) {
        int32_t st;
        uint64_t len;
        uint32_t ns, nl;
        dcdf_encoder_result(e, i, &st, &len, &ns, &nl, nullptr);
        out[i].status = st;
        out[i].len = 0;
        out[i].bytes = nullptr;
        out[i].minmax = nullptr;
        out[i].snapshots = ns;
        out[i].logs = nl;
        if (st != DCDF_OK) continue;
        out[i].bytes = (uint8_t*)std::malloc(len ? len : 1);
        out[i].minmax = (int64_t*)std::malloc(16ull * tiles[i].instants);
        if (!out[i].bytes || !out[i].minmax) return DCDF_ERR_NOMEM;
        lens[i] = len;
        out[i].len = len;
        std::memcpy(out[i].minmax, mm.data() + e->minmax_off[i], 16ull * tiles[i].instants);
    }
    // encoded bytes: device slots -> pinned buffer (async, one copy per tile) -> the caller's buffers (threads)
    if (ring_lock.owns_lock()) ring_lock.unlock();
    return k2r::encoder_download(e, [&](size_t i, uint64_t) { return out[i].bytes; });
}

namespace k2r {
void host_parallel_for(size_t n, const std::function<void(size_t)>& f) { parallel_for(n, f); }
int host_thread_count() { return host_threads(); }

int encoder_download(dcdf_encoder* e, const std::function<uint8_t*(size_t, uint64_t)>& dst, const std::function<void(size_t)>& landed) {
    {
        const int mrc = materialize(e);
        if (mrc != DCDF_OK) return mrc;
    }
    const size_t n = e->desc.size();
    std::vector<uint64_t> lens(n, 0);
    for (size_t i = 0; i < n; i++)
        if (e->pre_status[i] == DCDF_OK && e->results[i].status == ST_OK) lens[i] = e->results[i].len;
    PinRing& ring = pin_ring();
    std::lock_guard<std::mutex> ring_lock(ring.mu);
    if (!ring.init()) return DCDF_ERR_NOMEM;
    // The copy engine fills one 32 MB slot of the pinned ring after the other (one asynchronous copy per object, one event per slot);
    // worker threads take the objects of a slot as soon as its event has fired -- destination from `dst`, memcpy, `landed` (the
    // caller's hash) -- and a slot is filled again when its last object has been taken out.  Copy, unpacking and hashing of
    // different slots overlap; nothing waits for a whole buffer.
    struct Task {
        size_t tile;
        int slot;
        uint64_t off;
    };
    std::mutex mu;
    std::condition_variable cv_task, cv_slot;
    std::deque<Task> tasks;
    int outstanding[kDownSlots] = {0, 0, 0, 0};
    bool closing = false;
    std::atomic<bool> nomem{false}, hipfail{false};
    auto worker = [&] {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_task.wait(lk, [&] { return !tasks.empty() || closing; });
                if (tasks.empty()) return;
                t = tasks.front();
                tasks.pop_front();
            }
            if (hipEventSynchronize(ring.slot_ev[t.slot]) != hipSuccess) hipfail = true;
            else if (uint8_t* d = dst(t.tile, lens[t.tile])) {
                std::memcpy(d, ring.slot(t.slot) + t.off, lens[t.tile]);
                if (landed) landed(t.tile);
            } else nomem = true;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--outstanding[t.slot] == 0) cv_slot.notify_all();
            }
        }
    };
    size_t n_big = 0;
    for (size_t i = 0; i < n; i++) n_big += lens[i] != 0 && lens[i] <= kSlotBytes;
    std::vector<std::thread> pool;
    for (size_t t = 0, nt = std::min<size_t>((size_t)host_threads(), n_big); t < nt; t++) pool.emplace_back(worker);
    // Every exit leaves the ring as the next caller expects it: workers gone, no copy still landing in a pinned buffer.
    auto finish = [&](int rc) -> int {
        {
            std::lock_guard<std::mutex> lk(mu);
            closing = true;
        }
        cv_task.notify_all();
        for (auto& t : pool) t.join();
        if (rc != DCDF_OK || hipfail) (void)hipStreamSynchronize(ring.stream);
        if (rc != DCDF_OK) return rc;
        if (hipfail) return DCDF_ERR_NO_DEVICE;
        return nomem ? DCDF_ERR_NOMEM : DCDF_OK;
    };
#define K2R_HIP_Q(call)                                                 \
    do {                                                                \
        hipError_t _e = (call);                                         \
        if (_e != hipSuccess) return finish(k2r::map_hip_error(_e));    \
    } while (0)
    size_t i = 0;
    int fill = 0;
    while (i < n) {
        if (lens[i] == 0) {
            i++;
            continue;
        }
        if (lens[i] > kSlotBytes) {  // oversize result: plain copy
            uint8_t* d = dst(i, lens[i]);
            if (!d) return finish(DCDF_ERR_NOMEM);
            K2R_HIP_Q(hipMemcpy(d, e->args[i].out, lens[i], hipMemcpyDeviceToHost));
            if (landed) landed(i);
            i++;
            continue;
        }
        const int s = fill % kDownSlots;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_slot.wait(lk, [&] { return outstanding[s] == 0; });
        }
        std::vector<Task> batch;
        uint64_t used = 0;
        while (i < n && (lens[i] == 0 || (lens[i] <= kSlotBytes && used + lens[i] <= kSlotBytes))) {
            if (lens[i]) {
                K2R_HIP_Q(hipMemcpyAsync(ring.slot(s) + used, e->args[i].out, lens[i], hipMemcpyDeviceToHost, ring.stream));
                batch.push_back(Task{i, s, used});
                used += (lens[i] + 63) & ~63ull;
            }
            i++;
        }
        K2R_HIP_Q(hipEventRecord(ring.slot_ev[s], ring.stream));
        {
            std::lock_guard<std::mutex> lk(mu);
            outstanding[s] = (int)batch.size();
            for (const Task& t : batch) tasks.push_back(t);
        }
        cv_task.notify_all();
        fill++;
    }
#undef K2R_HIP_Q
    return finish(DCDF_OK);
}
}  // namespace k2r

extern "C" int dcdf_chunk_build_batch(const dcdf_tile_desc* tiles, size_t n, int k, int mem, dcdf_encoded** out) {
    if (!tiles || !out || n == 0 || (mem != DCDF_MEM_HOST && mem != DCDF_MEM_DEVICE)) return DCDF_ERR_BAD_ARG;
    if (!Runtime::get().ok) return DCDF_ERR_NO_DEVICE;
    dcdf_encoded* res = (dcdf_encoded*)std::calloc(n, sizeof(dcdf_encoded));
    if (!res) return DCDF_ERR_NOMEM;
    const uint64_t budget = 4ull << 30;  // raw bytes staged per group
    size_t i = 0;
    while (i < n) {
        uint64_t acc = 0;
        size_t j = i;
        while (j < n) {
            const uint64_t b = (uint64_t)tiles[j].instants * tiles[j].rows * tiles[j].cols * elem_size(tiles[j].dtype);
            if (j > i && acc + b > budget) break;
            acc += b;
            j++;
        }
        const int rc = build_group(tiles + i, j - i, k, mem, res + i);
        if (rc != DCDF_OK) {
            dcdf_free_encoded(res, n);
            return rc;
        }
        i = j;
    }
    *out = res;
    return DCDF_OK;
}

extern "C" int dcdf_chunk_build(const dcdf_tile_desc* tile, int k, int mem, dcdf_encoded** out) {
    return dcdf_chunk_build_batch(tile, 1, k, mem, out);
}

extern "C" const char* dcdf_strerror(int code) {
    switch (code) {
        case DCDF_OK: return "ok";
        case DCDF_ERR_BAD_ARG: return "bad argument";
        case DCDF_ERR_NONFINITE: return "cannot convert a non-finite float to fixed point (fixed.rs:39-41)";
        case DCDF_ERR_PRECISION: return "fixed-point conversion loses precision and round is false (fixed.rs:47-59)";
        case DCDF_ERR_OVERFLOW: return "fixed-point value overflows i64 (fixed.rs:65-70)";
        case DCDF_ERR_BOUNDS: return "query out of bounds";
        case DCDF_ERR_TOO_MANY_LOGS: return "too many logs in one block (block.rs:27-32)";
        case DCDF_ERR_FORMAT: return "malformed encoded chunk";
        case DCDF_ERR_UNSUPPORTED: return "unsupported: sidelen above 1024 (universal kernel's limit; the fused kernel covers k = 2, sidelen 16..256)";
        case DCDF_ERR_NO_DEVICE: return "no usable gfx950 device / HIP failure";
        case DCDF_ERR_NOMEM: return "out of memory";
        case DCDF_ERR_CAPACITY: return "result buffer too small";
        case DCDF_ERR_INTERNAL: return "internal consistency guard tripped in a kernel (bug; see stderr)";
        case DCDF_ERR_HIP: return "HIP runtime failure (see dcdf_last_hip_error)";
        case DCDF_ERR_HIP_INVALID: return "HIP rejected an argument (bad device pointer or value)";
    }
    return "unknown error";
}
extern "C" const char* dcdf_device_name(void) {
    Runtime& rt = Runtime::get();
    return rt.ok ? rt.name.c_str() : nullptr;
}
extern "C" int dcdf_abi_version(void) { return 3; }
extern "C" int dcdf_device_pool_trim(uint64_t* freed_bytes) {
    if (!Runtime::get().ok) return DCDF_ERR_NO_DEVICE;
    (void)hipDeviceSynchronize();
    const size_t f = k2r::DevPool::get().drain();
    if (freed_bytes) *freed_bytes = (uint64_t)f;
    return DCDF_OK;
}
extern "C" int dcdf_last_hip_error(void) { return k2r::last_hip_error(); }
