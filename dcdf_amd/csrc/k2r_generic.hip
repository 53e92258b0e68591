// k2r_generic.hip -- the universal encoder: any k >= 2, any sidelen = k^H (H = 0 .. ), full int64 values (all 8 Dac
// planes), padded or not.  Everything the fused kernel (k2r_encode.h) declines -- k != 2, sidelen < 8 or > 256, a stored
// value beyond 2^30 -- is encoded here, still on the GPU, bit-identical to Chunk::build + Chunk::write_to
// (chunk.rs:42-96,235-243).  Throughput is not the point of this path (one workgroup per tile, level arrays in HBM scratch,
// workgroup scans); the fused kernel stays the dispatch target whenever its contract holds.
//
// Formulation (data-parallel, level by level):
//   * nodes of depth d are stored in "k^2-ary Morton" order: children of node m are m*k^2 + (i*k + j), the reference's child
//     order (snapshot.rs:468-474).  BFS order (snapshot.rs:126-146, log.rs:132-154) restricted to one depth is then index
//     order restricted to the visited nodes, so stream positions are prefix sums of the visited / internal flags.
//   * bottom-up: (min, max) of t and of the block's snapshot instant s, `diff`, `equal` (snapshot.rs:439-500,
//     log.rs:725-817); a node is None exactly when its origin cell lies outside the tile (padding is at the bottom/right).
//   * top-down per depth: visited = parent internal; Snapshot: internal = min != max; Log: internal = min_t != max_t and not
//     equal.  Values go to the Lmax / Lmin streams, flags to T / eqB (one byte per bit in scratch).
//   * sizes from the streams (dac.rs:68-74, bitmap.rs:169-171), chunk.rs:62, then the winner is serialized: BitMap words +
//     rank index (bitmap.rs:66-112), Dac planes by compaction (dac.rs:101-131).
#include <hip/hip_runtime.h>

#include "k2r_encode.h"
#include "k2r_launch.h"

namespace k2r {

constexpr int GEN_NT = 1024;
constexpr int GEN_MAXD = 40;

struct GenGeom {
    uint32_t k, k2, H;     // arity, k*k, depth of the cells
    uint64_t S;            // sidelen
    uint64_t lo[GEN_MAXD]; // first node of depth d
    uint64_t NN;           // all nodes
};

__host__ __device__ inline GenGeom gen_geom(uint32_t k, uint32_t H) {
    GenGeom g;
    g.k = k; g.k2 = k * k; g.H = H;
    uint64_t n = 1, off = 0, s = 1;
    for (uint32_t d = 0; d <= H; d++) {
        g.lo[d] = off;
        off += n;
        n *= g.k2;
        if (d < H) s *= k;
    }
    g.S = s;
    g.NN = off;
    return g;
}
// bytes of scratch one workgroup needs
__host__ __device__ inline uint64_t gen_scratch_bytes(const GenGeom& g) {
    return g.NN * (9 * 8 + 5) + 4096;
}

struct GenScratch {
    int64_t *tmin, *tmax, *smin, *smax, *diff, *VS, *MS, *VL, *ML;
    uint8_t *flg, *TS, *TL, *EL, *tmp;
    __device__ void bind(uint8_t* p, uint64_t NN) {
        int64_t* q = (int64_t*)p;
        tmin = q; tmax = q + NN; smin = q + 2 * NN; smax = q + 3 * NN; diff = q + 4 * NN;
        VS = q + 5 * NN; MS = q + 6 * NN; VL = q + 7 * NN; ML = q + 8 * NN;
        uint8_t* b = (uint8_t*)(q + 9 * NN);
        flg = b; TS = b + NN; TL = b + 2 * NN; EL = b + 3 * NN; tmp = b + 4 * NN;
    }
};
enum : uint8_t { GF_VALID = 1, GF_EQUAL = 2, GF_VIS_S = 4, GF_INT_S = 8, GF_VIS_L = 16, GF_INT_L = 32 };

struct GenShared {
    uint32_t sw[GEN_NT / 64 + 1];
    uint64_t cnt[8];
    int32_t err;
    uint32_t work;
    uint64_t u64a, u64b;
};

// workgroup exclusive scan of one 32-bit value per thread; *total = sum over the workgroup
__device__ inline uint32_t block_scan_excl(GenShared& sh, uint32_t v, uint32_t* total) {
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) sh.sw[wave] = x;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int w = 0; w < GEN_NT / 64; w++) {
            const uint32_t t = sh.sw[w];
            sh.sw[w] = run;
            run += t;
        }
        sh.sw[GEN_NT / 64] = run;
    }
    __syncthreads();
    const uint32_t r = x - v + sh.sw[wave];
    *total = sh.sw[GEN_NT / 64];
    __syncthreads();
    return r;
}

// origin cell of node m of depth d
__device__ inline void gen_origin(const GenGeom& g, uint32_t d, uint64_t m, uint64_t& row, uint64_t& col) {
    uint64_t r = 0, c = 0, mul = 1;
    for (uint32_t e = 0; e < d; e++) {
        const uint32_t dig = (uint32_t)(m % g.k2);
        m /= g.k2;
        r += (uint64_t)(dig / g.k) * mul;
        c += (uint64_t)(dig % g.k) * mul;
        mul *= g.k;
    }
    // r, c are in units of the node's side
    uint64_t side = 1;
    for (uint32_t e = d; e < g.H; e++) side *= g.k;
    row = r * side;
    col = c * side;
}

__device__ inline uint32_t zz_bytes(int64_t v) {  // planes a value occupies (dac.rs:101-120): at least one
    uint64_t z = (uint64_t)(v >> 63) ^ ((uint64_t)v << 1);
    uint32_t n = 1;
    while (z >>= 8) n++;
    return n;
}

// serialized size of the Dac of values[0..n) (dac.rs:68-74); cnt = scratch in LDS
__device__ uint64_t gen_dac_size(GenShared& sh, const int64_t* values, uint64_t n) {
    const int tid = (int)threadIdx.x;
    if (tid < 8) sh.cnt[tid] = 0;
    __syncthreads();
    uint32_t local[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint64_t i = tid; i < n; i += GEN_NT) {
        const uint32_t b = zz_bytes(values[i]);
        for (uint32_t j = 0; j < b; j++) local[j]++;
    }
    for (int j = 0; j < 8; j++)
        if (local[j]) atomicAdd((unsigned long long*)&sh.cnt[j], (unsigned long long)local[j]);
    __syncthreads();
    uint64_t size = 1;
    for (int j = 0; j < 8; j++) {
        const uint64_t nj = sh.cnt[j];
        if (nj == 0) break;
        size += 8 + 4 * (nj / 128) + 4 * ((nj + 31) / 32) + nj;
    }
    __syncthreads();
    return size;
}
__device__ inline uint64_t gen_bitmap_size(uint64_t nbits) { return 8 + 4 * (nbits / 128) + 4 * ((nbits + 31) / 32); }

// BitMap::write_to of the bits b[0..len) (one byte per bit) at dst; returns the bytes written (bitmap.rs:66-112,128-138)
__device__ uint64_t gen_write_bitmap(GenShared& sh, const uint8_t* b, uint64_t len, uint8_t* dst) {
    const int tid = (int)threadIdx.x;
    const uint64_t words = (len + 31) / 32, nidx = len / 128;
    if (tid == 0) {
        store_be32(dst, (uint32_t)len);
        store_be32(dst + 4, 4u);
    }
    uint8_t* const idx = dst + 8;
    uint8_t* const wd = dst + 8 + 4 * nidx;
    uint32_t carry = 0;
    for (uint64_t base = 0; base < words; base += GEN_NT) {
        const uint64_t w = base + tid;
        uint32_t x = 0;
        if (w < words) {
            for (uint32_t i = 0; i < 32; i++) {
                const uint64_t p = w * 32 + i;
                if (p < len && b[p]) x |= 0x80000000u >> i;
            }
            store_be32(wd + 4 * w, x);
        }
        uint32_t total;
        const uint32_t ex = block_scan_excl(sh, popc32(x), &total);
        if (w < words && (w & 3) == 3 && (w >> 2) < nidx) store_be32(idx + 4 * (w >> 2), carry + ex + popc32(x));
        carry += total;
    }
    return 8 + 4 * nidx + 4 * words;
}

// Dac::write_to of values[0..n) at dst (dac.rs:37-44,101-131); tmp = n bytes of scratch; returns the bytes written
__device__ uint64_t gen_write_dac(GenShared& sh, const int64_t* values, uint64_t n, uint8_t* dst, uint8_t* tmp) {
    const int tid = (int)threadIdx.x;
    uint64_t off = 1;
    uint32_t nlev = 0;
    // plane j: the values with more than j bytes, in order
    for (uint32_t j = 0; j < 8; j++) {
        // how many, and the continuation bits of this plane (compacted into tmp)
        uint64_t carry = 0;
        for (uint64_t base = 0; base < n; base += GEN_NT) {
            const uint64_t i = base + tid;
            uint32_t nb = 0;
            if (i < n) nb = zz_bytes(values[i]);
            const uint32_t f = nb > j ? 1u : 0u;
            uint32_t total;
            const uint32_t ex = block_scan_excl(sh, f, &total);
            if (f) tmp[carry + ex] = nb > j + 1 ? 1 : 0;
            carry += total;
        }
        const uint64_t nj = carry;
        if (nj == 0) break;  // take_while, dac.rs:124-128
        nlev = j + 1;
        __syncthreads();
        const uint64_t bmsz = gen_write_bitmap(sh, tmp, nj, dst + off);
        off += bmsz;
        __syncthreads();
        carry = 0;
        for (uint64_t base = 0; base < n; base += GEN_NT) {
            const uint64_t i = base + tid;
            uint32_t nb = 0;
            uint64_t z = 0;
            if (i < n) {
                const int64_t v = values[i];
                z = (uint64_t)(v >> 63) ^ ((uint64_t)v << 1);
                nb = zz_bytes(v);
            }
            const uint32_t f = nb > j ? 1u : 0u;
            uint32_t total;
            const uint32_t ex = block_scan_excl(sh, f, &total);
            if (f) dst[off + carry + ex] = (uint8_t)(z >> (8 * j));
            carry += total;
        }
        off += nj;
        __syncthreads();
    }
    if (tid == 0) dst[0] = (uint8_t)nlev;
    return off;
}

// One tile.  `out == nullptr` or too small: sizes are still computed and res->len holds the bytes needed.
__device__ void gen_encode_tile(GenShared& sh, const TileArgs& ta, TileResult* res, const GenGeom& g, GenScratch& sc) {
    const int tid = (int)threadIdx.x;
    const uint32_t H = g.H;
    uint8_t* const out = ta.out;
    const uint64_t cap = ta.out_cap;
    uint64_t off = 6;
    uint32_t n_blocks = 0, blk_count = 0, n_snap = 0, n_log = 0;
    uint64_t blk_hdr = 6;
    bool fits = cap >= 7;
    if (tid == 0) sh.err = 0;
    __syncthreads();

    for (uint32_t inst = 0; inst < ta.instants; inst++) {
        const bool have_s = inst > 0;
        // ---- cells (snapshot.rs:452-457, log.rs:741-759) ----
        for (uint64_t m = tid; m < g.NN - g.lo[H]; m += GEN_NT) {
            uint64_t r, c;
            gen_origin(g, H, m, r, c);
            const bool valid = r < ta.rows && c < ta.cols;
            int32_t err = 0;
            int64_t t = 0;
            if (valid) t = load_stored(ta, (int64_t)inst * ta.st + (int64_t)r * ta.sr + (int64_t)c * ta.sc, err);
            if (err) atomicMin(&sh.err, err);
            const uint64_t a = g.lo[H] + m;
            sc.tmin[a] = t;
            sc.tmax[a] = t;
            const int64_t s = have_s ? sc.smin[a] : 0;
            sc.diff[a] = valid ? t - s : 0;
            sc.flg[a] = (uint8_t)((valid ? GF_VALID : 0) | GF_EQUAL);
        }
        __syncthreads();
        if (sh.err != 0) break;
        // ---- bottom-up (snapshot.rs:476-497, log.rs:776-806) ----
        for (int d = (int)H - 1; d >= 0; d--) {
            const uint64_t nd = g.lo[d + 1] - g.lo[d];
            for (uint64_t m = tid; m < nd; m += GEN_NT) {
                const uint64_t a = g.lo[d] + m, c0 = g.lo[d + 1] + m * g.k2;
                const bool valid = sc.flg[c0] & GF_VALID;  // the first child holds the node's origin
                int64_t mn = sc.tmin[c0], mx = sc.tmax[c0];
                const int64_t df = sc.diff[c0];
                bool eq = sc.flg[c0] & GF_EQUAL;
                for (uint32_t q = 1; q < g.k2; q++) {
                    const uint64_t c = c0 + q;
                    const uint8_t f = sc.flg[c];
                    if (f & GF_VALID) {
                        if (sc.tmin[c] < mn) mn = sc.tmin[c];
                        if (sc.tmax[c] > mx) mx = sc.tmax[c];
                    }
                    eq = eq && (f & GF_EQUAL) && sc.diff[c] == df;
                }
                sc.tmin[a] = valid ? mn : 0;
                sc.tmax[a] = valid ? mx : 0;
                sc.diff[a] = df;  // (the snapshot's own pyramid, smin / smax, is kept as it was built)
                sc.flg[a] = (uint8_t)((valid ? GF_VALID : 0) | (eq ? GF_EQUAL : 0));
            }
            __syncthreads();
        }
        // ---- top-down: both candidates' streams (snapshot.rs:122-146, log.rs:128-154) ----
        uint64_t vS = 0, iS = 0, vL = 0, iL = 0, zL = 0;  // stream lengths so far: Lmax, Lmin (, eqB) of snapshot / log
        uint64_t ltS = 0, ltL = 0;                           // T lengths (nodes with children)
        for (uint32_t d = 0; d <= H; d++) {
            const uint64_t nd = (d == H ? g.NN : g.lo[d + 1]) - g.lo[d];
            const bool has_children = d < H;
            uint64_t cvS = 0, ciS = 0, cvL = 0, ciL = 0, czL = 0;
            for (uint64_t base = 0; base < nd; base += GEN_NT) {
                const uint64_t m = base + tid;
                const bool in = m < nd;
                const uint64_t a = g.lo[d] + (in ? m : 0);
                uint8_t f = in ? sc.flg[a] : 0;
                const uint8_t pf = (in && d > 0) ? sc.flg[g.lo[d - 1] + m / g.k2] : 0;
                const bool visS = in && (d == 0 || (pf & GF_INT_S)), visL = in && have_s && (d == 0 || (pf & GF_INT_L));
                const int64_t mn = in ? sc.tmin[a] : 0, mx = in ? sc.tmax[a] : 0;
                const int64_t smn = (in && have_s) ? sc.smin[a] : 0, smx = (in && have_s) ? sc.smax[a] : 0;
                const bool valid = f & GF_VALID;
                const bool intS = visS && has_children && mn != mx;                                   // snapshot.rs:133
                const bool intL = visL && has_children && valid && mn != mx && !(f & GF_EQUAL);        // log.rs:137-152
                const bool zl = visL && has_children && !intL;
                uint32_t t1, t2, t3;
                const uint32_t e1 = block_scan_excl(sh, (visS ? 1u : 0u) | (intS ? 1u << 16 : 0u), &t1);
                const uint32_t e2 = block_scan_excl(sh, (visL ? 1u : 0u) | (intL ? 1u << 16 : 0u), &t2);
                const uint32_t e3 = block_scan_excl(sh, zl ? 1u : 0u, &t3);
                if (visS) {
                    const uint64_t p = vS + cvS + (e1 & 0xffffu);
                    int64_t pmx = 0, pmn = 0;
                    if (d > 0) {
                        const uint64_t pa = g.lo[d - 1] + m / g.k2;
                        pmx = sc.tmax[pa];
                        pmn = sc.tmin[pa];
                    }
                    sc.VS[p] = d == 0 ? mx : pmx - mx;  // snapshot.rs:123,139
                    if (has_children) sc.TS[p] = intS ? 1 : 0;
                    if (intS) sc.MS[iS + ciS + (e1 >> 16)] = d == 0 ? mn : mn - pmn;  // snapshot.rs:123,140
                }
                if (visL) {
                    const uint64_t p = vL + cvL + (e2 & 0xffffu);
                    sc.VL[p] = mx - smx;  // log.rs:133 (0 - 0 for nodes outside the tile)
                    if (has_children) sc.TL[p] = intL ? 1 : 0;
                    if (intL) sc.ML[iL + ciL + (e2 >> 16)] = mn - smn;  // log.rs:148
                    if (zl) sc.EL[zL + czL + e3] = (valid && mn != mx) ? 1 : 0;  // "equal" rather than uniform (log.rs:137-144)
                }
                if (in) sc.flg[a] = (uint8_t)((f & (GF_VALID | GF_EQUAL)) | (visS ? GF_VIS_S : 0) | (intS ? GF_INT_S : 0) |
                                              (visL ? GF_VIS_L : 0) | (intL ? GF_INT_L : 0));
                cvS += t1 & 0xffffu; ciS += t1 >> 16; cvL += t2 & 0xffffu; ciL += t2 >> 16; czL += t3;
            }
            vS += cvS; iS += ciS; vL += cvL; iL += ciL; zL += czL;
            if (has_children) {
                ltS += cvS;
                ltL += cvL;
            }
            __syncthreads();
        }
        // ---- sizes and the heuristic (snapshot.rs:87-92, log.rs:95-97, chunk.rs:62) ----
        const uint64_t snap_size = 13 + gen_bitmap_size(ltS) + gen_dac_size(sh, sc.VS, vS) + gen_dac_size(sh, sc.MS, iS);
        uint64_t log_size = 0;
        if (have_s) log_size = 13 + gen_bitmap_size(ltL) + gen_bitmap_size(zL) + gen_dac_size(sh, sc.VL, vL) + gen_dac_size(sh, sc.ML, iL);
        const bool as_snapshot = !have_s || blk_count - 1 == 254 || snap_size <= log_size;
        const uint64_t isize = as_snapshot ? snap_size : log_size;
        if (as_snapshot) {
            if (have_s) {
                if (fits && tid == 0) out[blk_hdr] = (uint8_t)blk_count;  // block.rs:89
                n_blocks++;
            }
            blk_hdr = off;
            off += 1;
            blk_count = 0;
            n_snap++;
            // this instant is the block's snapshot from now on
            for (uint64_t a = tid; a < g.NN; a += GEN_NT) {
                sc.smin[a] = sc.tmin[a];
                sc.smax[a] = sc.tmax[a];
            }
        } else {
            n_log++;
        }
        if (off + isize > cap) fits = false;
        if (ta.minmax && tid == 0) {
            ta.minmax[2 * inst] = sc.tmin[0];
            ta.minmax[2 * inst + 1] = sc.tmax[0];
        }
        __syncthreads();
        if (fits) {
            uint8_t* io = out + off;
            if (tid == 0) {
                io[0] = (uint8_t)g.k;  // snapshot.rs:49, log.rs:54
                store_be32(io + 1, ta.rows);
                store_be32(io + 5, ta.cols);
                store_be32(io + 9, (uint32_t)g.S);
            }
            uint64_t o = 13;
            if (as_snapshot) {
                o += gen_write_bitmap(sh, sc.TS, ltS, io + o);
                __syncthreads();
                o += gen_write_dac(sh, sc.VS, vS, io + o, sc.tmp);
                o += gen_write_dac(sh, sc.MS, iS, io + o, sc.tmp);
            } else {
                o += gen_write_bitmap(sh, sc.TL, ltL, io + o);
                __syncthreads();
                o += gen_write_bitmap(sh, sc.EL, zL, io + o);
                __syncthreads();
                o += gen_write_dac(sh, sc.VL, vL, io + o, sc.tmp);
                o += gen_write_dac(sh, sc.ML, iL, io + o, sc.tmp);
            }
            if (o != isize && tid == 0) atomicMin(&sh.err, (int32_t)ST_INTERNAL);
        }
        off += isize;
        blk_count++;
        __syncthreads();
    }
    __syncthreads();
    if (tid == 0) {
        const int32_t e = sh.err;
        int32_t status = e != 0 ? e : (fits ? (int32_t)ST_OK : (int32_t)ST_OUT_CAPACITY);
        if (status == ST_OK) {
            out[0] = (uint8_t)ta.dtype;  // chunk.rs:236-238
            out[1] = (uint8_t)ta.fbits;
            out[blk_hdr] = (uint8_t)blk_count;
            store_be32(out + 2, n_blocks + 1);
        }
        res->status = status;
        res->snapshots = n_snap;
        res->logs = n_log;
        res->stash_logs = 0;
        res->len = (status == ST_OK || status == ST_OUT_CAPACITY) ? off : 0;  // on OUT_CAPACITY: the bytes a retry needs
        for (int i = 0; i < 6; i++) res->dbg[i] = 0;
    }
    __syncthreads();
}

__global__ void __launch_bounds__(GEN_NT)
k_encode_generic(const TileArgs* __restrict__ tiles, TileResult* __restrict__ results, const uint32_t* __restrict__ order, uint32_t n,
                 uint32_t* __restrict__ queue, uint8_t* __restrict__ scratch, uint64_t scratch_per_wg, uint32_t k, uint32_t H) {
    __shared__ GenShared sh;
    const GenGeom g = gen_geom(k, H);
    GenScratch sc;
    sc.bind(scratch + (uint64_t)blockIdx.x * scratch_per_wg, g.NN);
    for (;;) {
        if (threadIdx.x == 0) sh.work = atomicAdd(queue, 1u);
        __syncthreads();
        const uint32_t w = sh.work;
        __syncthreads();
        if (w >= n) break;
        const uint32_t ti = order[w];
        gen_encode_tile(sh, tiles[ti], &results[ti], g, sc);
    }
}

// Dac::from(values).write_to (dac.rs:37-44,101-131) for a caller-supplied array: one workgroup; out_len[0] = bytes written.
// Used by the superchunk assembly for its instant-major min / max Dacs (superchunk.rs:190-198,246-247).
__global__ void __launch_bounds__(GEN_NT) k_dac_pack(const int64_t* __restrict__ values, uint64_t n, uint8_t* __restrict__ out,
                                                     uint8_t* __restrict__ tmp, uint64_t* __restrict__ out_len) {
    __shared__ GenShared sh;
    const uint64_t len = gen_write_dac(sh, values, n, out, tmp);
    if (threadIdx.x == 0) out_len[0] = len;
}
hipError_t launch_dac_pack(const int64_t* values, uint64_t n, uint8_t* out, uint8_t* tmp, uint64_t* out_len, hipStream_t stream) {
    hipLaunchKernelGGL(k_dac_pack, dim3(1), dim3(GEN_NT), 0, stream, values, n, out, tmp, out_len);
    return hipGetLastError();
}

hipError_t launch_encode_generic(const EncodeLaunch& L, uint8_t* scratch, uint64_t scratch_per_wg, uint32_t k, uint32_t H, hipStream_t stream) {
    hipLaunchKernelGGL(k_encode_generic, dim3(L.grid), dim3(GEN_NT), 0, stream, L.tiles, L.results, L.order, L.n, L.queue, scratch,
                       scratch_per_wg, k, H);
    return hipGetLastError();
}
uint64_t generic_scratch_bytes(uint32_t k, uint32_t H) { return gen_scratch_bytes(gen_geom(k, H)); }

}  // namespace k2r
