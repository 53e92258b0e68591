// k2r_suggest.hip -- Fraction suggestion for a float tile on the GPU (reference fixed.rs:96-159, called per
// buffer from mmbuffer.rs:596-613,658-675).  Two streaming reductions over the (strided) tile:
//   pass 1: the largest non-NaN value                       -> whole_bits, max_fraction_bits (fixed.rs:107-131)
//   pass 2: "does any value still have a fraction after the widest shift?" (=> Fraction::Round) and the largest
//           max_fraction_bits - trailing_zeros(shifted as i64) otherwise (=> Fraction::Precise), fixed.rs:138-158
// The reference returns Round from the first offending element; the answer does not depend on which one is first.
// Both passes are HBM streaming reads (4 or 8 bytes per cell each); partial results per workgroup are combined on
// the host (at most a few thousand entries).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "k2r_runtime.h"

namespace k2r {

struct SfView {
    const void* base;
    int64_t st, sr, sc;
    uint32_t instants, rows, cols;
};
struct SfPartial {
    double max;       // pass 1: max of the non-NaN values seen (valid iff seen)
    uint32_t seen;    // pass 1: saw a non-NaN value
    uint32_t bits;    // pass 2: max these_bits
    uint32_t frac;    // pass 2: some value keeps a fraction
    uint32_t _pad;
};

template <class F>
__device__ __forceinline__ F sf_load(const SfView& v, uint64_t i) {
    const uint64_t plane = (uint64_t)v.rows * v.cols;
    const uint64_t t = i / plane, rem = i - t * plane;
    const uint64_t r = rem / v.cols, c = rem - r * v.cols;
    return ((const F*)v.base)[(int64_t)t * v.st + (int64_t)r * v.sr + (int64_t)c * v.sc];
}

// Workgroup b looks at view b / bpt (bpt workgroups per view stride over its cells); maxbits[view] = pass 2's widest shift.
// One launch per pass covers every float tile of a superchunk level (dcdf_suggest_fraction: a batch of one).
template <class F, int PASS>
__global__ void __launch_bounds__(256) k_suggest(const SfView* __restrict__ views, const uint32_t* __restrict__ maxbits, uint32_t bpt,
                                                 SfPartial* out) {
    __shared__ double s_max[4];
    __shared__ uint32_t s_a[4], s_b[4];
    const SfView v = views[blockIdx.x / bpt];
    const uint64_t n = (uint64_t)v.instants * v.rows * v.cols;
    const uint32_t max_fraction_bits = PASS == 2 ? maxbits[blockIdx.x / bpt] : 0u;
    double mx = 0.0;
    uint32_t seen = 0, bits = 0, frac = 0;
    const double scale = (double)((int64_t)1 << max_fraction_bits);
    for (uint64_t i = (uint64_t)(blockIdx.x % bpt) * 256 + threadIdx.x; i < n; i += (uint64_t)bpt * 256) {
        const double x = (double)sf_load<F>(v, i);
        if (x != x) continue;  // NaN: skipped by both loops (fixed.rs:108-123,140-142)
        if (PASS == 1) {
            mx = seen ? (x > mx ? x : mx) : x;
            seen = 1;
        } else {
            const double shifted = x * scale;
            if (shifted - trunc(shifted) != 0.0) {  // fract() != 0; NaN for infinities, which also compares unequal
                frac = 1;
            } else {
                long long si;  // `shifted as i64` saturates in Rust
                if (shifted >= 9223372036854775808.0) si = 0x7fffffffffffffffLL;
                else if (shifted <= -9223372036854775808.0) si = (long long)0x8000000000000000ULL;
                else si = (long long)shifted;
                const uint32_t tz = si == 0 ? 64u : (uint32_t)__builtin_ctzll((unsigned long long)si);
                const uint32_t these = max_fraction_bits > tz ? max_fraction_bits - tz : 0u;  // saturating_sub
                bits = these > bits ? these : bits;
            }
        }
    }
    // wave reduction (64 lanes), then the four waves through LDS
    for (int d = 32; d >= 1; d >>= 1) {
        const double omx = __shfl_xor(mx, d, 64);
        const uint32_t oseen = __shfl_xor(seen, d, 64), obits = __shfl_xor(bits, d, 64), ofrac = __shfl_xor(frac, d, 64);
        if (oseen) mx = seen ? (omx > mx ? omx : mx) : omx;
        seen |= oseen;
        bits = obits > bits ? obits : bits;
        frac |= ofrac;
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_max[w] = mx;
        s_a[w] = PASS == 1 ? seen : bits;
        s_b[w] = frac;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        SfPartial p{0.0, 0, 0, 0, 0};
        for (int k = 0; k < 4; k++) {
            if (PASS == 1) {
                if (s_a[k]) p.max = p.seen ? (s_max[k] > p.max ? s_max[k] : p.max) : s_max[k];
                p.seen |= s_a[k];
            } else {
                p.bits = s_a[k] > p.bits ? s_a[k] : p.bits;
                p.frac |= s_b[k];
            }
        }
        out[blockIdx.x] = p;
    }
}

}  // namespace k2r

using namespace k2r;

namespace k2r {
// suggest_fraction (fixed.rs:96-159) for n float tiles in DEVICE memory, two launches in all: (round, bits, status) per tile.
int suggest_fraction_batch(const dcdf_tile_desc* tiles, size_t n, int32_t* out_round, int32_t* out_bits, int32_t* status) {
    Runtime& rt = Runtime::get();
    if (!rt.ok) return DCDF_ERR_NO_DEVICE;
    if (n == 0) return DCDF_OK;
    const int dtype = tiles[0].dtype;
    std::vector<SfView> views(n);
    uint64_t most = 0;
    for (size_t i = 0; i < n; i++) {
        if (tiles[i].dtype != dtype) return DCDF_ERR_BAD_ARG;
        views[i] = SfView{tiles[i].base, tiles[i].stride_t, tiles[i].stride_r, tiles[i].stride_c, tiles[i].instants, tiles[i].rows, tiles[i].cols};
        most = std::max<uint64_t>(most, (uint64_t)tiles[i].instants * tiles[i].rows * tiles[i].cols);
        status[i] = DCDF_OK;
    }
    // workgroups per tile: enough to fill the card across the batch, no more than the largest tile has 256-cell pieces
    const uint32_t bpt = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((most + 255) / 256, ((uint64_t)rt.cus * 8 + n - 1) / n));
    const uint32_t grid = (uint32_t)(n * bpt);
    DevBuf d_views, d_bits, d_part;
    K2R_HIP(d_views.alloc(n * sizeof(SfView)));
    K2R_HIP(hipMemcpy(d_views.p, views.data(), n * sizeof(SfView), hipMemcpyHostToDevice));
    K2R_HIP(d_bits.alloc(n * 4));
    K2R_HIP(d_part.alloc((size_t)grid * sizeof(SfPartial)));
    std::vector<SfPartial> part(grid);
    if (dtype == DCDF_F32) hipLaunchKernelGGL((k_suggest<float, 1>), dim3(grid), dim3(256), 0, 0, d_views.as<SfView>(), d_bits.as<uint32_t>(), bpt, d_part.as<SfPartial>());
    else hipLaunchKernelGGL((k_suggest<double, 1>), dim3(grid), dim3(256), 0, 0, d_views.as<SfView>(), d_bits.as<uint32_t>(), bpt, d_part.as<SfPartial>());
    K2R_HIP(hipGetLastError());
    K2R_HIP(hipMemcpy(part.data(), d_part.p, (size_t)grid * sizeof(SfPartial), hipMemcpyDeviceToHost));
    std::vector<uint32_t> maxbits(n, 0);
    std::vector<char> all_nan(n, 0);
    for (size_t i = 0; i < n; i++) {
        bool seen = false;
        double mx = 0.0;
        for (uint32_t b = 0; b < bpt; b++) {
            const SfPartial& p = part[i * bpt + b];
            if (p.seen) {
                mx = seen ? std::max(mx, p.max) : p.max;
                seen = true;
            }
        }
        if (!seen) {  // all NaN (fixed.rs:124-127)
            all_nan[i] = 1;
            continue;
        }
        // whole_bits = 1 + log2(max).floor() as usize  (saturating float->usize cast: negative / NaN -> 0), fixed.rs:129
        const double lg = std::floor(std::log2(mx));
        if (lg > 61.0) {  // TOTAL_BITS - whole_bits underflows: the reference panics (fixed.rs:133)
            status[i] = DCDF_ERR_OVERFLOW;
            continue;
        }
        maxbits[i] = 62u - (1u + (lg > 0.0 ? (uint32_t)lg : 0u));
    }
    K2R_HIP(hipMemcpy(d_bits.p, maxbits.data(), n * 4, hipMemcpyHostToDevice));
    if (dtype == DCDF_F32) hipLaunchKernelGGL((k_suggest<float, 2>), dim3(grid), dim3(256), 0, 0, d_views.as<SfView>(), d_bits.as<uint32_t>(), bpt, d_part.as<SfPartial>());
    else hipLaunchKernelGGL((k_suggest<double, 2>), dim3(grid), dim3(256), 0, 0, d_views.as<SfView>(), d_bits.as<uint32_t>(), bpt, d_part.as<SfPartial>());
    K2R_HIP(hipGetLastError());
    K2R_HIP(hipMemcpy(part.data(), d_part.p, (size_t)grid * sizeof(SfPartial), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) {
        out_round[i] = 0;
        out_bits[i] = 0;
        if (all_nan[i] || status[i] != DCDF_OK) continue;
        uint32_t bits = 0, frac = 0;
        for (uint32_t b = 0; b < bpt; b++) {
            bits = std::max(bits, part[i * bpt + b].bits);
            frac |= part[i * bpt + b].frac;
        }
        out_round[i] = frac ? 1 : 0;
        out_bits[i] = (int32_t)(frac ? maxbits[i] : bits);
    }
    return DCDF_OK;
}
}  // namespace k2r

extern "C" int dcdf_suggest_fraction(const dcdf_tile_desc* tile, int mem, int32_t* out_round, int32_t* out_bits) {
    if (!tile || !out_round || !out_bits || !tile->base) return DCDF_ERR_BAD_ARG;
    if (tile->dtype != DCDF_F32 && tile->dtype != DCDF_F64) return DCDF_ERR_BAD_ARG;
    if (mem != DCDF_MEM_HOST && mem != DCDF_MEM_DEVICE) return DCDF_ERR_BAD_ARG;
    const uint64_t n = (uint64_t)tile->instants * tile->rows * tile->cols;
    if (n == 0) return DCDF_ERR_BAD_ARG;  // the reference unwraps the first element (fixed.rs:107)
    Runtime& rt = Runtime::get();
    if (!rt.ok) return DCDF_ERR_NO_DEVICE;
    K2R_HIP(hipSetDevice(rt.device));
    const size_t esz = tile->dtype == DCDF_F32 ? 4 : 8;
    dcdf_tile_desc dev = *tile;
    DevBuf staged;
    if (mem == DCDF_MEM_HOST) {  // pack the view densely (row by row) and upload it
        std::vector<char> dense(n * esz);
        const char* src = (const char*)tile->base;
        size_t o = 0;
        for (uint32_t t = 0; t < tile->instants; t++)
            for (uint32_t r = 0; r < tile->rows; r++) {
                const char* row = src + ((int64_t)t * tile->stride_t + (int64_t)r * tile->stride_r) * (int64_t)esz;
                if (tile->stride_c == 1) {
                    std::memcpy(&dense[o], row, (size_t)tile->cols * esz);
                    o += (size_t)tile->cols * esz;
                } else {
                    for (uint32_t c = 0; c < tile->cols; c++, o += esz) std::memcpy(&dense[o], row + (int64_t)c * tile->stride_c * (int64_t)esz, esz);
                }
            }
        K2R_HIP(staged.alloc(n * esz));
        K2R_HIP(hipMemcpy(staged.p, dense.data(), n * esz, hipMemcpyHostToDevice));
        dev.base = staged.p;
        dev.stride_t = (int64_t)tile->rows * tile->cols;
        dev.stride_r = tile->cols;
        dev.stride_c = 1;
    }
    int32_t st = DCDF_OK;
    const int rc = suggest_fraction_batch(&dev, 1, out_round, out_bits, &st);
    return rc != DCDF_OK ? rc : st;
}
