// k2r_synth.hip -- device generator of the deterministic synthetic raster (dcdf_amd/synth.py).
// Bench/test utility: lets bench.py fill tens of GB in HBM without a host round trip, bit-identical to
// the numpy model (the cosine table is passed in by the caller so both sides use the same integers).
#include <hip/hip_runtime.h>

#include "k2r_runtime.h"

namespace k2r {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t rng(uint64_t seed, int64_t t, int64_t r, int64_t c, uint64_t salt) {
    return splitmix64(seed ^ ((uint64_t)t << 40) ^ ((uint64_t)r << 20) ^ (uint64_t)c ^ salt);
}

template <class T>
__global__ void __launch_bounds__(256)
k_synth(T* __restrict__ dst, const int32_t* __restrict__ costab, uint64_t seed, int64_t t0, int64_t nt, int64_t r0,
        int64_t nr, int64_t c0, int64_t nc, int wide) {
    __shared__ int32_t cs[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) cs[i] = costab[i];
    __syncthreads();
    const int64_t total = nt * nr * nc;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = c0 + i % nc, r = r0 + (i / nc) % nr;
        int64_t t = t0 + i / (nc * nr);
        if (t % 50 == 49 && t > 0) t -= 1;  // every 50th instant repeats its predecessor
        const int64_t base = cs[(3 * r + 5 * c) & 1023] + (cs[(7 * r - 2 * c) & 1023] >> 1) + (cs[(r + 11 * c) & 1023] >> 1);
        const int64_t ph = t % 365;
        const int64_t season = (ph < 183 ? ph : 365 - ph) * 64 / 183;
        const uint64_t h1 = rng(seed, t, r >> 4, c >> 4, 0x1111);
        const int64_t event = (h1 & 7) == 0 ? (int64_t)((h1 >> 8) % 513) - 256 : 0;
        const uint64_t h2 = rng(seed, t, r, c, 0x2222);
        const int64_t speckle = (h2 & 63) == 0 ? (int64_t)((h2 >> 8) % 17) - 8 : 0;
        int64_t v = base + season + event + speckle;
        const uint64_t h3 = rng(seed, t, r >> 6, c >> 6, 0x3333);
        if (h3 % 20 == 0) v = (int64_t)((h3 >> 16) % 4097) - 2048;
        dst[i] = wide ? (T)(v * 2 + 1) : (T)v;
    }
}

}  // namespace k2r

// Fills dst[(t1-t0)][(r1-r0)][(c1-c0)] (device memory, dense) with the synthetic raster cells of the given
// global coordinates.  dtype DCDF_I32 -> plain values; DCDF_I64 -> 2*v+1 ("fixed point with the NaN tag").
extern "C" int dcdf_synth_fill(void* dst_device, int32_t dtype, uint64_t seed, int64_t t0, int64_t t1, int64_t r0,
                               int64_t r1, int64_t c0, int64_t c1, const int32_t* costab_host) {
    using namespace k2r;
    if (!dst_device || !costab_host || t1 <= t0 || r1 <= r0 || c1 <= c0) return DCDF_ERR_BAD_ARG;
    if (dtype != DCDF_I32 && dtype != DCDF_I64) return DCDF_ERR_BAD_ARG;
    Runtime& rt = Runtime::get();
    if (!rt.ok) return DCDF_ERR_NO_DEVICE;
    DevBuf tab;
    K2R_HIP(tab.alloc(4096));
    K2R_HIP(hipMemcpy(tab.p, costab_host, 4096, hipMemcpyHostToDevice));
    const int grid = rt.cus * 8;
    if (dtype == DCDF_I32)
        hipLaunchKernelGGL(k_synth<int32_t>, dim3(grid), dim3(256), 0, 0, (int32_t*)dst_device, tab.as<int32_t>(), seed,
                           t0, t1 - t0, r0, r1 - r0, c0, c1 - c0, 0);
    else
        hipLaunchKernelGGL(k_synth<int64_t>, dim3(grid), dim3(256), 0, 0, (int64_t*)dst_device, tab.as<int32_t>(), seed,
                           t0, t1 - t0, r0, r1 - r0, c0, c1 - c0, 1);
    K2R_HIP(hipGetLastError());
    K2R_HIP(hipDeviceSynchronize());
    return DCDF_OK;
}


// ---- PMC calibration (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access
// pattern before trusting an absolute") --------------------------------------------------------------------
// Reads n_tiles dense [instants,256,256] int32 tiles ONCE with exactly the encoder's access pattern (thread =
// Morton 8x8 block, four 16-byte row pieces per 4x4 sub-block) and folds them into a checksum, so that
// rocprofv3's FETCH_SIZE for this launch can be compared with the known byte count.
namespace k2r {
__global__ void __launch_bounds__(1024)
k_calib_read(const int32_t* __restrict__ base, uint32_t n_tiles, uint32_t instants, unsigned long long* __restrict__ sink) {
    uint32_t br, bc;
    morton_decode(threadIdx.x, br, bc);
    const uint32_t r0 = br * 8, c0 = bc * 8;
    unsigned long long acc = 0;
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        for (uint32_t i = 0; i < instants; i++) {
            const int32_t* ib = base + ((size_t)t * instants + i) * 65536;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t rj = r0 + 4 * (j >> 1), cj = c0 + 4 * (j & 1);
#pragma unroll
                for (int dr = 0; dr < 4; dr++) {
                    const int4 a = *(const int4*)(ib + (rj + dr) * 256 + cj);
                    acc += (unsigned)(a.x ^ a.y ^ a.z ^ a.w);
                }
            }
        }
    }
    if (acc == 0x123456789abcdefull) sink[0] = acc;  // keep the loads alive
}
}  // namespace k2r

extern "C" int dcdf_calib_read(const void* tiles_device, uint32_t n_tiles, uint32_t instants) {
    using namespace k2r;
    Runtime& rt = Runtime::get();
    if (!rt.ok) return DCDF_ERR_NO_DEVICE;
    if (!tiles_device || n_tiles == 0 || instants == 0) return DCDF_ERR_BAD_ARG;
    DevBuf sink;
    K2R_HIP(sink.alloc(8));
    hipLaunchKernelGGL(k_calib_read, dim3(rt.cus), dim3(1024), 0, 0, (const int32_t*)tiles_device, n_tiles, instants,
                       sink.as<unsigned long long>());
    K2R_HIP(hipGetLastError());
    K2R_HIP(hipDeviceSynchronize());
    return DCDF_OK;
}

// ---- device memory for callers without a HIP binding of their own (the ctypes mirror, tests, tools): plain
// hipMalloc / hipFree / hipMemcpy behind the C ABI, so that device-resident sessions need no third-party runtime ----
extern "C" int dcdf_device_alloc(size_t bytes, void** out) {
    using namespace k2r;
    if (!out) return DCDF_ERR_BAD_ARG;
    if (!Runtime::get().ok) return DCDF_ERR_NO_DEVICE;
    void* p = nullptr;
    K2R_HIP(hipMalloc(&p, bytes ? bytes : 16));
    *out = p;
    return DCDF_OK;
}
extern "C" int dcdf_device_free(void* p) {
    using namespace k2r;
    if (p) K2R_HIP(hipFree(p));
    return DCDF_OK;
}
extern "C" int dcdf_device_copy(void* dst, const void* src, size_t bytes, int to_device) {
    using namespace k2r;
    if ((!dst || !src) && bytes) return DCDF_ERR_BAD_ARG;
    if (!Runtime::get().ok) return DCDF_ERR_NO_DEVICE;
    if (bytes) K2R_HIP(hipMemcpy(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
    return DCDF_OK;
}
