// tools/ubench_store.hip -- what do the emission passes' stores cost?  (round 4, experiment E; a measurement tool, not product code)
//
// One 1024-thread workgroup per CU issues N store wave-instructions into its own 64 KB output window (the size of one Log of
// the benchmark), in the shapes the encoder's emission uses, with a few VALU instructions between two stores:
//   shape 0: one byte per lane, lanes at pseudo-random offsets of the window        (second bytes, arrival order)
//   shape 1: one byte per lane, consecutive lanes 1..3 bytes apart                 (second bytes, level order)
//   shape 2: one unaligned dword per lane at pseudo-random offsets                 (plane-0 bytes of four siblings, arrival order)
//   shape 3: one unaligned dword per lane, consecutive lanes 4 bytes apart         (the same in level order)
//   shape 4: LDS atomic OR, one word per lane at pseudo-random words of a 10 KB bitmap   (continuation bits, arrival order)
//   shape 5: LDS atomic OR, eight consecutive lanes per word                             (continuation bits, level order)
// Reported: cycles of the CU per wave-instruction (all 16 waves issuing), from s_memtime.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                         \
    do {                                                                 \
        hipError_t e_ = (x);                                             \
        if (e_ != hipSuccess) {                                          \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            std::exit(1);                                                \
        }                                                                \
    } while (0)

template <int SHAPE>
__global__ void __launch_bounds__(1024) k_store(uint8_t* __restrict__ out, uint32_t iters, uint32_t alu, uint64_t* __restrict__ cycles) {
    __shared__ uint32_t bm[2560];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint8_t* win = out + (size_t)blockIdx.x * 65536u;
    for (uint32_t i = tid; i < 2560; i += 1024) bm[i] = 0;
    __syncthreads();
    uint32_t h = tid * 2654435761u + 12345u, x = tid;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t it = 0; it < iters; it++) {
        h = h * 1664525u + 1013904223u;
        const uint32_t rnd = (h >> 8) & 0xffffu;
        const uint32_t seq = ((it * 16u + wave) * 64u + lane);  // a running level-order position
        for (uint32_t a = 0; a < alu; a++) x = (x ^ (x >> 3)) + it;
#if defined(__HIP_DEVICE_COMPILE__)
        if (SHAPE == 0) {
            *(__attribute__((address_space(1))) uint8_t*)(win + (rnd & 0xfff0u) + (lane & 15u)) = (uint8_t)x;
        } else if (SHAPE == 1) {
            *(__attribute__((address_space(1))) uint8_t*)(win + ((seq * 2u) & 0xffffu)) = (uint8_t)x;
        } else if (SHAPE == 2) {
            typedef uint32_t __attribute__((aligned(1))) u32u;
            *(__attribute__((address_space(1))) u32u*)(win + (rnd & 0xfff8u) + 1u) = x;
        } else if (SHAPE == 3) {
            typedef uint32_t __attribute__((aligned(1))) u32u;
            *(__attribute__((address_space(1))) u32u*)(win + ((seq * 4u) & 0xfff0u) + 4u * (lane & 3u) + 1u) = x;
        } else if (SHAPE == 4) {
            atomicOr(&bm[rnd % 2560u], 1u << (x & 31u));
        } else {
            atomicOr(&bm[(seq >> 3) % 2560u], 0xfu << (4u * (lane & 7u)));
        }
#else
        (void)rnd; (void)seq; (void)win;
#endif
    }
    __syncthreads();
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    if (x == 0x12345678u) out[tid] = (uint8_t)bm[tid];
}

template <int SHAPE>
static void run(uint8_t* out, uint64_t* cyc, uint32_t iters, uint32_t alu, const char* what) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    uint64_t h[256];
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_store<SHAPE>), dim3(256), dim3(1024), 0, 0, out, iters, alu, cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double mean = 0;
    for (int i = 0; i < 256; i++) mean += (double)h[i];
    mean /= 256;
    std::printf("shape %d alu %3u: %8.3f ms, %8.1f cycles per wave-instruction and CU (16 waves issuing: %6.1f per wave)   %s\n", SHAPE, alu, best,
                mean / (iters * 16.0), mean / iters, what);
    std::fflush(stdout);
}

int main() {
    uint8_t* out;
    uint64_t* cyc;
    CHECK(hipMalloc(&out, 256 * 65536 + 4096));  // (+ slack: the unaligned dword shapes end up to 5 bytes past a window)
    CHECK(hipMalloc(&cyc, 256 * 8));
    for (uint32_t alu : {0u, 8u, 32u}) {
        run<0>(out, cyc, 2000, alu, "byte, scattered");
        run<1>(out, cyc, 2000, alu, "byte, level order");
        run<2>(out, cyc, 2000, alu, "dword, scattered");
        run<3>(out, cyc, 2000, alu, "dword, level order");
        run<4>(out, cyc, 2000, alu, "LDS atomic OR, scattered");
        run<5>(out, cyc, 2000, alu, "LDS atomic OR, 8 lanes per word");
    }
    return 0;
}
