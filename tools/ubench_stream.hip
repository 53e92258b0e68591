// tools/ubench_stream.hip -- what bounds phase 1 of the encoder?  (round 4, experiment D; a measurement tool, not product code)
//
// One 1024-thread workgroup per CU streams [instants, 256, 256] int32 tiles of a [T, 4096, 4096] raster exactly as phase 1 of
// k2r_encode.h does -- per wave a group of 16-byte row loads, a wait, a little arithmetic -- with the lane -> cell mapping,
// the number of loads in flight per wave, the number of busy CUs and the amount of arithmetic between the loads as parameters.
//
//   pattern 0: lane = 8x8 block in Morton order, sub-blocks one after another (16 B pieces, 32 B apart: half a line per touch)
//   pattern 1: lane = 4x4 sub-block in Morton order (8 lanes cover 128 contiguous bytes of a row)
//   pattern 2: lane = 4 cells of a row, a wave-instruction = one 1 KB tile row
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_stream tools/ubench_stream.hip       run: tools/ubench_stream
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
            std::exit(1);                                                             \
        }                                                                             \
    } while (0)

__device__ __forceinline__ uint32_t compact_even(uint32_t x) {
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0f0f0f0fu;
    x = (x | (x >> 4)) & 0x00ff00ffu;
    x = (x | (x >> 8)) & 0x0000ffffu;
    return x;
}

typedef __attribute__((address_space(1))) const char* gptr;

// DEPTH = sub-blocks (groups of 4 row loads [+ 2 copy loads]) requested before the first one is consumed
template <int PATTERN, int DEPTH, bool COPY>
__global__ void __launch_bounds__(1024) k_stream(const int32_t* __restrict__ raster, const uint32_t* __restrict__ copy, uint32_t tiles_x,
                                                 uint32_t n_tiles, uint32_t instants, uint32_t alu, uint32_t* __restrict__ sink,
                                                 uint32_t* __restrict__ queue) {
    __shared__ uint32_t work;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t acc = 0;
    for (;;) {
        if (tid == 0) work = atomicAdd(queue, 1u);
        __syncthreads();
        const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)work);
        __syncthreads();
        if (w >= n_tiles) break;
        const uint32_t ty = w / tiles_x, tx = w % tiles_x;
        const uint32_t sr = tiles_x * 256u;  // row stride in cells
        for (uint32_t inst = 0; inst < instants; inst++) {
            const int32_t* ib = raster + ((size_t)inst * sr * sr + (size_t)ty * 256u * sr + (size_t)tx * 256u);
            const uint32_t* cb = copy + (size_t)blockIdx.x * 32768u;  // the workgroup's compact copy: 128 KB
            int4 buf[DEPTH][4];
            uint4 cbuf[DEPTH][2];
            auto issue = [&](int slot, uint32_t step) {  // step = 0..3: the step-th quarter of the tile
                uint32_t r, c, cslot;
                if (PATTERN == 0) {
                    const uint32_t br = compact_even(tid >> 1), bc = compact_even(tid);
                    r = br * 8 + 4 * (step >> 1);
                    c = bc * 8 + 4 * (step & 1);
                    cslot = step * 1024u + tid;
                } else if (PATTERN == 1) {
                    const uint32_t m2 = (16u * step + wave) * 64u + lane;  // height-2 node in Morton order
                    r = compact_even(m2 >> 1) * 4;
                    c = compact_even(m2) * 4;
                    cslot = m2;
                } else {
                    r = (step * 16u + wave) * 4u;  // four consecutive rows, lane = 4 cells of each
                    c = lane * 4;
                    cslot = (step * 16u + wave) * 64u + lane;
                }
#pragma unroll
                for (int dr = 0; dr < 4; dr++) {
                    const uint32_t ob = ((r + dr) * sr + c) << 2;
#if defined(__HIP_DEVICE_COMPILE__)
                    buf[slot][dr] = *(__attribute__((address_space(1))) const int4*)((gptr)ib + ob);
#else
                    (void)ob;
#endif
                }
                if (COPY) {
#if defined(__HIP_DEVICE_COMPILE__)
                    cbuf[slot][0] = *(__attribute__((address_space(1))) const uint4*)((gptr)cb + cslot * 32u);
                    cbuf[slot][1] = *(__attribute__((address_space(1))) const uint4*)((gptr)cb + cslot * 32u + 16u);
#else
                    (void)cslot;
#endif
                }
            };
            auto consume = [&](int slot) {
                uint32_t x = 0;
#pragma unroll
                for (int dr = 0; dr < 4; dr++) x ^= (uint32_t)(buf[slot][dr].x ^ buf[slot][dr].y ^ buf[slot][dr].z ^ buf[slot][dr].w);
                if (COPY) x ^= cbuf[slot][0].x ^ cbuf[slot][0].w ^ cbuf[slot][1].y ^ cbuf[slot][1].z;
                for (uint32_t i = 0; i < alu; i++) x = x * 1664525u + 1013904223u;  // dependent integer work between the loads
                acc ^= x;
            };
            if (DEPTH == 1) {
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue(0, (uint32_t)s);
                    consume(0);
                }
            } else if (DEPTH == 2) {
                issue(0, 0);
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (s < 3) issue((s + 1) & 1, (uint32_t)(s + 1));
                    consume(s & 1);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 4; s++) issue(s, (uint32_t)s);
#pragma unroll
                for (int s = 0; s < 4; s++) consume(s);
            }
        }
    }
    if (acc == 0x12345678u) sink[blockIdx.x * 1024u + tid] = acc;
}

template <int P, int D, bool C>
static double run(const int32_t* raster, const uint32_t* copy, uint32_t tiles_x, uint32_t n_tiles, uint32_t instants, uint32_t alu,
                  uint32_t grid, uint32_t* sink, uint32_t* queue) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipMemset(queue, 0, 4));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_stream<P, D, C>), dim3(grid), dim3(1024), 0, 0, raster, copy, tiles_x, n_tiles, instants, alu, sink, queue);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const uint32_t tiles_x = 16, T = 32;  // [32, 4096, 4096] int32 = 2 GiB
    const size_t cells = (size_t)T * 4096 * 4096;
    int32_t* raster;
    uint32_t *copy, *sink, *queue;
    CHECK(hipMalloc(&raster, cells * 4));
    CHECK(hipMemset(raster, 1, cells * 4));
    CHECK(hipMalloc(&copy, 256 * 131072));
    CHECK(hipMemset(copy, 2, 256 * 131072));
    CHECK(hipMalloc(&sink, 256 * 1024 * 4));
    CHECK(hipMalloc(&queue, 4));
    std::printf("%-8s %-6s %-5s %-5s %-5s %9s %12s %12s\n", "pattern", "depth", "copy", "grid", "alu", "ms", "GB/s total", "GB/s per CU");
    auto report = [&](int p, int d, bool c, uint32_t grid, uint32_t alu, double ms, uint32_t n_tiles) {
        const double bytes = (double)n_tiles * T * 65536.0 * (c ? 6.0 : 4.0);
        std::printf("%-8d %-6d %-5d %-5u %-5u %9.3f %12.1f %12.2f\n", p, d, (int)c, grid, alu, ms, bytes / ms / 1e6, bytes / ms / 1e6 / grid);
        std::fflush(stdout);
    };
#define RUN(P, D, C, GRID, ALU, NT) report(P, D, C, GRID, ALU, run<P, D, C>(raster, copy, tiles_x, NT, T, ALU, GRID, sink, queue), NT)
    for (uint32_t grid : {256u, 64u, 16u}) {
        const uint32_t nt = grid;  // one tile (32 instants) per workgroup
        for (uint32_t alu : {0u, 64u, 256u}) {
            RUN(0, 1, true, grid, alu, nt);
            RUN(1, 1, true, grid, alu, nt);
            RUN(2, 1, true, grid, alu, nt);
            RUN(0, 2, true, grid, alu, nt);
            RUN(1, 2, true, grid, alu, nt);
            RUN(1, 4, true, grid, alu, nt);
            RUN(2, 4, true, grid, alu, nt);
        }
        RUN(0, 1, false, grid, 0, nt);
        RUN(1, 1, false, grid, 0, nt);
        RUN(1, 4, false, grid, 0, nt);
    }
    return 0;
}
