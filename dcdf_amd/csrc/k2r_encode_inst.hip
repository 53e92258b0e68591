// k2r_encode_inst.hip -- ONE instantiation of the fused encoder kernel per translation unit
// (compiled with -DK2R_L=<log2 sidelen> -DK2R_P=<0|1 padded> -DK2R_V=<0..4 row loader: generic, int32, float32, int64, float64>): the 36
// instantiations are large, fully unrolled kernels and build in parallel this way.
#include <hip/hip_runtime.h>

#include "k2r_encode.h"
#include "k2r_launch.h"

#define K2R_CAT2(a, b, c, d) a##_L##b##_P##c##_V##d
#define K2R_CAT(a, b, c, d) K2R_CAT2(a, b, c, d)

namespace k2r {

// Persistent workgroups: each pops tile indices from a device work queue until it is empty.
template <int LOG2S, bool PADDED, int VEC>
__global__ void __launch_bounds__(EncCfg<LOG2S>::NT)
k_encode(const TileArgs* __restrict__ tiles, TileResult* __restrict__ results, const uint32_t* __restrict__ order,
         uint32_t n, uint32_t* __restrict__ queue, uint64_t* __restrict__ lists) {
    using C = EncCfg<LOG2S>;
    __shared__ EncShared<C> sh;
    GpuExec<EncShared<C>, EncRegs, C::NT> ex(sh);
    // per-workgroup global scratch: the two overflow lists, the compact snapshot copy (16 u64 words per thread), the stash overflow
    uint64_t* listV = lists + (size_t)blockIdx.x * (C::MAXV + C::MAXT + 2 + 16 * C::NT + (5 * 4 + 3 * 16) * C::NBLK / 2);
    uint64_t* listM = listV + (C::MAXV + 1);
    uint32_t* scmp = (uint32_t*)(listM + (C::MAXT + 1));
    uint32_t* ovf = scmp + 32 * C::NT;  // stash records beyond the LDS pool's shares (EncPool::OVI_WORDS + OVQ_WORDS words)
    for (;;) {
        if (threadIdx.x == 0) sh.work = atomicAdd(queue, 1u);
        __syncthreads();
        // read back through readfirstlane: the compiler must KNOW the tile index is wave-uniform, otherwise every field
        // of TileArgs (base pointer, strides, shape...) is fetched per lane and all address arithmetic lands on the VALU
        const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.work);
        __syncthreads();
        if (w >= n) break;  // uniform: every wave of the workgroup leaves together
        const uint32_t ti = (uint32_t)__builtin_amdgcn_readfirstlane((int)order[w]);
        encode_chunk<C, PADDED, VEC>(ex, tiles[ti], &results[ti], listV, listM, scmp, ovf);
    }
}

hipError_t K2R_CAT(launch_encode, K2R_L, K2R_P, K2R_V)(const EncodeLaunch& L, hipStream_t stream) {
    using C = EncCfg<K2R_L>;
    hipLaunchKernelGGL((k_encode<K2R_L, (K2R_P != 0), K2R_V>), dim3(L.grid), dim3(C::NT), 0, stream, L.tiles,
                       L.results, L.order, L.n, L.queue, L.lists);
    return hipGetLastError();
}
int K2R_CAT(occ_encode, K2R_L, K2R_P, K2R_V)() {
    using C = EncCfg<K2R_L>;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_encode<K2R_L, (K2R_P != 0), K2R_V>, C::NT, 0) !=
        hipSuccess)
        nb = 1;
    return nb < 1 ? 1 : nb;
}

}  // namespace k2r
