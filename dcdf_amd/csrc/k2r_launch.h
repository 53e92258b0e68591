// k2r_launch.h -- host-side launch descriptors shared by k2r_kernels.hip and the C-ABI runtime.
#pragma once
#include <hip/hip_runtime.h>

#include "k2r_common.h"

namespace k2r {

struct EncClass {  // one kernel instantiation
    int log2s;     // 4..8 (sidelen 8 and below go to the universal kernel)
    bool padded;
    int vec;       // 0 generic loads, 1 int32 vector loads, 2 float32, 3 int64, 4 float64 vector loads (k2r_encode.h load_sub16)
    bool operator==(const EncClass& o) const { return log2s == o.log2s && padded == o.padded && vec == o.vec; }
};

struct EncodeLaunch {
    const TileArgs* tiles;  // device
    TileResult* results;    // device
    const uint32_t* order;  // device: tile indices of this class
    uint32_t n;             // tiles in this class
    uint32_t* queue;        // device: zeroed work counter
    uint64_t* lists;        // device: grid * encode_list_words(cls) u64
    uint32_t grid;
};

hipError_t launch_encode(const EncClass& cls, const EncodeLaunch& L, hipStream_t stream);
// the universal encoder (k2r_generic.hip): any k, sidelen = k^H, full int64 values; one workgroup per tile, `scratch_per_wg`
// bytes of global scratch each (generic_scratch_bytes)
hipError_t launch_encode_generic(const EncodeLaunch& L, uint8_t* scratch, uint64_t scratch_per_wg, uint32_t k, uint32_t H, hipStream_t stream);
uint64_t generic_scratch_bytes(uint32_t k, uint32_t H);
int encode_blocks_per_cu(const EncClass& cls);
size_t encode_list_words(const EncClass& cls);
int encode_threads(const EncClass& cls);

}  // namespace k2r
