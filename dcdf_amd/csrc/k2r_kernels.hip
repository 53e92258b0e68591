// k2r_kernels.hip -- dispatch table over the encoder instantiations (k2r_encode_inst.hip).
#include <hip/hip_runtime.h>

#include "k2r_launch.h"

namespace k2r {

#define K2R_DECL(L, P, V)                                                      \
    hipError_t launch_encode_L##L##_P##P##_V##V(const EncodeLaunch&, hipStream_t); \
    int occ_encode_L##L##_P##P##_V##V();
#define K2R_DECL3(L) K2R_DECL(L, 0, 0) K2R_DECL(L, 1, 0) K2R_DECL(L, 0, 1) K2R_DECL(L, 0, 2) K2R_DECL(L, 0, 3) K2R_DECL(L, 0, 4)
K2R_DECL3(4) K2R_DECL3(5) K2R_DECL3(6) K2R_DECL3(7) K2R_DECL3(8)

typedef hipError_t (*launch_fn)(const EncodeLaunch&, hipStream_t);
typedef int (*occ_fn)();
#define K2R_ROW(L) {launch_encode_L##L##_P0_V0, launch_encode_L##L##_P1_V0, launch_encode_L##L##_P0_V1, launch_encode_L##L##_P0_V2, launch_encode_L##L##_P0_V3, launch_encode_L##L##_P0_V4}
#define K2R_OROW(L) {occ_encode_L##L##_P0_V0, occ_encode_L##L##_P1_V0, occ_encode_L##L##_P0_V1, occ_encode_L##L##_P0_V2, occ_encode_L##L##_P0_V3, occ_encode_L##L##_P0_V4}
static const launch_fn kLaunch[5][6] = {K2R_ROW(4), K2R_ROW(5), K2R_ROW(6), K2R_ROW(7), K2R_ROW(8)};
static const occ_fn kOcc[5][6] = {K2R_OROW(4), K2R_OROW(5), K2R_OROW(6), K2R_OROW(7), K2R_OROW(8)};

static int variant(const EncClass& c) { return c.padded ? 1 : (c.vec >= 1 && c.vec <= 4 ? 1 + c.vec : 0); }

hipError_t launch_encode(const EncClass& cls, const EncodeLaunch& L, hipStream_t stream) {
    if (cls.log2s < 4 || cls.log2s > 8) return hipErrorInvalidValue;
    return kLaunch[cls.log2s - 4][variant(cls)](L, stream);
}
int encode_blocks_per_cu(const EncClass& cls) {
    if (cls.log2s < 4 || cls.log2s > 8) return 1;
    return kOcc[cls.log2s - 4][variant(cls)]();
}
size_t encode_list_words(const EncClass& cls) {  // u64 words of global scratch per workgroup (k2r_encode_inst.hip)
    const int H = cls.log2s;
    const size_t maxv = ((1u << (2 * (H + 1))) - 1) / 3, maxt = ((1u << (2 * H)) - 1) / 3;
    const size_t nblk = (size_t)1 << (2 * (H - 3));
    // + the compact snapshot copy (128 B per thread) + the stash overflow area (EncPool::OVI_WORDS + OVQ_WORDS 32-bit words)
    return maxv + maxt + 2 + 16 * nblk + (5 * 4 + 3 * 16) * nblk / 2;
}
int encode_threads(const EncClass& cls) { return 1 << (2 * (cls.log2s - 3)); }

}  // namespace k2r
