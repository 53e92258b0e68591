// k2r_runtime.h -- host runtime shared by the C-ABI translation units (device discovery, error mapping).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <mutex>
#include <string>

#include "../../include/dcdf_k2r.h"
#include "k2r_common.h"

namespace k2r {

// Tree depth as the reference computes it: ceil(ln(m) / ln(k)) in f64 (snapshot.rs:118-119, log.rs:124-125,
// superchunk.rs:98-101).  NOT always the smallest H with k^H >= m: ln(125)/ln(5) = 3.0000000000000004, so a 125-wide tile
// with k = 5 gets 4 levels and sidelen 625 (likewise k = 6, m = 216 -> 1296); a drop-in has to build and accept the same tree.
inline uint32_t ref_levels(uint64_t m, uint32_t k) {
    if (m <= 1) return 0;
    const double e = std::ceil(std::log((double)m) / std::log((double)k));
    return e > 0 ? (uint32_t)e : 0u;
}
inline uint64_t ref_sidelen(uint64_t m, uint32_t k) {
    uint64_t s = 1;
    for (uint32_t i = ref_levels(m, k); i > 0; i--) s *= k;
    return s;
}

struct Runtime {
    bool ok = false;
    int device = 0;
    int cus = 0;
    std::string name;
    static Runtime& get();  // lazily initialised; ok == false when no usable device
};

// RAII device buffer
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    hipError_t alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        else p = nullptr;
        return e;
    }
    template <class T>
    T* as() const { return (T*)p; }
};

inline int map_status(int32_t st) {
    switch (st) {
        case ST_OK: return DCDF_OK;
        case ST_NONFINITE: return DCDF_ERR_NONFINITE;
        case ST_PRECISION: return DCDF_ERR_PRECISION;
        case ST_OVERFLOW: return DCDF_ERR_OVERFLOW;
        case ST_UNSUPPORTED: return DCDF_ERR_UNSUPPORTED;
        case ST_BAD_ARG: return DCDF_ERR_BAD_ARG;
        case ST_INTERNAL: return DCDF_ERR_INTERNAL;
        default: return DCDF_ERR_INTERNAL;  // unknown kernel status
    }
}

// HIP failures keep their identity: out-of-memory, no-device and launch/runtime failures map to distinct DCDF codes,
// and the raw hipError_t of the last failure on this thread can be read back with dcdf_last_hip_error().
inline int& last_hip_error() {
    static thread_local int e = 0;
    return e;
}
inline int map_hip_error(hipError_t e) {
    last_hip_error() = (int)e;
    switch (e) {
        case hipErrorOutOfMemory: return DCDF_ERR_NOMEM;
        case hipErrorNoDevice:
        case hipErrorInvalidDevice:
        case hipErrorInsufficientDriver:
        case hipErrorNotInitialized: return DCDF_ERR_NO_DEVICE;
        case hipErrorInvalidValue:
        case hipErrorInvalidDevicePointer:
        case hipErrorInvalidMemcpyDirection: return DCDF_ERR_HIP_INVALID;
        default: return DCDF_ERR_HIP;  // launch failures, illegal address, ECC ... (see dcdf_last_hip_error)
    }
}
#define K2R_HIP(call)                                  \
    do {                                               \
        hipError_t _e = (call);                        \
        if (_e != hipSuccess) return k2r::map_hip_error(_e); \
    } while (0)

}  // namespace k2r
