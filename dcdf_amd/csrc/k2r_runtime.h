// k2r_runtime.h -- host runtime shared by the C-ABI translation units (device discovery, error mapping).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

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

// A few large device blocks kept between calls.  hipMalloc / hipFree of gigabytes cost tens of milliseconds each (and hipFree
// synchronises the device); a caller that appends slice after slice (Variable::append -> dcdf_superchunk_build) asks for the
// same sizes again and again.  Only the big, write-before-read buffers of a session go through it (output slots, packing and
// scratch areas: DevBuf::alloc_pooled); contents are never assumed.  K2R_POOL=0 disables it.  The pool is never destroyed: HIP
// calls at static-destruction time are not safe.
struct DevPool {
    struct Blk {
        void* p;
        size_t n;
    };
    std::mutex mu;
    std::vector<Blk> blocks;  // oldest first
    size_t held = 0;
    static constexpr size_t kMinBytes = 16u << 20, kMaxHeld = 12ull << 30, kMaxBlocks = 12;
    static DevPool& get() {
        static DevPool* pool = new DevPool();
        return *pool;
    }
    static bool enabled() {
        static const bool on = [] {
            const char* e = std::getenv("K2R_POOL");
            return !(e && e[0] == '0');
        }();
        return on;
    }
    void* take(size_t n, size_t* got) {  // the smallest kept block of n .. 1.5 n bytes
        std::lock_guard<std::mutex> lk(mu);
        size_t best = blocks.size();
        for (size_t i = 0; i < blocks.size(); i++)
            if (blocks[i].n >= n && blocks[i].n <= n + n / 2 && (best == blocks.size() || blocks[i].n < blocks[best].n)) best = i;
        if (best == blocks.size()) return nullptr;
        void* p = blocks[best].p;
        *got = blocks[best].n;
        held -= blocks[best].n;
        blocks.erase(blocks.begin() + (long)best);
        return p;
    }
    // every kept block back to the driver (out of memory elsewhere in the process; dcdf_device_pool_trim)
    size_t drain() {
        std::vector<Blk> all;
        {
            std::lock_guard<std::mutex> lk(mu);
            all.swap(blocks);
            held = 0;
        }
        size_t freed = 0;
        for (const Blk& b : all) {
            (void)hipFree(b.p);
            freed += b.n;
        }
        return freed;
    }
    void give(void* p, size_t n) {
        // hipFree used to order the release behind everything queued on the device; a block parked here may be handed to
        // another host thread at once, so it must be just as idle (error paths release with copies or kernels still queued)
        (void)hipDeviceSynchronize();
        std::vector<Blk> drop;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (n < kMinBytes || n > kMaxHeld) drop.push_back(Blk{p, n});
            else {
                blocks.push_back(Blk{p, n});
                held += n;
                while (held > kMaxHeld || blocks.size() > kMaxBlocks) {
                    drop.push_back(blocks.front());
                    held -= blocks.front().n;
                    blocks.erase(blocks.begin());
                }
            }
        }
        for (const Blk& b : drop) (void)hipFree(b.p);
    }
};

// RAII device buffer
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    bool pooled = false;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) {
            if (pooled) DevPool::get().give(p, bytes);
            else (void)hipFree(p);
        }
        p = nullptr;
        bytes = 0;
        pooled = false;
    }
    hipError_t alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess && DevPool::enabled() && DevPool::get().drain() > 0) {  // memory idling in the pool: retry once
            (void)hipGetLastError();
            e = hipMalloc(&p, n);
        }
        if (e == hipSuccess) bytes = n;
        else p = nullptr;
        return e;
    }
    // for large buffers that are written before they are read (see DevPool); `bytes` may come back larger than asked
    hipError_t alloc_pooled(size_t n) {
        release();
        if (n == 0) n = 16;
        if (DevPool::enabled() && n >= DevPool::kMinBytes) {
            size_t got = 0;
            if (void* q = DevPool::get().take(n, &got)) {
                p = q;
                bytes = got;
                pooled = true;
                return hipSuccess;
            }
        }
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) {
            bytes = n;
            pooled = DevPool::enabled() && n >= DevPool::kMinBytes;
        } else {
            p = nullptr;
            if (DevPool::enabled()) {  // out of memory with blocks parked in the pool: give them back and try once more
                if (DevPool::get().drain() > 0) {
                    (void)hipGetLastError();
                    e = hipMalloc(&p, n);
                    if (e == hipSuccess) {
                        bytes = n;
                        pooled = n >= DevPool::kMinBytes;
                    } else {
                        p = nullptr;
                    }
                }
            }
        }
        return e;
    }
    template <class T>
    T* as() const { return (T*)p; }
};

}  // namespace k2r
struct dcdf_encoder;
namespace k2r {
// The encoded bytes of a finished session, device slots -> pinned double buffer -> wherever `dst(tile, len)` says (called once
// per tile with bytes, from worker threads; it may allocate the destination there).  k2r_capi_encode.hip.
// `landed(tile)`, if given, runs on the worker thread right after a tile's bytes have been copied to their destination.
int encoder_download(dcdf_encoder* e, const std::function<uint8_t*(size_t, uint64_t)>& dst,
                     const std::function<void(size_t)>& landed = nullptr);
// runs f(0..n) on a few host threads
void host_parallel_for(size_t n, const std::function<void(size_t)>& f);

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
