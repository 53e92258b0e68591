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

// Device blocks kept between calls.  hipMalloc / hipFree of gigabytes cost tens of milliseconds each, the dozen small tables of a
// session a few hundred microseconds each -- and hipFree waits for everything queued on the device, uploads of the NEXT band
// included (the banded superchunk assembly lost more to that than it gained from the overlap).  A caller that appends slice after
// slice (Variable::append -> dcdf_superchunk_build) asks for the same sizes again and again, so released blocks are parked: large
// ones (>= 16 MB) are handed out again for requests of 2/3 .. 1 of their size, small ones by power-of-two size class.  Contents are
// never assumed.  A parked block carries an event recorded on the null stream when it was released -- behind everything queued
// on the session streams at that moment -- and whoever takes it waits for that event first: a release never blocks the host, and
// a block is idle before it is used again, error paths included.  K2R_POOL=0 disables it.  The pool is never destroyed: HIP calls
// at static-destruction time are not safe.
struct DevPool {
    struct Blk {
        void* p;
        size_t n;
        hipEvent_t ev;
    };
    std::mutex mu;
    std::vector<Blk> large, small;  // oldest first
    size_t held_large = 0, held_small = 0;
    static constexpr size_t kMinBytes = 16u << 20, kMaxHeld = 12ull << 30, kMaxBlocks = 32;
    static constexpr size_t kMaxSmallHeld = 256u << 20, kMaxSmallBlocks = 512;
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
    static size_t size_class(size_t n) {  // what a request of n bytes allocates
        if (n >= kMinBytes) return n;
        size_t c = 256;
        while (c < n) c <<= 1;
        return c;
    }
    void* take(size_t n, size_t* got) {  // large: the smallest kept block of n .. 1.5 n bytes; small: one of the request's class
        Blk b{nullptr, 0, nullptr};
        {
            std::lock_guard<std::mutex> lk(mu);
            if (n >= kMinBytes) {
                size_t best = large.size();
                for (size_t i = 0; i < large.size(); i++)
                    if (large[i].n >= n && large[i].n <= n + n / 2 && (best == large.size() || large[i].n < large[best].n)) best = i;
                if (best == large.size()) return nullptr;
                b = large[best];
                held_large -= b.n;
                large.erase(large.begin() + (long)best);
            } else {
                const size_t c = size_class(n);
                size_t at = small.size();
                for (size_t i = small.size(); i-- > 0;)
                    if (small[i].n == c) {
                        at = i;
                        break;
                    }
                if (at == small.size()) return nullptr;
                b = small[at];
                held_small -= b.n;
                small.erase(small.begin() + (long)at);
            }
        }
        if (b.ev) {
            if (hipEventSynchronize(b.ev) != hipSuccess) (void)hipDeviceSynchronize();
            (void)hipEventDestroy(b.ev);
        }
        *got = b.n;
        return b.p;
    }
    static void drop(const Blk& b) {
        if (b.ev) (void)hipEventDestroy(b.ev);
        (void)hipFree(b.p);  // (waits for the device by itself)
    }
    // every kept block back to the driver (out of memory elsewhere in the process; dcdf_device_pool_trim)
    size_t drain() {
        std::vector<Blk> all;
        {
            std::lock_guard<std::mutex> lk(mu);
            all.swap(large);
            all.insert(all.end(), small.begin(), small.end());
            small.clear();
            held_large = held_small = 0;
        }
        size_t freed = 0;
        for (const Blk& b : all) {
            drop(b);
            freed += b.n;
        }
        return freed;
    }
    void give(void* p, size_t n) {
        Blk b{p, n, nullptr};
        if (hipEventCreateWithFlags(&b.ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(b.ev, nullptr) != hipSuccess) {
            (void)hipGetLastError();
            if (b.ev) (void)hipEventDestroy(b.ev);
            b.ev = nullptr;
            (void)hipDeviceSynchronize();  // (no event to wait for later: idle now)
        }
        std::vector<Blk> out;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (n >= kMinBytes) {
                if (n > kMaxHeld) out.push_back(b);
                else {
                    large.push_back(b);
                    held_large += n;
                    while (held_large > kMaxHeld || large.size() > kMaxBlocks) {
                        out.push_back(large.front());
                        held_large -= large.front().n;
                        large.erase(large.begin());
                    }
                }
            } else {
                small.push_back(b);
                held_small += n;
                while (held_small > kMaxSmallHeld || small.size() > kMaxSmallBlocks) {
                    out.push_back(small.front());
                    held_small -= small.front().n;
                    small.erase(small.begin());
                }
            }
        }
        for (const Blk& d : out) drop(d);
    }
};

// HIP streams kept between calls: creating one costs a millisecond and destroying one stalls every other thread's HIP calls for
// several (measured in the banded superchunk assembly, where a stream died while other threads were mid-band).  A stream handed
// back must be idle.  Never destroyed.
struct StreamPool {
    std::mutex mu;
    std::vector<hipStream_t> blocking, nonblocking;
    static StreamPool& get() {
        static StreamPool* sp = new StreamPool();
        return *sp;
    }
    hipError_t take(bool nonblock, hipStream_t* out) {
        {
            std::lock_guard<std::mutex> lk(mu);
            auto& v = nonblock ? nonblocking : blocking;
            if (!v.empty()) {
                *out = v.back();
                v.pop_back();
                return hipSuccess;
            }
        }
        return hipStreamCreateWithFlags(out, nonblock ? hipStreamNonBlocking : hipStreamDefault);
    }
    void give(bool nonblock, hipStream_t s) {
        if (!s) return;
        (void)hipStreamSynchronize(s);
        std::lock_guard<std::mutex> lk(mu);
        (nonblock ? nonblocking : blocking).push_back(s);
    }
};

// RAII device buffer (from the pool above; `bytes` may come back larger than asked)
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
        const bool pool = DevPool::enabled();
        if (pool) {
            size_t got = 0;
            if (void* q = DevPool::get().take(n, &got)) {
                p = q;
                bytes = got;
                pooled = true;
                return hipSuccess;
            }
            n = DevPool::size_class(n);
        }
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess && pool && DevPool::get().drain() > 0) {  // memory idling in the pool: give it back and try once more
            (void)hipGetLastError();
            e = hipMalloc(&p, n);
        }
        if (e == hipSuccess) {
            bytes = n;
            pooled = pool;
        } else {
            p = nullptr;
        }
        return e;
    }
    hipError_t alloc_pooled(size_t n) { return alloc(n); }
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
// how many: see host_threads() in k2r_capi_encode.hip (K2R_HOST_THREADS overrides)
int host_thread_count();

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
