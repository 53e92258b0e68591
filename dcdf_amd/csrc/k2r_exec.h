// k2r_exec.h -- execution contexts for the bulk-synchronous kernel bodies.
//
// A kernel body is a sequence of phases.  `ex.par(f)` runs f(tid, regs) for every
// thread of the workgroup and ends with a workgroup barrier; code between phases is
// wave-uniform and may only READ shared state.  `ex.scan<NF>()` is a workgroup
// exclusive prefix sum over the per-thread fields regs.sc[0..NF) that also leaves
// the totals in shared.tot[].
//
//   GpuExec : the gfx950 context (one HIP thread per logical thread; wave64 shuffles + LDS).
//   SimExec : sequential host context, compiled ONLY by tests/sim (see k2r_common.h).
#pragma once
#include "k2r_common.h"

namespace k2r {

constexpr int MAX_SCAN_FIELDS = 16;  // 32-bit words per thread that scan<>/reduce<> can combine

#if defined(__HIPCC__)

template <class SH, class TR, int NT>
struct GpuExec {
    SH& sh;
    TR r;
    const int tid;
    uint32_t gfail = 0;  // guard failure bits of this thread (k2r_encode.h guard_pos)
    static constexpr int kNT = NT;
    static constexpr bool kSim = false;

    __device__ GpuExec(SH& s) : sh(s), tid((int)threadIdx.x) {}

    // Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits (vmcnt(0)) for every global
    // store of the wave to be acknowledged by L2 -- microseconds per phase here, for bytes nobody in the workgroup
    // reads back.  Phases that DO hand global data to other threads use barrier_global().
    __device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
    __device__ __forceinline__ void barrier_global() { __syncthreads(); }

    template <class F>
    __device__ __forceinline__ void par(F&& f) {
        f(tid, r);
        lds_barrier();
    }
    // phase without trailing barrier (caller guarantees no hazard before the next barrier)
    template <class F>
    __device__ __forceinline__ void par_nosync(F&& f) {
        f(tid, r);
    }
    __device__ __forceinline__ void barrier() { lds_barrier(); }
    // diagnostic builds only: attribute the cycles since the previous stamp to phase k (thread 0's view)
    __device__ __forceinline__ void stamp(int k) {
#ifdef K2R_PROFILE
        if (tid == 0) {
            const uint64_t t = clock64();
            sh.prof[k] += t - sh.prof_last;
            sh.prof_last = t;
        }
#else
        (void)k;
#endif
    }

    // Values read from LDS are wave-uniform by construction here but the compiler cannot know: pin them into
    // SGPRs so that the (large) amount of layout arithmetic derived from them stays off the vector registers.
    __device__ __forceinline__ uint32_t uni(uint32_t v) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
    __device__ __forceinline__ int32_t uni(int32_t v) const { return __builtin_amdgcn_readfirstlane(v); }
    __device__ __forceinline__ uint64_t uni(uint64_t v) const {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
        return ((uint64_t)hi << 32) | lo;
    }

    // Inclusive prefix sum over the 64 lanes of a wave with DPP moves (row_shr 1/2/4/8 inside each row of 16, then
    // row_bcast:15 / row_bcast:31 to carry row totals): ten VALU instructions and no LDS traffic, where a
    // __shfl_up ladder costs six dependent ds_bpermute round trips.  Inactive lanes contribute 0.
    __device__ __forceinline__ static uint32_t wave_incl_scan(uint32_t v) {
        int x = (int)v;
        x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);   // row_shr:1
        x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);   // row_shr:2
        x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);   // row_shr:4
        x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);   // row_shr:8
        x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
        x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
        return (uint32_t)x;
    }

    // workgroup sum of r.sc[F0..F0+NF) -> sh.tot[F0..F0+NF) (no prefixes): wave totals through sh.wsum, then a 16-lane
    // row reduction of them (no atomics)
    template <int NF, int F0 = 0>
    __device__ __forceinline__ void reduce() {
        constexpr int W = NT < 64 ? NT : 64;
        constexpr int NW = NT < 64 ? 1 : NT / 64;
        const int lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int f = F0; f < F0 + NF; f++) {
            const uint32_t inc = wave_incl_scan(r.sc[f]);
            if (lane == W - 1) sh.wsum[wave][f] = inc;
        }
        lds_barrier();
#pragma unroll
        for (int f = F0; f < F0 + NF; f++) {
            int x = (lane < NW) ? (int)sh.wsum[lane < NW ? lane : 0][f] : 0;
            x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);  // row_shr:1
            x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);  // row_shr:2
            x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);  // row_shr:4
            x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);  // row_shr:8
            if (tid == NW - 1) sh.tot[f] = (uint32_t)x;
        }
        lds_barrier();
    }

    // Split form of scan<>, for a scan whose totals are consumed by one thread and whose prefixes are needed one
    // phase later: scan_begin turns r.sc[f] into wave-local exclusive prefixes and leaves the wave totals in sh.wsum
    // (one barrier); scan_finish -- called by every thread at the top of the next phase -- adds the wave bases and
    // returns the workgroup totals (in every thread).  No round trip through sh.tot, no second barrier.
    template <int NF>
    __device__ __forceinline__ void scan_begin() {
        constexpr int W = NT < 64 ? NT : 64;
        const int lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int f = 0; f < NF; f++) {
            const uint32_t inc = wave_incl_scan(r.sc[f]);
            if (lane == W - 1) sh.wsum[wave][f] = inc;
            r.sc[f] = inc - r.sc[f];
        }
        lds_barrier();
    }
    static constexpr int planner() { return 0; }
    // the value `v` of lane `lane` (a compile-time-foldable constant) of this wave, in every lane
    __device__ __forceinline__ uint32_t lane_value(uint32_t v, int lane) const { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
    template <int NF>
    __device__ __forceinline__ void scan_finish(uint32_t (&tot)[NF]) {
        constexpr int NW = NT < 64 ? 1 : NT / 64;
        const int lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int f = 0; f < NF; f++) {
            int x = (lane < NW) ? (int)sh.wsum[lane < NW ? lane : 0][f] : 0;
            x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);  // row_shr:1
            x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);  // row_shr:2
            x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);  // row_shr:4
            x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);  // row_shr:8
            tot[f] = (uint32_t)__builtin_amdgcn_readlane(x, NW - 1);
            r.sc[f] += wave == 0 ? 0u : (uint32_t)__builtin_amdgcn_readlane(x, wave - 1);
        }
    }


    // ---- SIMT section: the body is ordinary per-lane code that uses the wave primitives below.  All 64 lanes of a
    // wave reach every primitive together (no divergence around them).  The sequential context runs such a body on
    // fibers that rendezvous at the primitives (tests/sim), so the very same source is checked on a CPU. ----
    template <class F>
    __device__ __forceinline__ void simt(F&& f) {
        f(tid);
    }
    __device__ __forceinline__ void sync(int) { lds_barrier(); }
    __device__ __forceinline__ void w_fence(int) { __builtin_amdgcn_wave_barrier(); }  // orders this wave's LDS traffic for the compiler
    __device__ __forceinline__ uint64_t w_ballot(int, bool p) const { return __builtin_amdgcn_ballot_w64(p); }
    // set bits of m below this lane
    __device__ __forceinline__ uint32_t w_mbcnt(int, uint64_t m) const {
        return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    }
    __device__ __forceinline__ int32_t w_lane(int, int32_t v, int lane) const { return __builtin_amdgcn_readlane(v, lane); }
    __device__ __forceinline__ int32_t w_first(int, int32_t v) const { return __builtin_amdgcn_readfirstlane(v); }
    // value of the previous lane; the first lane of each row of 16 gets its own value back (DPP row_shr:1)
    __device__ __forceinline__ int32_t w_prev(int, int32_t v) const { return __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false); }
    // min / max over aligned groups of G = 4 or 16 lanes, result in every lane of the group (quad_perm swaps, then
    // row_half_mirror / row_mirror): 2 or 4 DPP-fused VALU operations, no LDS traffic
    template <int G>
    __device__ __forceinline__ int32_t w_gmin(int, int32_t x) const {
        int32_t y = __builtin_amdgcn_mov_dpp(x, 0xB1, 0xf, 0xf, true);  // quad_perm:[1,0,3,2]
        x = y < x ? y : x;
        y = __builtin_amdgcn_mov_dpp(x, 0x4E, 0xf, 0xf, true);          // quad_perm:[2,3,0,1]
        x = y < x ? y : x;
        if (G == 16) x = w_gmin16_from4(0, x);
        return x;
    }
    template <int G>
    __device__ __forceinline__ int32_t w_gmax(int, int32_t x) const {
        int32_t y = __builtin_amdgcn_mov_dpp(x, 0xB1, 0xf, 0xf, true);
        x = y > x ? y : x;
        y = __builtin_amdgcn_mov_dpp(x, 0x4E, 0xf, 0xf, true);
        x = y > x ? y : x;
        if (G == 16) x = w_gmax16_from4(0, x);
        return x;
    }
    // the same over rows of 16, given values already reduced over quads
    __device__ __forceinline__ int32_t w_gmin16_from4(int, int32_t x) const {
        int32_t y = __builtin_amdgcn_mov_dpp(x, 0x141, 0xf, 0xf, true);  // row_half_mirror
        x = y < x ? y : x;
        y = __builtin_amdgcn_mov_dpp(x, 0x140, 0xf, 0xf, true);          // row_mirror
        return y < x ? y : x;
    }
    __device__ __forceinline__ int32_t w_gmax16_from4(int, int32_t x) const {
        int32_t y = __builtin_amdgcn_mov_dpp(x, 0x141, 0xf, 0xf, true);
        x = y > x ? y : x;
        y = __builtin_amdgcn_mov_dpp(x, 0x140, 0xf, 0xf, true);
        return y > x ? y : x;
    }
    // inclusive prefix sum inside each row of 16 lanes (row_shr 1, 2, 4, 8 with zero fill)
    __device__ __forceinline__ uint32_t w_rowscan(int, uint32_t v) const {
        int x = (int)v;
        x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
        x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
        x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
        x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
        return (uint32_t)x;
    }
    __device__ __forceinline__ uint32_t w_iscan(int, uint32_t v) const { return wave_incl_scan(v); }
    __device__ __forceinline__ uint32_t lds_or_nr(uint32_t* p, uint32_t v) { atomicOr(p, v); return 0; }
    __device__ __forceinline__ uint32_t lds_or(uint32_t* p, uint32_t v) { return atomicOr(p, v); }
    __device__ __forceinline__ uint32_t lds_add(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
    __device__ __forceinline__ int32_t lds_min(int32_t* p, int32_t v) { return atomicMin(p, v); }

    // workgroup exclusive scan of r.sc[0..NF); totals -> sh.tot[0..NF) unless TOT is false, in which case the prefixes
    // (registers) are the only result and the closing barrier is saved
    template <int NF, bool TOT = true>
    __device__ __forceinline__ void scan() {
        constexpr int W = NT < 64 ? NT : 64;    // active lanes per wave
        constexpr int NW = NT < 64 ? 1 : NT / 64;
        const int lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        uint32_t incl[NF];
#pragma unroll
        for (int f = 0; f < NF; f++) {
            incl[f] = wave_incl_scan(r.sc[f]);
            if (lane == W - 1) sh.wsum[wave][f] = incl[f];
        }
        lds_barrier();
        // the NW (<= 16) wave totals of a field sit in lanes 0..NW-1 of every wave: a 4-step row scan turns them into
        // the wave bases -- one 16-lane LDS read per field instead of NW broadcast reads
#pragma unroll
        for (int f = 0; f < NF; f++) {
            int x = (lane < NW) ? (int)sh.wsum[lane < NW ? lane : 0][f] : 0;
            x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);  // row_shr:1
            x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);  // row_shr:2
            x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);  // row_shr:4
            x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);  // row_shr:8
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane(x, NW - 1);
            const uint32_t base = wave == 0 ? 0u : (uint32_t)__builtin_amdgcn_readlane(x, wave - 1);
            r.sc[f] = base + incl[f] - r.sc[f];
            if (TOT && tid == 0) sh.tot[f] = tot;
        }
        if (TOT) lds_barrier();
    }
};

#endif  // __HIPCC__

#if !defined(__HIP_DEVICE_COMPILE__)
}  // namespace k2r
#include <ucontext.h>

#include <functional>
#include <memory>
#include <vector>
namespace k2r {

// Cooperative SIMT emulation for SimExec::simt: one fiber per logical thread; fibers of a wave rendezvous at the wave
// primitives (every lane deposits a value, then all of them read the 64 values), all fibers at sync().
struct SimtSched {
    int n = 0, cur = 0, lanes = 64;
    std::vector<ucontext_t> ctx;
    ucontext_t main_ctx;
    std::vector<std::unique_ptr<char[]>> stacks;
    std::vector<char> done;
    std::function<void(int)> body;
    struct WaveSlot {
        uint64_t buf[2][64];
        int arrived = 0;
        uint32_t gen = 0;
    };
    std::vector<WaveSlot> waves;
    int bar_arrived = 0;
    uint32_t bar_gen = 0;
    static SimtSched*& current() {
        static thread_local SimtSched* s = nullptr;
        return s;
    }
    static void tramp() {
        SimtSched* s = current();
        const int id = s->cur;
        s->body(id);
        s->done[id] = 1;
    }
    void yield() { swapcontext(&ctx[cur], &main_ctx); }
    void run(int nthreads, std::function<void(int)> f) {
        constexpr size_t kStack = 256 * 1024;
        n = nthreads;
        lanes = n < 64 ? n : 64;
        body = std::move(f);
        ctx.resize(n);
        done.assign(n, 0);
        stacks.resize(n);
        waves.assign((n + 63) / 64, WaveSlot());
        bar_arrived = 0;
        SimtSched* prev = current();
        current() = this;
        for (int i = 0; i < n; i++) {
            if (!stacks[i]) stacks[i].reset(new char[kStack]);
            getcontext(&ctx[i]);
            ctx[i].uc_stack.ss_sp = stacks[i].get();
            ctx[i].uc_stack.ss_size = kStack;
            ctx[i].uc_link = &main_ctx;
            makecontext(&ctx[i], (void (*)())tramp, 0);
        }
        int remaining = n;
        while (remaining > 0) {
            remaining = 0;
            for (int i = 0; i < n; i++) {
                if (done[i]) continue;
                cur = i;
                swapcontext(&main_ctx, &ctx[i]);
                if (!done[i]) remaining++;
            }
        }
        current() = prev;
    }
    const uint64_t* gather(int tid, uint64_t v) {
        WaveSlot& w = waves[tid >> 6];
        const uint32_t g = w.gen;
        w.buf[g & 1][tid & 63] = v;
        if (++w.arrived == lanes) {
            w.arrived = 0;
            w.gen++;
        } else {
            while (w.gen == g) yield();
        }
        return w.buf[g & 1];
    }
    void sync_all() {
        const uint32_t g = bar_gen;
        if (++bar_arrived == n) {
            bar_arrived = 0;
            bar_gen++;
        } else {
            while (bar_gen == g) yield();
        }
    }
};

template <class SH, class TR, int NT>
struct SimExec {
    SH& sh;
    std::vector<TR> regs;
    uint32_t gfail = 0;
    static constexpr int kNT = NT;
    static constexpr bool kSim = true;

    explicit SimExec(SH& s) : sh(s), regs(NT) {}

    template <class F>
    void par(F&& f) {
        for (int t = 0; t < NT; t++) f(t, regs[t]);
    }
    template <class F>
    void par_nosync(F&& f) {
        for (int t = 0; t < NT; t++) f(t, regs[t]);
    }
    void barrier() {}
    void barrier_global() {}
    void stamp(int) {}
    template <class T>
    T uni(T v) const { return v; }
    template <int NF, int F0 = 0>
    void reduce() {
        for (int f = F0; f < F0 + NF; f++) {
            uint32_t run = 0;
            for (int t = 0; t < NT; t++) run += regs[t].sc[f];
            sh.tot[f] = run;
        }
    }

    template <int NF>
    void scan_begin() {
        scan<NF, true>();
    }
    static constexpr int planner() { return 0; }
    uint32_t lane_value(uint32_t v, int) const { return v; }  // (never reached: the sequential context plans on one thread)
    template <int NF>
    void scan_finish(uint32_t (&t)[NF]) {
        for (int f = 0; f < NF; f++) t[f] = sh.tot[f];
    }


    // ---- SIMT section (see GpuExec): fibers + rendezvous ----
    SimtSched sched;
    template <class F>
    void simt(F&& f) {
        sched.run(NT, [&](int t) { f(t); });
    }
    void sync(int) { sched.sync_all(); }
    void w_fence(int tid) { sched.gather(tid, 0); }
    uint64_t w_ballot(int tid, bool p) {
        const uint64_t* a = sched.gather(tid, p ? 1 : 0);
        uint64_t m = 0;
        for (int i = 0; i < sched.lanes; i++) m |= (a[i] & 1) << i;
        return m;
    }
    uint32_t w_mbcnt(int tid, uint64_t m) const { return (uint32_t)__builtin_popcountll(m & (((uint64_t)1 << (tid & 63)) - 1)); }
    int32_t w_lane(int tid, int32_t v, int lane) { return (int32_t)(uint32_t)sched.gather(tid, (uint32_t)v)[lane]; }
    int32_t w_first(int tid, int32_t v) { return w_lane(tid, v, 0); }
    int32_t w_prev(int tid, int32_t v) {
        const uint64_t* a = sched.gather(tid, (uint32_t)v);
        const int l = tid & 63;
        return (l & 15) == 0 ? v : (int32_t)(uint32_t)a[l - 1];
    }
    template <int G>
    int32_t w_gmin(int tid, int32_t v) {
        const uint64_t* a = sched.gather(tid, (uint32_t)v);
        const int l0 = (tid & 63) & ~(G - 1);
        int32_t r = (int32_t)(uint32_t)a[l0];
        for (int i = 1; i < G; i++) r = (int32_t)(uint32_t)a[l0 + i] < r ? (int32_t)(uint32_t)a[l0 + i] : r;
        return r;
    }
    template <int G>
    int32_t w_gmax(int tid, int32_t v) {
        const uint64_t* a = sched.gather(tid, (uint32_t)v);
        const int l0 = (tid & 63) & ~(G - 1);
        int32_t r = (int32_t)(uint32_t)a[l0];
        for (int i = 1; i < G; i++) r = (int32_t)(uint32_t)a[l0 + i] > r ? (int32_t)(uint32_t)a[l0 + i] : r;
        return r;
    }
    int32_t w_gmin16_from4(int tid, int32_t v) { return w_gmin<16>(tid, v); }
    int32_t w_gmax16_from4(int tid, int32_t v) { return w_gmax<16>(tid, v); }
    uint32_t w_rowscan(int tid, uint32_t v) {
        const uint64_t* a = sched.gather(tid, v);
        uint32_t r = 0;
        for (int i = (tid & 63) & ~15; i <= (tid & 63); i++) r += (uint32_t)a[i];
        return r;
    }
    uint32_t w_iscan(int tid, uint32_t v) {
        const uint64_t* a = sched.gather(tid, v);
        uint32_t r = 0;
        for (int i = 0; i <= (tid & 63); i++) r += (uint32_t)a[i];
        return r;
    }
    uint32_t lds_or_nr(uint32_t* p, uint32_t v) {
        *p |= v;
        return 0;
    }
    uint32_t lds_or(uint32_t* p, uint32_t v) {
        uint32_t o = *p;
        *p = o | v;
        return o;
    }
    uint32_t lds_add(uint32_t* p, uint32_t v) {
        uint32_t o = *p;
        *p = o + v;
        return o;
    }
    int32_t lds_min(int32_t* p, int32_t v) {
        int32_t o = *p;
        if (v < o) *p = v;
        return o;
    }

    template <int NF, bool TOT = true>
    void scan() {
        for (int f = 0; f < NF; f++) {
            uint32_t run = 0;
            for (int t = 0; t < NT; t++) {
                const uint32_t v = regs[t].sc[f];
                regs[t].sc[f] = run;
                run += v;
            }
            sh.tot[f] = run;
        }
    }
};
#endif

}  // namespace k2r
