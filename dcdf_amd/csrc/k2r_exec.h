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

#ifdef K2R_PROFILE
    // Diagnostic builds: every WAVE keeps two running sums between stamps -- cycles it spent running its own code ("work") and
    // cycles it spent parked at workgroup barriers ("wait": the slowest wave of the phase sets it) -- read with s_memtime on
    // both sides of every barrier.  stamp(k) adds them to the wave's own row of sh.pw (lane 0 writes; no other wave touches the
    // row).  Per phase, max-over-waves of "work" is the critical path and mean-over-waves the average load; their difference
    // is what the phase loses to imbalance.
    uint64_t pw_last = 0, pw_work = 0, pw_wait = 0;
    __device__ __forceinline__ void pw_before() {
        const uint64_t t = __builtin_amdgcn_s_memtime();
        pw_work += t - pw_last;
        pw_last = t;
    }
    __device__ __forceinline__ void pw_after() {
        const uint64_t t = __builtin_amdgcn_s_memtime();
        pw_wait += t - pw_last;
        pw_last = t;
    }
    __device__ __forceinline__ void pw_start() { pw_last = __builtin_amdgcn_s_memtime(); pw_work = pw_wait = 0; }
#else
    __device__ __forceinline__ void pw_before() {}
    __device__ __forceinline__ void pw_after() {}
    __device__ __forceinline__ void pw_start() {}
#endif

    // Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits (vmcnt(0)) for every global
    // store of the wave to be acknowledged by L2 -- microseconds per phase here, for bytes nobody in the workgroup
    // reads back.  Phases that DO hand global data to other threads use barrier_global().
    __device__ __forceinline__ void lds_barrier() {
        pw_before();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        pw_after();
    }
    // Orders this wave's LDS stores before its later LDS loads ACROSS lanes (the LDS executes a wave's operations in order; the
    // compiler and the wait counter just have to keep them in program order): what one wave needs between two levels of a tree it
    // builds alone.
    __device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    __device__ __forceinline__ void barrier_global() {
        pw_before();
        __syncthreads();
        pw_after();
    }

    // Hand-off of global data to ANOTHER workgroup (the shared snapshot copy of a chunk encoded in parts, k2r_encode.h): the
    // producer's waves have drained their stores (barrier_global), one lane writes the XCD's L2 back and sets the flag;
    // the consumer polls the flag with one lane (agent scope), invalidates its CU's L1, and the workgroup joins at a barrier
    // before any plain load of the data (MI355X_MICROARCH.md, inter-workgroup visibility).  The poll is bounded: a producer
    // that never publishes (it cannot: every exit path does) would otherwise hang the queue.
    __device__ __forceinline__ void publish(uint32_t* flag, uint32_t value, uint32_t payload) {
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(flag + 1, payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __device__ __forceinline__ uint32_t await(const uint32_t* flag, uint32_t timeout_value) {
        if (tid == 0) {
            uint32_t v = 0;
            for (uint32_t spin = 0; spin < (1u << 22); spin++) {
                v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v != 0) break;
                __builtin_amdgcn_s_sleep(32);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            sh.err = v != 0 ? (int32_t)v : (int32_t)timeout_value;  // (sh.err is idle between instants; the caller resets it)
        }
        __syncthreads();
        const uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane(sh.err);
        __syncthreads();
        return v;
    }
    __device__ __forceinline__ uint32_t flag_payload(const uint32_t* flag) {
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }

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
        pw_before();  // the cycles since the last barrier are this wave's own work
        if ((tid & 63) == 0) {
            sh.pw[tid >> 6][k][0] += pw_work;
            sh.pw[tid >> 6][k][1] += pw_wait;
        }
        pw_work = pw_wait = 0;
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

    // The same for N independent values, step by step across all of them: a DPP instruction cannot issue right behind the
    // VALU instruction that produced its source (two wait states), so N interleaved chains run back to back where N
    // separate scans are half s_nop.
    template <int N>
    __device__ __forceinline__ static void wave_incl_scan_n(int (&x)[N]) {
#pragma unroll
        for (int i = 0; i < N; i++) x[i] += __builtin_amdgcn_update_dpp(0, x[i], 0x111, 0xf, 0xf, true);
#pragma unroll
        for (int i = 0; i < N; i++) x[i] += __builtin_amdgcn_update_dpp(0, x[i], 0x112, 0xf, 0xf, true);
#pragma unroll
        for (int i = 0; i < N; i++) x[i] += __builtin_amdgcn_update_dpp(0, x[i], 0x114, 0xf, 0xf, true);
#pragma unroll
        for (int i = 0; i < N; i++) x[i] += __builtin_amdgcn_update_dpp(0, x[i], 0x118, 0xf, 0xf, true);
#pragma unroll
        for (int i = 0; i < N; i++) x[i] += __builtin_amdgcn_update_dpp(0, x[i], 0x142, 0xa, 0xf, false);
#pragma unroll
        for (int i = 0; i < N; i++) x[i] += __builtin_amdgcn_update_dpp(0, x[i], 0x143, 0xc, 0xf, false);
    }
    // ... and the prefix sums inside each row of 16 lanes only (the first four steps)
    template <int N>
    __device__ __forceinline__ static void row_incl_scan_n(int (&x)[N]) {
#pragma unroll
        for (int i = 0; i < N; i++) x[i] += __builtin_amdgcn_update_dpp(0, x[i], 0x111, 0xf, 0xf, true);
#pragma unroll
        for (int i = 0; i < N; i++) x[i] += __builtin_amdgcn_update_dpp(0, x[i], 0x112, 0xf, 0xf, true);
#pragma unroll
        for (int i = 0; i < N; i++) x[i] += __builtin_amdgcn_update_dpp(0, x[i], 0x114, 0xf, 0xf, true);
#pragma unroll
        for (int i = 0; i < N; i++) x[i] += __builtin_amdgcn_update_dpp(0, x[i], 0x118, 0xf, 0xf, true);
    }

    // workgroup sum of r.sc[F0..F0+NF) -> sh.tot[F0..F0+NF) (no prefixes): wave totals through sh.wsum, then a 16-lane
    // row reduction of them (no atomics)
    template <int NF, int F0 = 0>
    __device__ __forceinline__ void reduce() {
        constexpr int W = NT < 64 ? NT : 64;
        constexpr int NW = NT < 64 ? 1 : NT / 64;
        const int lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        int inc[NF];
#pragma unroll
        for (int f = 0; f < NF; f++) inc[f] = (int)r.sc[F0 + f];
        wave_incl_scan_n(inc);
        if (lane == W - 1) {
#pragma unroll
            for (int f = 0; f < NF; f++) sh.wsum[wave][F0 + f] = (uint32_t)inc[f];
        }
        lds_barrier();
        int x[NF];
#pragma unroll
        for (int f = 0; f < NF; f++) x[f] = (lane < NW) ? (int)sh.wsum[lane < NW ? lane : 0][F0 + f] : 0;
        row_incl_scan_n(x);
        if (tid == NW - 1) {
#pragma unroll
            for (int f = 0; f < NF; f++) sh.tot[F0 + f] = (uint32_t)x[f];
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
        int inc[NF];
#pragma unroll
        for (int f = 0; f < NF; f++) inc[f] = (int)r.sc[f];
        wave_incl_scan_n(inc);
        if (lane == W - 1) {
#pragma unroll
            for (int f = 0; f < NF; f++) sh.wsum[wave][f] = (uint32_t)inc[f];
        }
#pragma unroll
        for (int f = 0; f < NF; f++) r.sc[f] = (uint32_t)inc[f] - r.sc[f];
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
        int x[NF];
#pragma unroll
        for (int f = 0; f < NF; f++) x[f] = (lane < NW) ? (int)sh.wsum[lane < NW ? lane : 0][f] : 0;
        row_incl_scan_n(x);
#pragma unroll
        for (int f = 0; f < NF; f++) {
            tot[f] = (uint32_t)__builtin_amdgcn_readlane(x[f], NW - 1);
            r.sc[f] += wave == 0 ? 0u : (uint32_t)__builtin_amdgcn_readlane(x[f], wave - 1);
        }
    }


    // Lanes 4q .. 4q+3 of a wave form a quad (DPP quad_perm: no LDS, full rate).  quad_perm<CTRL>: lane i of a quad reads lane
    // (CTRL >> 2i) & 3 of the same quad; the reductions leave the result in all four lanes.
    template <int CTRL>
    __device__ __forceinline__ static uint32_t quad_perm(uint32_t v) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
    }
    template <int K>
    __device__ __forceinline__ static uint32_t quad_bcast(uint32_t v) { return quad_perm<K * 0x55>(v); }
    __device__ __forceinline__ static int32_t quad_min(int32_t x) {
        int32_t y = (int32_t)quad_perm<0xB1>((uint32_t)x);
        x = x < y ? x : y;
        y = (int32_t)quad_perm<0x4E>((uint32_t)x);
        return x < y ? x : y;
    }
    __device__ __forceinline__ static int32_t quad_max(int32_t x) {
        int32_t y = (int32_t)quad_perm<0xB1>((uint32_t)x);
        x = x > y ? x : y;
        y = (int32_t)quad_perm<0x4E>((uint32_t)x);
        return x > y ? x : y;
    }
    __device__ __forceinline__ static uint32_t quad_or(uint32_t x) {
        x |= quad_perm<0xB1>(x);
        return x | quad_perm<0x4E>(x);
    }
    // rows of 16 lanes: row_ror<N> rotates a row by N lanes (DPP); the reductions leave the result in all sixteen lanes and expect
    // the value to be uniform over each quad already
    template <int N>
    __device__ __forceinline__ static uint32_t row_ror(uint32_t v) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + N, 0xf, 0xf, true);
    }
    __device__ __forceinline__ static int32_t row_min_of_quads(int32_t x) {
        int32_t y = (int32_t)row_ror<4>((uint32_t)x);
        x = x < y ? x : y;
        y = (int32_t)row_ror<8>((uint32_t)x);
        return x < y ? x : y;
    }
    __device__ __forceinline__ static int32_t row_max_of_quads(int32_t x) {
        int32_t y = (int32_t)row_ror<4>((uint32_t)x);
        x = x > y ? x : y;
        y = (int32_t)row_ror<8>((uint32_t)x);
        return x > y ? x : y;
    }
    __device__ __forceinline__ static uint32_t row_or_of_quads(uint32_t x) {
        x |= row_ror<4>(x);
        return x | row_ror<8>(x);
    }
    // ... and the four rows of a wave, for values uniform over each row: scalar results
    __device__ __forceinline__ static int32_t wave_min_of_rows(int32_t x) {
        const int32_t a = __builtin_amdgcn_readlane(x, 0), b = __builtin_amdgcn_readlane(x, 16), c = __builtin_amdgcn_readlane(x, 32),
                      d = __builtin_amdgcn_readlane(x, 48);
        const int32_t p = a < b ? a : b, q = c < d ? c : d;
        return p < q ? p : q;
    }
    __device__ __forceinline__ static int32_t wave_max_of_rows(int32_t x) {
        const int32_t a = __builtin_amdgcn_readlane(x, 0), b = __builtin_amdgcn_readlane(x, 16), c = __builtin_amdgcn_readlane(x, 32),
                      d = __builtin_amdgcn_readlane(x, 48);
        const int32_t p = a > b ? a : b, q = c > d ? c : d;
        return p > q ? p : q;
    }
    __device__ __forceinline__ static uint32_t wave_or_of_rows(uint32_t x) {
        return (uint32_t)(__builtin_amdgcn_readlane((int)x, 0) | __builtin_amdgcn_readlane((int)x, 16) | __builtin_amdgcn_readlane((int)x, 32) |
                          __builtin_amdgcn_readlane((int)x, 48));
    }
    // the value v of lane `src` (0..63) of this wave (ds_bpermute: the LDS crossbar, no LDS memory)
    __device__ __forceinline__ static uint32_t lane_pull(uint32_t src, uint32_t v) {
        return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v);
    }

    // Slots for the stash records of ONE wave in ONE returning LDS atomic per counter: `c` = this lane's demand on two
    // counters, packed (low half: cntA, high half: cntB; a wave asks for far less than 2^16 of either).  A wave scan gives
    // every lane its offset, lanes 0 and 1 add the wave's totals to the two counters in the same instruction, and the lane's
    // first slots come back in a / b.  (The per-lane form -- one atomic per record inside a divergent branch -- cost five
    // dependent LDS round trips per sub-block of phase 1.)  Must be called by every lane of the wave.
    // enda / endb: one past the last slot of the whole wave (wave-uniform: "does any of this wave's records overflow?").
    __device__ __forceinline__ void stash_alloc(uint32_t c, uint32_t* cntA, uint32_t* cntB, uint32_t& a, uint32_t& b, uint32_t& enda,
                                                uint32_t& endb) {
        constexpr int W = NT < 64 ? NT : 64;
        const int lane = tid & 63;
        const uint32_t incl = wave_incl_scan(c);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, W - 1);
        uint32_t ret = 0;
        if (lane < 2) ret = atomicAdd(lane == 0 ? cntA : cntB, lane == 0 ? (tot & 0xffffu) : (tot >> 16));
        const uint32_t ba = (uint32_t)__builtin_amdgcn_readlane((int)ret, 0), bb = (uint32_t)__builtin_amdgcn_readlane((int)ret, W > 1 ? 1 : 0);
        const uint32_t excl = incl - c;
        a = ba + (excl & 0xffffu);
        b = bb + (excl >> 16);
        enda = ba + (tot & 0xffffu);
        endb = bb + (tot >> 16);
    }

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
        int incl[NF];
#pragma unroll
        for (int f = 0; f < NF; f++) incl[f] = (int)r.sc[f];
        wave_incl_scan_n(incl);
        if (lane == W - 1) {
#pragma unroll
            for (int f = 0; f < NF; f++) sh.wsum[wave][f] = (uint32_t)incl[f];
        }
        lds_barrier();
        // the NW (<= 16) wave totals of a field sit in lanes 0..NW-1 of every wave: a 4-step row scan turns them into
        // the wave bases -- one 16-lane LDS read per field instead of NW broadcast reads
        int x[NF];
#pragma unroll
        for (int f = 0; f < NF; f++) x[f] = (lane < NW) ? (int)sh.wsum[lane < NW ? lane : 0][f] : 0;
        row_incl_scan_n(x);
#pragma unroll
        for (int f = 0; f < NF; f++) {
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane(x[f], NW - 1);
            const uint32_t base = wave == 0 ? 0u : (uint32_t)__builtin_amdgcn_readlane(x[f], wave - 1);
            r.sc[f] = base + (uint32_t)incl[f] - r.sc[f];
            if (TOT && tid == 0) sh.tot[f] = tot;
        }
        if (TOT) lds_barrier();
    }
};

#endif  // __HIPCC__

#if !defined(__HIP_DEVICE_COMPILE__)
}  // namespace k2r
#include <vector>
namespace k2r {

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
    void wave_lds_fence() {}
    void barrier_global() {}
    void publish(uint32_t* flag, uint32_t value, uint32_t payload) {
        flag[1] = payload;
        flag[0] = value;
    }
    uint32_t await(const uint32_t* flag, uint32_t timeout_value) { return flag[0] != 0 ? flag[0] : timeout_value; }  // (items run in queue order)
    uint32_t flag_payload(const uint32_t* flag) { return flag[1]; }
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


    void stash_alloc(uint32_t c, uint32_t* cntA, uint32_t* cntB, uint32_t& a, uint32_t& b, uint32_t& enda, uint32_t& endb) {
        a = *cntA;  // (threads run one after another: every thread is a "wave" of its own)
        b = *cntB;
        *cntA += c & 0xffffu;
        *cntB += c >> 16;
        enda = *cntA;
        endb = *cntB;
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
