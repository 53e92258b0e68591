// k2r_sim.cpp -- TEST INFRASTRUCTURE ONLY.
// Compiles the kernel bodies of dcdf_amd/csrc (the exact source hipcc builds for gfx950) with g++ and a
// sequential execution context, so kernel logic can be checked on a GPU-less machine (and under
// ASan/UBSan) before it runs on the card.  Not linked into libdcdf_k2r.so; the product cannot reach it.
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "../../dcdf_amd/csrc/k2r_encode.h"

using namespace k2r;

template <int LOG2S, bool PADDED, int VEC>
static void run(const TileArgs& ta, TileResult* res) {
    using C = EncCfg<LOG2S>;
    auto sh = std::make_unique<EncShared<C>>();
    std::memset(sh.get(), 0xA5, sizeof(*sh));  // poison: the kernel must initialise what it reads
    SimExec<EncShared<C>, EncRegs, C::NT> ex(*sh);
    std::memset((void*)ex.regs.data(), 0x5A, ex.regs.size() * sizeof(EncRegs));  // registers start as garbage on a GPU
    std::vector<uint64_t> listV(C::MAXV + 1), listM(C::MAXT + 1);
    std::vector<uint32_t> scmp(32 * C::NT, 0xA5A5A5A5u);
    std::vector<uint32_t> ovf((size_t)(5 * 4 + 3 * 16) * C::NBLK, 0xA5A5A5A5u);
    encode_chunk<C, PADDED, VEC>(ex, ta, res, listV.data(), listM.data(), scmp.data(), ovf.data());
}

template <int LOG2S>
static void run_l(const TileArgs& ta, TileResult* res, bool padded, int vec) {
    if (padded) run<LOG2S, true, 0>(ta, res);
    else if (vec == 1) run<LOG2S, false, 1>(ta, res);
    else if (vec == 2) run<LOG2S, false, 2>(ta, res);
    else if (vec == 3) run<LOG2S, false, 3>(ta, res);
    else if (vec == 4) run<LOG2S, false, 4>(ta, res);
    else run<LOG2S, false, 0>(ta, res);
}

static uint32_t g_split_at = 0;  // > 0: encode as two speculative halves [0, m) + [m, T) and splice (mirrors k_stitch)
extern "C" void sim_set_split(uint32_t m) { g_split_at = m; }

static uint32_t g_last_stash_logs = 0;
extern "C" uint32_t sim_last_stash_logs() { return g_last_stash_logs; }

extern "C" int sim_encode(const void* base, int dtype, int64_t st, int64_t sr, int64_t sc, uint32_t instants,
                          uint32_t rows, uint32_t cols, int fbits, int round, uint8_t* out, uint64_t cap,
                          int64_t* minmax, int force_novec, int32_t* status, uint32_t* snapshots, uint32_t* logs,
                          uint64_t* len) {
    TileArgs ta{};
    ta.base = base; ta.st = st; ta.sr = sr; ta.sc = sc;
    ta.instants = instants; ta.rows = rows; ta.cols = cols;
    ta.dtype = dtype; ta.fbits = (dtype == ENC_F32 || dtype == ENC_F64) ? (uint32_t)fbits : 0; ta.round = (uint32_t)round;
    ta.out = out; ta.out_cap = cap; ta.minmax = minmax;
    if (const char* sw = std::getenv("K2R_SIM_STASH_WORDS")) ta.stash_words = (uint32_t)std::atoi(sw);  // force the fallback passes
    uint32_t m = rows > cols ? rows : cols;
    int lg = 0;
    while ((1u << lg) < m) lg++;
    if (lg < 3 || lg > 8 || instants == 0) return -8;
    const uint32_t S = 1u << lg;
    const bool padded = rows != S || cols != S;
    const uint64_t esz = (dtype == ENC_I64 || dtype == ENC_F64) ? 8 : 4;
    const int64_t al = (int64_t)(16 / esz);
    const bool rows16 = !force_novec && !padded && sc == 1 && (sr % al) == 0 && sr > 0 && (st % al) == 0 && ((uintptr_t)base % 16) == 0 &&
                        ((uint64_t)(rows - 1) * (uint64_t)sr + cols) * esz < (1ull << 31);
    const int vec = !rows16 ? 0 : (dtype == ENC_I32 ? 1 : (dtype == ENC_F32 ? 2 : (dtype == ENC_I64 ? 3 : 4)));
    TileResult res{};
    auto run_lg = [&](const TileArgs& a, TileResult* r) {
        switch (lg) {
            case 3: run_l<3>(a, r, padded, vec); break;
            case 4: run_l<4>(a, r, padded, vec); break;
            case 5: run_l<5>(a, r, padded, vec); break;
            case 6: run_l<6>(a, r, padded, vec); break;
            case 7: run_l<7>(a, r, padded, vec); break;
            case 8: run_l<8>(a, r, padded, vec); break;
        }
    };
    if (g_split_at > 0 && g_split_at < instants) {
        // the splice of k2r_capi_encode.hip's k_stitch, on the host
        TileArgs a = ta, b = ta;
        a.inst_end = g_split_at;
        b.inst_begin = g_split_at;
        b.inst_end = instants;
        std::vector<uint8_t> outb(cap);
        b.out = outb.data();
        TileResult rb{};
        run_lg(a, &res);
        run_lg(b, &rb);
        int32_t st = ST_OK;
        if (res.status != ST_OK) st = res.status;
        else if (res.snapshots != 1) st = ST_RESPLIT;
        else if (rb.status != ST_OK) st = rb.status;
        else if (res.len + rb.len > cap) st = ST_OUT_CAPACITY;
        if (st == ST_OK) {
            std::memcpy(out + res.len, outb.data(), rb.len);
            out[6] = (uint8_t)rb.carry_count;
            store_be32(out + 2, 1u + rb.snapshots);
            res.len += rb.len;
            res.snapshots += rb.snapshots;
            res.logs += rb.logs;
            res.stash_logs += rb.stash_logs;
        } else {
            res.status = st;
            res.len = 0;
        }
    } else {
        run_lg(ta, &res);
    }
    g_last_stash_logs = res.stash_logs;
    *status = res.status; *snapshots = res.snapshots; *logs = res.logs; *len = res.len;
    return 0;
}
