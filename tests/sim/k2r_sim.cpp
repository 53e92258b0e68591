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

// > 0: encode in speculative parts and splice (mirrors k_stitch): two parts [0, m) + [m, T), or g_parts parts of decreasing
// length as k2r_capi_encode.hip's part_bounds cuts them
static uint32_t g_split_at = 0, g_parts = 0;
extern "C" void sim_set_split(uint32_t m) { g_split_at = m; g_parts = 0; }
extern "C" void sim_set_parts(uint32_t p) { g_parts = p; g_split_at = 0; }

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
    std::vector<uint32_t> at;  // where the parts after the first begin
    if (g_split_at > 0 && g_split_at < instants) at.push_back(g_split_at);
    if (g_parts >= 2) {
        uint32_t pos = 0, left = instants;
        while (at.size() + 2 <= g_parts && left >= 4) {
            pos += (left + 1) / 2;
            left -= (left + 1) / 2;
            at.push_back(pos);
        }
    }
    if (!at.empty()) {
        // the parts in queue order (the first, then its continuations), then the splice of k_stitch on the host
        std::vector<uint32_t> flag(4, 0), shared_cmp((size_t)S * S / 2 + 64, 0xA5A5A5A5u);
        const size_t np = at.size() + 1;
        std::vector<TileArgs> pa(np, ta);
        std::vector<TileResult> pr(np);
        std::vector<std::vector<uint8_t>> pout(np);
        for (size_t p = 0; p < np; p++) {
            pa[p].inst_begin = p == 0 ? 0 : at[p - 1];
            pa[p].inst_end = p + 1 < np ? at[p] : instants;
            pa[p].shared_flag = flag.data();
            pa[p].shared_cmp = shared_cmp.data();
            if (p > 0) {
                pout[p].assign(cap, 0);
                pa[p].out = pout[p].data();
            }
            pr[p] = TileResult{};
            run_lg(pa[p], &pr[p]);
        }
        res = pr[0];
        int32_t st = ST_OK;
        uint64_t total = res.len;
        if (res.status != ST_OK) st = res.status;
        else if (res.snapshots != 1) st = ST_RESPLIT;
        for (size_t p = 1; p < np && st == ST_OK; p++) {
            if (pr[p].status != ST_OK) st = pr[p].status;
            else if (p + 1 < np && pr[p].snapshots != 0) st = ST_RESPLIT;
            total += pr[p].len;
        }
        if (st == ST_OK && total > cap) st = ST_OUT_CAPACITY;
        if (st == ST_OK) {
            uint64_t o = res.len;
            for (size_t p = 1; p < np; p++) {
                std::memcpy(out + o, pout[p].data(), pr[p].len);
                o += pr[p].len;
                res.snapshots += pr[p].snapshots;
                res.logs += pr[p].logs;
                res.stash_logs += pr[p].stash_logs;
            }
            out[6] = (uint8_t)pr[np - 1].carry_count;
            store_be32(out + 2, 1u + pr[np - 1].snapshots);
            res.len = total;
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
