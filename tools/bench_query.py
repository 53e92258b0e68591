#!/usr/bin/env python3
"""BASELINE configs[4]: random get_window (fill_window) + search_window queries against the encoded 4096x4096x365
raster on one GPU (decode-path throughput).  The raster is encoded on the device (same workload as
`bench.py --workload config2`), the encoded chunks are opened through the C ABI (dcdf_chunk_open), and batches of
chunk-level queries -- random chunk, random window inside it -- go through dcdf_query_fill_window_batch /
dcdf_query_search_batch.  A sample of the answers is checked against the synthetic model (brute force)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=1000000)
    ap.add_argument("--batch", type=int, default=250000)
    ap.add_argument("--segments", type=int, default=12, help="time segments of the raster to encode (12 = all 365 instants)")
    ap.add_argument("--max-t", type=int, default=8)
    ap.add_argument("--max-side", type=int, default=16)
    args = ap.parse_args()
    import torch
    import dcdf_amd as dc
    from dcdf_amd import _lib as L, synth
    from dcdf_amd.encoder import Encoder, synth_fill

    S = 256
    grid = [(seg, i, j) for seg in range(args.segments) for i in range(16) for j in range(16)]
    sizes = [(min(365, 32 * seg + 32) - 32 * seg) * S * S for seg, _, _ in grid]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    flat = torch.empty((int(offs[-1]),), dtype=torch.int32, device="cuda")
    descs = []
    for (seg, i, j), o, z in zip(grid, offs, sizes):
        t0, t1 = 32 * seg, min(365, 32 * seg + 32)
        v = flat[int(o):int(o) + z].view(t1 - t0, S, S)
        synth_fill(v.data_ptr(), L.DCDF_I32, 0xDCDF0003, t0, t1, S * i, S * i + S, S * j, S * j + S)
        descs.append((v.data_ptr(), L.DCDF_I32, (S * S, S, 1), (t1 - t0, S, S)))
    torch.cuda.synchronize()
    enc = Encoder(descs, k=2)
    enc.run()
    t0 = time.perf_counter()
    chunks = [dc.Chunk(enc.fetch(c)) for c in range(len(grid))]
    open_s = time.perf_counter() - t0
    enc.close()
    del flat
    torch.cuda.empty_cache()

    rng = np.random.default_rng(5)
    nq = args.queries
    res = {"chunks": len(chunks), "open_seconds": open_s, "queries": nq}
    fw_ms = fw_wall = se_ms = se_wall = 0.0
    cells = hits = 0
    checked = 0
    for b0 in range(0, nq, args.batch):
        n = min(args.batch, nq - b0)
        ci = rng.integers(0, len(chunks), n)
        T = np.array([chunks[c].shape()[0] for c in ci])
        s = (rng.random(n) * T).astype(np.int64)
        e = np.minimum(T, s + rng.integers(1, args.max_t + 1, n))
        t = rng.integers(0, S, n)
        bo = np.minimum(S, t + rng.integers(1, args.max_side + 1, n))
        l = rng.integers(0, S, n)
        r = np.minimum(S, l + rng.integers(1, args.max_side + 1, n))
        cub = np.stack([s, e, t, bo, l, r], axis=1).astype(np.uint32)
        cubes = cub.ctypes.data_as(C.POINTER(L.Cube))
        vol = ((e - s) * (bo - t) * (r - l)).astype(np.uint64)
        woff = np.concatenate([[0], np.cumsum(vol)[:-1]]).astype(np.uint64)
        total = int(vol.sum())
        handles = (C.c_void_p * n)(*[chunks[c]._h for c in ci])
        out = np.empty(total, dtype=np.int64)
        ms = C.c_float()
        w0 = time.perf_counter()
        L.check(L.lib().dcdf_query_fill_window_batch(handles, cubes, C.c_size_t(n), C.c_void_p(out.ctypes.data),
                                                     C.c_void_p(woff.ctypes.data), C.byref(ms)), "fill_window_batch")
        fw_wall += time.perf_counter() - w0
        fw_ms += ms.value
        cells += total
        # value band of about a tenth of the raster's range around a random level
        lower = rng.integers(-300, 300, n).astype(np.int64)
        upper = lower + 60
        counts = np.zeros(n, dtype=np.uint64)
        soff = np.zeros(n, dtype=np.uint64)
        trip = np.empty((total, 3), dtype=np.uint32)
        w0 = time.perf_counter()
        L.check(L.lib().dcdf_query_search_batch(handles, cubes, C.c_void_p(lower.ctypes.data), C.c_void_p(upper.ctypes.data),
                                                C.c_size_t(n), C.c_void_p(trip.ctypes.data), C.c_size_t(total),
                                                C.c_void_p(counts.ctypes.data), C.c_void_p(soff.ctypes.data), C.byref(ms)),
                "search_batch")
        se_wall += time.perf_counter() - w0
        se_ms += ms.value
        hits += int(counts.sum())
        for q in rng.integers(0, n, 25):  # spot check against the synthetic model
            seg, i, j = grid[ci[q]]
            ref = synth.cells(0xDCDF0003, 32 * seg + int(s[q]), 32 * seg + int(e[q]), S * i + int(t[q]), S * i + int(bo[q]),
                              S * j + int(l[q]), S * j + int(r[q]), np.int32)
            got = out[int(woff[q]):int(woff[q]) + int(vol[q])].reshape(ref.shape)
            assert (got == ref).all(), "fill_window mismatch"
            tr = trip[int(soff[q]):int(soff[q]) + int(counts[q])]
            m = (ref >= lower[q]) & (ref <= upper[q])
            exp = np.argwhere(m) + np.array([int(s[q]), int(t[q]), int(l[q])])
            assert sorted(map(tuple, tr.tolist())) == sorted(map(tuple, exp.tolist())), "search mismatch"
            checked += 1
    res.update({"fill_window": {"queries_per_s_kernel": nq / (fw_ms * 1e-3), "cells_per_s_kernel": cells / (fw_ms * 1e-3),
                                "queries_per_s_end_to_end": nq / fw_wall, "kernel_ms": fw_ms, "cells": cells},
                "search_window": {"queries_per_s_kernel": nq / (se_ms * 1e-3), "queries_per_s_end_to_end": nq / se_wall,
                                  "kernel_ms": se_ms, "hits": hits},
                "answers_checked_vs_model": checked})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
