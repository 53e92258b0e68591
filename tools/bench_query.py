#!/usr/bin/env python3
"""BASELINE configs[4]: random get_window (fill_window) + search_window queries against the encoded 4096x4096x365
raster on one GPU (decode-path throughput), as specified in SURVEY.md 8(d) config 5.  The raster is encoded on
the device (same workload as `bench.py --workload config2`), the encoded chunks are opened through the C ABI
(dcdf_chunk_open), dataset-level cubes are split into chunk-level sub-queries at tile/segment boundaries and go
through dcdf_query_fill_window_batch / dcdf_query_search_batch.  A sample of the answers is reassembled and checked
against the synthetic model (brute force)."""
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

    # dataset-level cubes (t0 in U[0,365), len_t in U[1,8], h, w in U[1,64], origin uniform,
    # clipped), half fill_window and half search_window, each split at tile (256) and time-segment (32) boundaries
    # into chunk-level sub-queries exactly where Superchunk::subchunks_for / Span::fill_window would split them.
    rng = np.random.default_rng(0xDCDF0005)
    smp = synth.cells(0xDCDF0003, 0, 365, 1000, 1064, 2000, 2064, np.int32).ravel()
    edges = np.percentile(smp, np.arange(0, 101, 10)).astype(np.int64)
    PCT = np.stack([edges[:-1], edges[1:]], axis=1)  # ten 10-percentile-wide bands of the value range
    nq = args.queries
    TT = min(365, 32 * args.segments)
    res = {"chunks": len(chunks), "open_seconds": open_s, "queries": nq}
    fw_ms = fw_wall = se_ms = se_wall = 0.0
    cells = hits = nsub_f = nsub_s = 0
    checked = 0
    chunk_T = np.array([c.shape()[0] for c in chunks])

    def make(n):
        t0 = rng.integers(0, TT, n)
        t1 = np.minimum(TT, t0 + rng.integers(1, 9, n))
        r0 = rng.integers(0, 4096, n)
        r1 = np.minimum(4096, r0 + rng.integers(1, 65, n))
        c0 = rng.integers(0, 4096, n)
        c1 = np.minimum(4096, c0 + rng.integers(1, 65, n))
        subs = []  # (query id, chunk id, local cube)
        qid = np.arange(n)
        for dt in (0, 1):
            for dr in (0, 1):
                for dc in (0, 1):
                    seg, ti, tj = t0 // 32 + dt, r0 // 256 + dr, c0 // 256 + dc
                    a0, a1 = np.maximum(t0, seg * 32), np.minimum(t1, seg * 32 + 32)
                    b0, b1 = np.maximum(r0, ti * 256), np.minimum(r1, ti * 256 + 256)
                    d0, d1 = np.maximum(c0, tj * 256), np.minimum(c1, tj * 256 + 256)
                    ok = (a1 > a0) & (b1 > b0) & (d1 > d0)
                    cid = (seg * 256 + ti * 16 + tj)[ok]
                    subs.append(np.stack([qid[ok], cid, (a0 - seg * 32)[ok], (a1 - seg * 32)[ok], (b0 - ti * 256)[ok],
                                          (b1 - ti * 256)[ok], (d0 - tj * 256)[ok], (d1 - tj * 256)[ok]], axis=1))
        sub = np.concatenate(subs)
        sub = sub[np.argsort(sub[:, 0], kind="stable")]
        return (t0, t1, r0, r1, c0, c1), sub

    def brute(q, spec):
        t0, t1, r0, r1, c0, c1 = (int(x[q]) for x in spec)
        return synth.cells(0xDCDF0003, t0, t1, r0, r1, c0, c1, np.int32), (t0, r0, c0)

    for b0 in range(0, nq, args.batch):
        n = min(args.batch, nq - b0)
        half = n // 2
        # ---- fill_window half
        spec, sub = make(half)
        m = len(sub)
        cub = np.ascontiguousarray(sub[:, 2:8].astype(np.uint32))
        vol = ((sub[:, 3] - sub[:, 2]) * (sub[:, 5] - sub[:, 4]) * (sub[:, 7] - sub[:, 6])).astype(np.uint64)
        woff = np.concatenate([[0], np.cumsum(vol)[:-1]]).astype(np.uint64)
        total = int(vol.sum())
        handles = (C.c_void_p * m)(*[chunks[c]._h for c in sub[:, 1]])
        out = np.empty(total, dtype=np.int64)
        ms = C.c_float()
        w0 = time.perf_counter()
        L.check(L.lib().dcdf_query_fill_window_batch(handles, cub.ctypes.data_as(C.POINTER(L.Cube)), C.c_size_t(m),
                                                     C.c_void_p(out.ctypes.data), C.c_void_p(woff.ctypes.data), C.byref(ms)),
                "fill_window_batch")
        fw_wall += time.perf_counter() - w0
        fw_ms += ms.value
        cells += total
        nsub_f += m
        for q in ([] if os.environ.get("BENCH_QUERY_NO_CHECK") else rng.integers(0, half, 20)):  # spot check: reassemble the dataset-level window from its pieces
            ref, (t0, r0, c0) = brute(q, spec)
            got = np.zeros_like(ref, dtype=np.int64)
            for k in np.nonzero(sub[:, 0] == q)[0]:
                _, cid, a0, a1, b0_, b1, d0, d1 = (int(x) for x in sub[k])
                seg, ti, tj = cid // 256, (cid // 16) % 16, cid % 16
                piece = out[int(woff[k]):int(woff[k]) + int(vol[k])].reshape(a1 - a0, b1 - b0_, d1 - d0)
                got[seg * 32 + a0 - t0:seg * 32 + a1 - t0, ti * 256 + b0_ - r0:ti * 256 + b1 - r0,
                    tj * 256 + d0 - c0:tj * 256 + d1 - c0] = piece
            assert (got == ref).all(), "fill_window mismatch"
            checked += 1
        # ---- search_window half: [lower, upper] = a random 10-percentile-wide band of the value range
        spec, sub = make(n - half)
        m = len(sub)
        cub = np.ascontiguousarray(sub[:, 2:8].astype(np.uint32))
        vol = ((sub[:, 3] - sub[:, 2]) * (sub[:, 5] - sub[:, 4]) * (sub[:, 7] - sub[:, 6])).astype(np.uint64)
        total = int(vol.sum())
        handles = (C.c_void_p * m)(*[chunks[c]._h for c in sub[:, 1]])
        qlo = PCT[rng.integers(0, 10, n - half)]
        lower = np.ascontiguousarray(qlo[:, 0][sub[:, 0]]).astype(np.int64)
        upper = np.ascontiguousarray(qlo[:, 1][sub[:, 0]]).astype(np.int64)
        counts = np.zeros(m, dtype=np.uint64)
        soff = np.zeros(m, dtype=np.uint64)
        trip = np.empty((total, 3), dtype=np.uint32)
        w0 = time.perf_counter()
        L.check(L.lib().dcdf_query_search_batch(handles, cub.ctypes.data_as(C.POINTER(L.Cube)), C.c_void_p(lower.ctypes.data),
                                                C.c_void_p(upper.ctypes.data), C.c_size_t(m), C.c_void_p(trip.ctypes.data),
                                                C.c_size_t(total), C.c_void_p(counts.ctypes.data), C.c_void_p(soff.ctypes.data),
                                                C.byref(ms)), "search_batch")
        se_wall += time.perf_counter() - w0
        se_ms += ms.value
        hits += int(counts.sum())
        nsub_s += m
        for q in ([] if os.environ.get("BENCH_QUERY_NO_CHECK") else rng.integers(0, n - half, 20)):
            ref, (t0, r0, c0) = brute(q, spec)
            want = set(map(tuple, (np.argwhere((ref >= qlo[q, 0]) & (ref <= qlo[q, 1])) + np.array([t0, r0, c0])).tolist()))
            got = set()
            for k in np.nonzero(sub[:, 0] == q)[0]:
                cid = int(sub[k, 1])
                seg, ti, tj = cid // 256, (cid // 16) % 16, cid % 16
                tr = trip[int(soff[k]):int(soff[k]) + int(counts[k])].astype(np.int64)
                got |= set(map(tuple, (tr + np.array([seg * 32, ti * 256, tj * 256])).tolist()))
            assert got == want, "search mismatch"
            checked += 1
    nqf, nqs = nq // 2, nq - nq // 2
    res.update({"fill_window": {"queries": nqf, "chunk_level_subqueries": nsub_f, "queries_per_s_kernel": nqf / (fw_ms * 1e-3),
                                "cells_per_s_kernel": cells / (fw_ms * 1e-3), "queries_per_s_end_to_end": nqf / fw_wall,
                                "kernel_ms": fw_ms, "cells": cells},
                "search_window": {"queries": nqs, "chunk_level_subqueries": nsub_s, "queries_per_s_kernel": nqs / (se_ms * 1e-3),
                                  "queries_per_s_end_to_end": nqs / se_wall, "kernel_ms": se_ms, "hits": hits},
                "answers_checked_vs_model": checked})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
