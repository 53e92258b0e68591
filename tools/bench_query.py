#!/usr/bin/env python3
"""BASELINE configs[4]: 1M random get_window (fill_window) + search_window queries against the encoded 4096x4096x365
raster on one GPU -- decode-path throughput, measured like bench.py measures the encoder.

The raster is encoded on the device (the workload of `bench.py`), the encoded chunks are opened WHERE THEY LIE
(dcdf_chunk_open_batch over the session's device buffers: parsed on the device, every side-16 table in one launch),
dataset-level cubes are split into chunk-level sub-queries at tile / segment boundaries exactly where
Superchunk::subchunks_for (superchunk.rs:589-633) and Span::fill_window (span.rs:190-216) split them, and go through
dcdf_query_fill_window_batch_typed (int32 values of int32 chunks, written by the kernel into a device-resident result)
and dcdf_query_search_batch_mem.  Answers are checked against the synthetic model (brute force).

Prints ONE JSON line: metric = decoded cells/s and queries/s in-kernel; `roofline` = algorithmic bytes (the result bytes
plus the window's area share of the touched instants' Logs and of their blocks' Snapshots; SURVEY 8(d)'s whole-structure
upper bound is listed beside it) over the kernels' HIP-event time; `cpu_baseline` = the CPU oracle's Chunk::fill_window /
iter_search on a bounded sample of the SAME sub-queries, timed inside the library, 1 thread and all host cores;
`end_to_end` = host routing + calls, device-resident and host-resident results."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0
SEED = 0xDCDF0003


def encode_raster(segments, extent=4096):
    """The bench.py workload (first `segments` time segments), encoded on the device.  Returns (encoder, grid, T)."""
    from dcdf_amd import _lib as L
    from dcdf_amd.encoder import DeviceBuffer, Encoder, synth_fill
    S, nt = 256, extent // 256
    grid = [(seg, i, j) for seg in range(segments) for i in range(nt) for j in range(nt)]
    sizes = [(min(365, 32 * seg + 32) - 32 * seg) * S * S for seg, _, _ in grid]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    flat = DeviceBuffer(int(offs[-1]) * 4)  # (through the C ABI: no torch needed, and none initialised under pytest)
    descs = []
    for (seg, i, j), o, z in zip(grid, offs, sizes):
        t0, t1 = 32 * seg, min(365, 32 * seg + 32)
        ptr = flat.ptr + int(o) * 4
        synth_fill(ptr, L.DCDF_I32, SEED, t0, t1, S * i, S * i + S, S * j, S * j + S)
        descs.append((ptr, L.DCDF_I32, (S * S, S, 1), (t1 - t0, S, S)))
    enc = Encoder(descs, k=2)
    enc.run()
    return enc, grid, min(365, 32 * segments), flat


def make_queries(rng, n, TT, extent, nt):
    """n dataset-level cubes and their chunk-level sub-queries (query, chunk, local cube)."""
    t0 = rng.integers(0, TT, n)
    t1 = np.minimum(TT, t0 + rng.integers(1, 9, n))
    r0 = rng.integers(0, extent, n)
    r1 = np.minimum(extent, r0 + rng.integers(1, 65, n))
    c0 = rng.integers(0, extent, n)
    c1 = np.minimum(extent, c0 + rng.integers(1, 65, n))
    subs = []
    qid = np.arange(n)
    for dt in (0, 1):
        for dr in (0, 1):
            for dc_ in (0, 1):
                seg, ti, tj = t0 // 32 + dt, r0 // 256 + dr, c0 // 256 + dc_
                a0, a1 = np.maximum(t0, seg * 32), np.minimum(t1, seg * 32 + 32)
                b0, b1 = np.maximum(r0, ti * 256), np.minimum(r1, ti * 256 + 256)
                d0, d1 = np.maximum(c0, tj * 256), np.minimum(c1, tj * 256 + 256)
                ok = (a1 > a0) & (b1 > b0) & (d1 > d0)
                cid = ((seg * nt + ti) * nt + tj)[ok]
                subs.append(np.stack([qid[ok], cid, (a0 - seg * 32)[ok], (a1 - seg * 32)[ok], (b0 - ti * 256)[ok],
                                      (b1 - ti * 256)[ok], (d0 - tj * 256)[ok], (d1 - tj * 256)[ok]], axis=1))
    sub = np.concatenate(subs)
    sub = sub[np.argsort(sub[:, 0], kind="stable")]
    return (t0, t1, r0, r1, c0, c1), sub


def query_source_sha():
    """Identifies the decode kernels' sources a committed PMC profile belongs to (roofline.traffic is only quoted for them)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("k2r_common.h", "k2r_decode.h", "k2r_query.hip"):
        h.update(f.encode())
        h.update(open(os.path.join(ROOT, "dcdf_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def run(queries=1000000, batch=250000, segments=12, extent=4096, check=20, cpu_sample=4000, host_results=True, verbose=False,
        session=None, raster_level=True):
    """session = (encoder, instants): query the chunks an existing encoder session (bench.py's, already run) left on the device
    instead of encoding the raster here; the session stays open."""
    import dcdf_amd as dc
    from dcdf_amd import _lib as L, synth
    from dcdf_amd.encoder import DeviceBuffer
    nt = extent // 256
    if session is None:
        enc, grid, TT, raster = encode_raster(segments, extent)
    else:
        enc, TT = session
        raster = None
    t0 = time.perf_counter()
    chunks = enc.open_chunks()  # (returns when the device is done: the call synchronises)
    open_s = time.perf_counter() - t0
    if raster is not None:
        raster.free()  # the queries run against the encoded chunks alone
    from dcdf_amd.raster import EncodedRaster
    ER = EncodedRaster((TT, extent, extent), chunks)
    nat_f_wall = nat_s_wall = nat_f_ms = nat_s_ms = 0.0
    # per-instant byte ranges of every chunk (host metadata) -> the touched encoded bytes of a sub-query
    ioff, isnap, base = [], [], [0]
    for c in chunks:
        T = c.shape()[0]
        off = np.zeros(T + 1, dtype=np.uint64)
        sn = np.zeros(T, dtype=np.uint32)
        L.check(L.lib().dcdf_chunk_instant_layout(c._h, C.c_void_p(off.ctypes.data), C.c_void_p(sn.ctypes.data)))
        ioff.append(off.astype(np.int64))
        isnap.append(sn.astype(np.int64))
        base.append(base[-1] + T)
    base = np.array(base[:-1], dtype=np.int64)
    inst_bytes = np.concatenate([np.diff(o) for o in ioff])                           # bytes of (chunk, instant)
    snap_bytes = np.concatenate([np.diff(o)[s] * (s != np.arange(len(s))) for o, s in zip(ioff, isnap)])  # its block's snapshot (0 for a snapshot itself)
    touch = np.concatenate([[0], np.cumsum(inst_bytes + snap_bytes)])                 # prefix over global instants

    rng = np.random.default_rng(0xDCDF0005)
    smp = synth.cells(SEED, 0, TT, 1000, 1064, 2000, 2064, np.int32).ravel()
    edges = np.percentile(smp, np.arange(0, 101, 10)).astype(np.int64)
    PCT = np.stack([edges[:-1], edges[1:]], axis=1)  # ten 10-percentile-wide bands of the value range
    nq = queries
    fw_ms = fw_wall = fw_wall_host = se_ms = se_wall = 0.0
    cells = hits = nsub_f = nsub_s = 0
    enc_touched_f = enc_touched_s = 0
    enc_share_f = enc_share_s = 0.0
    checked = 0
    cpu_f, cpu_s = [], []  # bounded samples of sub-queries for the CPU leg: (chunk id, cube[, lower, upper])

    def brute(q, spec):
        a = [int(x[q]) for x in spec]
        return synth.cells(SEED, a[0], a[1], a[2], a[3], a[4], a[5], np.int32), (a[0], a[2], a[4])

    for b0 in range(0, nq, batch):
        n = min(batch, nq - b0)
        half = n // 2
        # ---- fill_window half: int32 values, decoded straight into a device-resident result ----
        w0 = time.perf_counter()
        spec, sub = make_queries(rng, half, TT, extent, nt)
        m = len(sub)
        cub = np.ascontiguousarray(sub[:, 2:8].astype(np.uint32))
        vol = ((sub[:, 3] - sub[:, 2]) * (sub[:, 5] - sub[:, 4]) * (sub[:, 7] - sub[:, 6])).astype(np.uint64)
        woff = np.concatenate([[0], np.cumsum(vol)[:-1]]).astype(np.uint64)
        total = int(vol.sum())
        handles = (C.c_void_p * m)(*[chunks[c]._h for c in sub[:, 1]])
        dev = DeviceBuffer(max(4, total * 4))
        ms = C.c_float()
        L.check(L.lib().dcdf_query_fill_window_batch_typed(handles, cub.ctypes.data_as(C.POINTER(L.Cube)), C.c_size_t(m),
                                                           C.c_void_p(dev.ptr), L.DCDF_I32, L.MEM_DEVICE,
                                                           C.c_void_p(woff.ctypes.data), C.byref(ms)), "fill_window_batch_typed")
        fw_wall += time.perf_counter() - w0
        fw_ms += ms.value
        cells += total
        nsub_f += m
        g0 = base[sub[:, 1]] + sub[:, 2]
        tb = touch[g0 + (sub[:, 3] - sub[:, 2])] - touch[g0]
        enc_touched_f += int(tb.sum())
        enc_share_f += float((tb * ((sub[:, 5] - sub[:, 4]) * (sub[:, 7] - sub[:, 6])) / 65536.0).sum())
        if host_results:  # the same answers as a HOST array (typed: 4 bytes per cell cross PCIe)
            out_h = np.empty(total, dtype=np.int32)
            w0 = time.perf_counter()
            L.check(L.lib().dcdf_query_fill_window_batch_typed(handles, cub.ctypes.data_as(C.POINTER(L.Cube)), C.c_size_t(m),
                                                               C.c_void_p(out_h.ctypes.data), L.DCDF_I32, L.MEM_HOST,
                                                               C.c_void_p(woff.ctypes.data), None), "fill_window_batch_typed")
            fw_wall_host += time.perf_counter() - w0
        if check:
            out = dev.read(0, total * 4, np.int32)
            if host_results:
                assert (out == out_h).all()
            for q in rng.integers(0, half, check):  # reassemble the dataset-level window from its pieces
                ref, (t0_, r0, c0) = brute(q, spec)
                got = np.zeros_like(ref)
                for k in np.nonzero(sub[:, 0] == q)[0]:
                    _, cid, a0, a1, b0_, b1, d0, d1 = (int(x) for x in sub[k])
                    seg, ti, tj = cid // (nt * nt), (cid // nt) % nt, cid % nt
                    piece = out[int(woff[k]):int(woff[k]) + int(vol[k])].reshape(a1 - a0, b1 - b0_, d1 - d0)
                    got[seg * 32 + a0 - t0_:seg * 32 + a1 - t0_, ti * 256 + b0_ - r0:ti * 256 + b1 - r0,
                        tj * 256 + d0 - c0:tj * 256 + d1 - c0] = piece
                assert (got == ref).all(), "fill_window mismatch"
                checked += 1
        if len(cpu_f) < cpu_sample:
            cpu_f += [tuple(int(x) for x in sub[k, 1:8]) for k in range(min(m, cpu_sample - len(cpu_f)))]
        dev.free()
        # the same dataset-level cubes through dcdf_raster_fill_window_batch: the split into pieces happens in the library
        w0 = time.perf_counter()
        dcub = np.ascontiguousarray(np.stack(spec[:6], axis=1).astype(np.uint32)) if raster_level else None
        if raster_level:
            dvol = ((dcub[:, 1].astype(np.int64) - dcub[:, 0]) * (dcub[:, 3].astype(np.int64) - dcub[:, 2]) * (dcub[:, 5].astype(np.int64) - dcub[:, 4]))
            doff = np.concatenate([[0], np.cumsum(dvol)[:-1]]).astype(np.uint64)
            assert int(dvol.sum()) == total
            dev = DeviceBuffer(max(4, total * 4))
            nat_f_ms += ER.fill_windows_flat(dcub, dtype=np.int32, out_device_ptr=dev.ptr, out_offset=doff)
            nat_f_wall += time.perf_counter() - w0
        if check and raster_level:
            outn = dev.read(0, total * 4, np.int32)
            for q in rng.integers(0, half, check):
                ref, _ = brute(q, spec)
                assert (outn[int(doff[q]):int(doff[q]) + ref.size].reshape(ref.shape) == ref).all(), "raster fill_window mismatch"
                checked += 1
        if raster_level:
            dev.free()
        # ---- search_window half: [lower, upper] = a random 10-percentile-wide band of the value range ----
        w0 = time.perf_counter()
        spec, sub = make_queries(rng, n - half, TT, extent, nt)
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
        dtrip = DeviceBuffer(max(12, total * 12))
        L.check(L.lib().dcdf_query_search_batch_mem(handles, cub.ctypes.data_as(C.POINTER(L.Cube)), C.c_void_p(lower.ctypes.data),
                                                    C.c_void_p(upper.ctypes.data), C.c_size_t(m), C.c_void_p(dtrip.ptr),
                                                    C.c_size_t(total), L.MEM_DEVICE, C.c_void_p(counts.ctypes.data),
                                                    C.c_void_p(soff.ctypes.data), C.byref(ms)), "search_batch_mem")
        se_wall += time.perf_counter() - w0
        se_ms += ms.value
        hits += int(counts.sum())
        nsub_s += m
        g0 = base[sub[:, 1]] + sub[:, 2]
        tb = touch[g0 + (sub[:, 3] - sub[:, 2])] - touch[g0]
        enc_touched_s += int(tb.sum())
        enc_share_s += float((tb * ((sub[:, 5] - sub[:, 4]) * (sub[:, 7] - sub[:, 6])) / 65536.0).sum())
        if check:
            trip = dtrip.read(0, int(counts.sum()) * 12, np.uint32).reshape(-1, 3).astype(np.int64)
            for q in rng.integers(0, n - half, check):
                ref, (t0_, r0, c0) = brute(q, spec)
                want = set(map(tuple, (np.argwhere((ref >= qlo[q, 0]) & (ref <= qlo[q, 1])) + np.array([t0_, r0, c0])).tolist()))
                got = set()
                for k in np.nonzero(sub[:, 0] == q)[0]:
                    cid = int(sub[k, 1])
                    seg, ti, tj = cid // (nt * nt), (cid // nt) % nt, cid % nt
                    tr = trip[int(soff[k]):int(soff[k]) + int(counts[k])]
                    got |= set(map(tuple, (tr + np.array([seg * 32, ti * 256, tj * 256])).tolist()))
                assert got == want, "search mismatch"
                checked += 1
        if len(cpu_s) < cpu_sample:
            cpu_s += [tuple(int(x) for x in sub[k, 1:8]) + (int(lower[k]), int(upper[k])) for k in range(min(m, cpu_sample - len(cpu_s)))]
        dtrip.free()
        w0 = time.perf_counter()
        if raster_level:
            dcub = np.ascontiguousarray(np.stack(spec[:6], axis=1).astype(np.uint32))
            dtrip = DeviceBuffer(max(12, total * 12))
            _, noff, ncnt, ms_n = ER.search_flat(dcub, qlo[:, 0], qlo[:, 1], out_device_ptr=dtrip.ptr, cap=total)
            nat_s_ms += ms_n
            nat_s_wall += time.perf_counter() - w0
            assert int(ncnt.sum()) == int(counts.sum())
        if check and raster_level:
            trip = dtrip.read(0, int(ncnt.sum()) * 12, np.uint32).reshape(-1, 3).astype(np.int64)
            for q in rng.integers(0, n - half, check):
                ref, (t0_, r0, c0) = brute(q, spec)
                want = set(map(tuple, (np.argwhere((ref >= qlo[q, 0]) & (ref <= qlo[q, 1])) + np.array([t0_, r0, c0])).tolist()))
                assert set(map(tuple, trip[int(noff[q]):int(noff[q]) + int(ncnt[q])].tolist())) == want, "raster search mismatch"
                checked += 1
        if raster_level:
            dtrip.free()
        if verbose:
            print("batch %d done" % b0, file=sys.stderr)

    nqf, nqs = nq // 2, nq - nq // 2
    out_bytes_f, out_bytes_s = cells * 4, hits * 12
    alg_f, alg_s = enc_share_f + out_bytes_f, enc_share_s + out_bytes_s
    res = {
        "metric": "decode path: decoded cells/s (fill_window) and queries/s (fill_window + search_window), in-kernel",
        "value": cells / (fw_ms * 1e-3), "unit": "cells/s",
        "queries_per_s": nq / ((fw_ms + se_ms) * 1e-3),
        "config": {"workload": "configs[4]: %d random cubes (t0 U[0,%d), len_t U[1,8], h, w U[1,64]) against the encoded %dx%dx%d int32 "
                               "raster, half fill_window, half search_window on a 10-percentile band; %d chunks" % (nq, TT, extent, extent, TT, len(chunks)),
                   "chunks": len(chunks), "queries": nq, "answers_checked_vs_model": checked},
        "open": {"entry": "dcdf_chunk_open_batch over the encoder's device buffers", "chunks": len(chunks), "seconds": open_s},
        "fill_window": {"queries": nqf, "chunk_level_subqueries": nsub_f, "kernel_ms": fw_ms, "cells": cells,
                        "cells_per_s_kernel": cells / (fw_ms * 1e-3), "queries_per_s_kernel": nqf / (fw_ms * 1e-3),
                        "result": "int32, device-resident"},
        "search_window": {"queries": nqs, "chunk_level_subqueries": nsub_s, "kernel_ms": se_ms, "hits": hits,
                          "queries_per_s_kernel": nqs / (se_ms * 1e-3), "result": "(instant,row,col) uint32 triples, device-resident"},
        "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "achieved": (alg_f + alg_s) / ((fw_ms + se_ms) * 1e-3) / 1e9,
                     "frac": (alg_f + alg_s) / ((fw_ms + se_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic": None,
                     "fill_window": {"algorithmic_bytes": alg_f, "encoded_bytes_share": enc_share_f, "encoded_bytes_upper_bound": enc_touched_f, "output_bytes": out_bytes_f,
                                     "achieved": alg_f / (fw_ms * 1e-3) / 1e9, "frac": alg_f / (fw_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                     "search_window": {"algorithmic_bytes": alg_s, "encoded_bytes_share": enc_share_s, "encoded_bytes_upper_bound": enc_touched_s, "output_bytes": out_bytes_s,
                                       "achieved": alg_s / (se_ms * 1e-3) / 1e9, "frac": alg_s / (se_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                     "note": "algorithmic bytes = the result bytes + the window's share of the touched structures: (bytes of the touched "
                             "instant's Log + of its block's Snapshot) x (window cells / tile cells) per chunk-level sub-query and instant, i.e. "
                             "the encoded information about the cells asked for.  SURVEY 8(d)'s upper bound -- the WHOLE Log and Snapshot per "
                             "touched instant -- is listed as encoded_bytes_upper_bound; a walk reads far less than that (it would put the "
                             "figure above the HBM peak).  No target fraction (pointer chasing)"},
        "end_to_end": {"fill_window_queries_per_s_device_result": nqf / fw_wall,
                       "fill_window_queries_per_s_host_result_int32": (nqf / fw_wall_host) if host_results and fw_wall_host else None,
                       "search_queries_per_s_device_result": nqs / se_wall,
                       "note": "host routing (numpy) + handle arrays + the call; device result: nothing but counts crosses PCIe",
                       "raster_level": {"entry": "dcdf_raster_fill_window_batch / dcdf_raster_search_batch (dataset-level cubes; the split into "
                                                 "chunk-level pieces and the placement of every piece happen in the library)",
                                        "fill_window_queries_per_s_device_result": nqf / nat_f_wall if nat_f_wall else None, "fill_window_kernel_ms": nat_f_ms,
                                        "search_queries_per_s_device_result": nqs / nat_s_wall if nat_s_wall else None, "search_kernel_ms": nat_s_ms}},
    }
    # ---- CPU baseline: the oracle's Chunk::fill_window / iter_search on the sampled sub-queries --------------------------------
    if cpu_sample:
        import oracle_lib as O
        need = sorted(set(s[0] for s in cpu_f) | set(s[0] for s in cpu_s))
        och = {c: O.Chunk(chunks[c].write_to()) for c in need}
        nthr = max(1, min(16, len(os.sched_getaffinity(0))))
        native = O.native_lib() is not None  # the oracle rebuilt -O3 -march=native on this host (handles are opened by the portable build: same structs)
        native = False  # (handles belong to the library that opened them: time the build that opened the chunks)
        fch, fcub = [och[s[0]] for s in cpu_f], [s[1:7] for s in cpu_f]
        sch, scub = [och[s[0]] for s in cpu_s], [s[1:7] for s in cpu_s]
        slo, shi = [s[7] for s in cpu_s], [s[8] for s in cpu_s]

        def timed(chs, cub, lo, hi, threads):  # repeat the sample until ~2 s have been measured
            tot_s, tot_w, reps = 0.0, 0, 0
            while tot_s < 2.0 and reps < 200:
                sec, work = O.bench_queries(chs, cub, lo, hi, threads=threads, native=native)
                tot_s += sec
                tot_w += work
                reps += 1
            return tot_s, tot_w, reps

        tf1, wf1, rf1 = timed(fch, fcub, None, None, 1)
        tfn, wfn, rfn = timed(fch, fcub, None, None, nthr)
        ts1, ws1, rs1 = timed(sch, scub, slo, shi, 1)
        tsn, wsn, rsn = timed(sch, scub, slo, shi, nthr)
        res["cpu_baseline"] = {
            "kind": "port", "unit": "chunk-level sub-queries/s",
            "sample": "the first %d fill_window and %d search sub-queries of this workload, repeated until 2 s are measured, on the CPU "
                      "oracle (C++ restatement of Chunk::fill_window / iter_search, clock inside the library, portable -O3 build)" % (len(cpu_f), len(cpu_s)),
            "fill_window": {"cores": 1, "value": len(cpu_f) * rf1 / tf1, "cells_per_s": wf1 / tf1, "seconds": tf1,
                            "all_cores": {"cores": nthr, "value": len(cpu_f) * rfn / tfn, "cells_per_s": wfn / tfn, "seconds": tfn}},
            "search_window": {"cores": 1, "value": len(cpu_s) * rs1 / ts1, "hits_per_s": ws1 / ts1, "seconds": ts1,
                              "all_cores": {"cores": nthr, "value": len(cpu_s) * rsn / tsn, "hits_per_s": wsn / tsn, "seconds": tsn}},
            "gpu_subqueries_per_s_kernel": {"fill_window": nsub_f / (fw_ms * 1e-3), "search_window": nsub_s / (se_ms * 1e-3)},
        }
    # HBM bytes of the decode kernels come from separate rocprofv3 PMC passes of this tool (tools/collect_query_profiles.sh),
    # quoted only for the same query count and kernel sources
    tf = os.path.join(ROOT, "profiles", "query_traffic_latest.json")
    if os.path.exists(tf):
        try:
            rec = json.load(open(tf))
            if rec.get("source_sha") == query_source_sha() and rec.get("queries") == nq:
                res["roofline"]["traffic"] = rec.get("hbm_bytes_all_query_kernels")
                res["roofline"]["traffic_detail"] = {k: rec.get(k) for k in ("tag", "fetch_bytes_x2", "write_bytes", "per_kernel")}
        except Exception:
            pass
    res["roofline"]["source_sha"] = query_source_sha()
    ER.close()
    for c in chunks:
        c.close()
    if session is None:
        enc.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=1000000)
    ap.add_argument("--batch", type=int, default=250000)
    ap.add_argument("--segments", type=int, default=12, help="time segments of the raster to encode (12 = all 365 instants)")
    ap.add_argument("--cpu-sample", type=int, default=4000)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-raster-level", action="store_true", help="chunk-level entry points only (PMC passes: the kernels are the same)")
    args = ap.parse_args()
    res = run(args.queries, args.batch, args.segments, check=0 if args.no_check or os.environ.get("BENCH_QUERY_NO_CHECK") else 20,
              cpu_sample=args.cpu_sample, verbose=True, raster_level=not args.no_raster_level)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
