#!/usr/bin/env python3
"""Ingest through the caller path: `Dataset.append` (dataset.rs:834-878 -> Superchunk::build, superchunk.rs:88-270) of the
4096 x 4096 x 365 int32 raster of BASELINE configs[2] from HOST numpy arrays, one `chunk_size = 32` slice per call -- staging, H2D,
per-tile (min, max), 256 Chunk::builds, min / max Dacs, download of the stored objects, their CIDs, span tree bookkeeping.
Beside it: the same slice through `dcdf_superchunk_build` with the input already in HBM (what the device-resident caller pays), and
the H2D rate of this box for the same arrays (the ceiling of a host-fed ingest).  Prints ONE JSON line."""
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
    ap.add_argument("--days", type=int, default=365)
    ap.add_argument("--extent", type=int, default=4096)
    ap.add_argument("--chunk-size", type=int, default=32)
    args = ap.parse_args()
    import dcdf_amd as dc
    from dcdf_amd import _lib as L
    from dcdf_amd import dataset as D
    from dcdf_amd.encoder import DeviceBuffer, synth_fill
    E, T, CS = args.extent, args.days, args.chunk_size
    # the raster in host memory (generated on the device slice by slice, copied out)
    host = np.empty((T, E, E), dtype=np.int32)
    dev = DeviceBuffer(CS * E * E * 4)
    for t0 in range(0, T, CS):
        t1 = min(T, t0 + CS)
        synth_fill(dev.ptr, L.DCDF_I32, 0xDCDF0003, t0, t1, 0, E, 0, E)
        host[t0:t1] = dev.read(0, (t1 - t0) * E * E * 4, np.int32).reshape(t1 - t0, E, E)
    # the H2D rate of these very arrays (pageable numpy memory -> HBM), best of 3 on one slice
    nb = CS * E * E * 4
    h2d, h2d_all = None, []
    for _ in range(3):
        w0 = time.perf_counter()
        L.check(L.lib().dcdf_device_copy(C.c_void_p(dev.ptr), C.c_void_p(host[:CS].ctypes.data), C.c_size_t(nb), 1), "device_copy")
        dt = time.perf_counter() - w0
        h2d_all.append(dt)
        h2d = dt if h2d is None else min(h2d, dt)
    # (a) device-resident slice through dcdf_superchunk_build
    synth_fill(dev.ptr, L.DCDF_I32, 0xDCDF0003, 0, CS, 0, E, 0, E)
    d = L.TileDesc()
    d.base, d.dtype = dev.ptr, L.DCDF_I32
    d.stride_t, d.stride_r, d.stride_c = E * E, E, 1
    d.instants, d.rows, d.cols = CS, E, E
    d.fractional_bits, d.round = 0, 0
    lv = [int(np.log2(E)) - 8, 8]
    levels = (C.c_uint32 * 2)(*lv)
    best_dev, stored = None, 0
    for _ in range(3):
        out = C.POINTER(L.SuperchunkBuild)()
        w0 = time.perf_counter()
        L.check(L.lib().dcdf_superchunk_build(C.byref(d), levels, C.c_size_t(2), 2, L.MEM_DEVICE, C.byref(out)), "superchunk_build")
        dt = time.perf_counter() - w0
        stored = int(sum(out.contents.objects[i].len for i in range(out.contents.n_objects)))
        L.lib().dcdf_free_superchunk(out)
        best_dev = dt if best_dev is None else min(best_dev, dt)
    dev.free()
    # (b) Dataset.append from host numpy, slice after slice
    res = D.Resolver()
    ds = D.Dataset.new([D.Coordinate.time("t", 0, 86400), D.Coordinate.range("y", 0, 1, E, np.float64), D.Coordinate.range("x", 0, 1, E, np.float64)],
                       (E, E), res)
    ds = ds.add_variable("v", span_size=16, chunk_size=CS, k2_levels=lv, dtype=np.int32)
    per = []
    w_all = time.perf_counter()
    for t0 in range(0, T, CS):
        w0 = time.perf_counter()
        ds = ds.append("v", host[t0:min(T, t0 + CS)])
        per.append(time.perf_counter() - w0)
    total = time.perf_counter() - w_all
    cid = ds.commit()
    # spot check: a window of the stored dataset equals the input
    v = ds.v
    got = v[40:43, 1000:1040, 2000:2040].data
    assert (got == host[40:43, 1000:1040, 2000:2040]).all(), "stored dataset differs from the input"
    cells = T * E * E
    full = [p for p, t0 in zip(per, range(0, T, CS)) if t0 + CS <= T]
    print(json.dumps({
        "entry": "Dataset.append('v', host numpy [%d,%d,%d] int32) x %d (chunk_size %d, k2_levels %s)" % (CS, E, E, len(per), CS, lv),
        "cells": cells, "seconds": total, "cells_per_s": cells / total, "input_GB_per_s": cells * 4 / total / 1e9,
        "seconds_per_full_slice": {"min": min(full), "median": sorted(full)[len(full) // 2], "max": max(full)},
        "h2d_of_one_slice": {"bytes": nb, "seconds": h2d, "GB_per_s": nb / h2d / 1e9, "each_of_3": h2d_all, "what": "hipMemcpy of the same pageable numpy slice, best of 3"},
        "ingest_over_h2d_rate": (cells * 4 / total) / (nb / h2d),
        "device_resident_slice": {"entry": "dcdf_superchunk_build, [%d,%d,%d] int32 already in HBM, objects + CIDs returned in host memory" % (CS, E, E),
                                  "seconds": best_dev, "cells_per_s": CS * E * E / best_dev, "stored_bytes": stored},
        "dataset_cid": cid.hex(), "objects_in_store": len(res.objects)}))


if __name__ == "__main__":
    main()
