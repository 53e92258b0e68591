#!/usr/bin/env python3
"""Throughput of the superchunk assembly (dcdf_superchunk_build = Superchunk::build, superchunk.rs:88-270, the caller of the chunk
path): one time segment of the bench.py raster, [32, 4096, 4096], device-resident, levels [4, 8] = 16 x 16 sub-chunks of 256^2.
Everything `Variable::append` needs for one chunk_size slice is inside the timed call: per-tile (min, max), fractional bits (float
input), 256 Chunk::builds in one launch, the min / max Dacs, SHA-256 of every stored object, the objects copied to the host."""
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
    ap.add_argument("--instants", type=int, default=32)
    ap.add_argument("--extent", type=int, default=4096)
    ap.add_argument("--dtype", choices=["i32", "f32"], default="i32")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    from dcdf_amd import _lib as L
    from dcdf_amd.encoder import DeviceBuffer, synth_fill
    T, E = args.instants, args.extent
    buf = DeviceBuffer(T * E * E * 4)
    S = 256
    # the synthetic raster is generated tile by tile into a dense [T, E, E] array: fill a staging tile, scatter its rows
    # (dcdf_synth_fill writes dense tiles) -- simpler: generate the whole plane range in one call
    synth_fill(buf.ptr, L.DCDF_I32, 0xDCDF0003, 0, T, 0, E, 0, E)
    fb, dtype = 0, L.DCDF_I32
    if args.dtype == "f32":  # the same integers as float32 multiples of 1/8
        import torch
        t = torch.empty(0)  # noqa: F841  (torch only to convert on the device)
        raise SystemExit("float32 variant: use tests (test_gpu_superchunk.py); the throughput figure is quoted for int32")
    d = L.TileDesc()
    d.base, d.dtype = buf.ptr, dtype
    d.stride_t, d.stride_r, d.stride_c = E * E, E, 1
    d.instants, d.rows, d.cols = T, E, E
    d.fractional_bits, d.round = fb, 0
    levels = (C.c_uint32 * 2)(int(np.log2(E)) - 8, 8)
    best, last = None, None
    for _ in range(args.reps):
        out = C.POINTER(L.SuperchunkBuild)()
        t0 = time.perf_counter()
        L.check(L.lib().dcdf_superchunk_build(C.byref(d), levels, C.c_size_t(2), 2, L.MEM_DEVICE, C.byref(out)), "superchunk_build")
        dt = time.perf_counter() - t0
        s = out.contents
        last = {"objects": int(s.n_objects), "size": int(s.size), "elided": int(s.elided), "external": int(s.external),
                "snapshots": int(s.snapshots), "logs": int(s.logs), "stored_bytes": int(sum(s.objects[i].len for i in range(s.n_objects)))}
        L.lib().dcdf_free_superchunk(out)
        best = dt if best is None else min(best, dt)
    cells = T * E * E
    print(json.dumps({"entry": "dcdf_superchunk_build (device-resident [%d,%d,%d] int32, levels [%d,8])" % (T, E, E, levels[0]),
                      "seconds": best, "cells_per_s": cells / best, "input_GB_per_s": cells * 4 / best / 1e9, **last}))


if __name__ == "__main__":
    main()
