#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (dcdf_chunk_build_batch): numpy tiles in host memory in,
encoded bytes in host memory out -- staging, H2D copy, kernel, D2H copy all inside the timed region.
DESIGN.md quotes this next to bench.py's HBM-resident number; it is never bench.py's `value`."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=128)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import dcdf_amd as dc
    from dcdf_amd import synth
    T, S = 32, 256
    base = [synth.cells(0xDCDF0002 + c, 0, T, 0, S, 0, S, np.int32) for c in range(min(args.chunks, 16))]
    tiles = [base[c % len(base)] for c in range(args.chunks)]
    dc.build_batch(tiles[:4])  # warm up (library load, allocations)
    best = None
    for _ in range(args.reps):
        t0 = time.perf_counter()
        out = dc.build_batch(tiles)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    assert not any(isinstance(o, Exception) for o in out)
    cells = args.chunks * T * S * S
    # the C entry point alone (what a Rust caller pays): no Python-side copies of the results
    import ctypes as C
    from dcdf_amd import _lib as L
    from dcdf_amd.chunk import _desc
    descs = (L.TileDesc * len(tiles))()
    for i, a in enumerate(tiles):
        descs[i] = _desc(a, 0, False)
    cbest = None
    for _ in range(args.reps):
        res = C.POINTER(L.Encoded)()
        t0 = time.perf_counter()
        L.check(L.lib().dcdf_chunk_build_batch(descs, C.c_size_t(len(tiles)), 2, L.MEM_HOST, C.byref(res)), "chunk_build")
        dt = time.perf_counter() - t0
        L.lib().dcdf_free_encoded(res, C.c_size_t(len(tiles)))
        cbest = dt if cbest is None else min(cbest, dt)
    print(json.dumps({"entry": "dcdf_chunk_build_batch (host buffers)", "chunks": args.chunks, "cells_per_s": cells / cbest,
                      "seconds": cbest, "input_GB_per_s": cells * 4 / cbest / 1e9, "cells_per_s_incl_python_copies": cells / best,
                      "encoded_bytes": int(sum(o.size for o in out))}))


if __name__ == "__main__":
    main()
