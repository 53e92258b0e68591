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
    print(json.dumps({"entry": "dcdf_chunk_build_batch (host buffers)", "chunks": args.chunks, "cells_per_s": cells / best,
                      "seconds": best, "input_GB_per_s": cells * 4 / best / 1e9,
                      "encoded_bytes": int(sum(o.size for o in out))}))


if __name__ == "__main__":
    main()
