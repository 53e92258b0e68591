#!/usr/bin/env python3
"""One-off randomized stress of the GPU encode path against the oracle: many small tiles of random shape, element type
and value distribution in one batch (all kernel classes at once), plus window/search spot checks."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import dcdf_amd as dc
    import oracle_lib as O
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    tiles, fbs = [], []
    for _ in range(300):
        T = int(rng.integers(1, 8))
        R, Cc = int(rng.integers(5, 140)), int(rng.integers(5, 140))
        if rng.random() < 0.3:
            R = Cc = int(2 ** rng.integers(3, 8))
        kind = rng.integers(0, 4)
        if kind == 0:
            a = rng.integers(-3, 4, size=(T, R, Cc))
        elif kind == 1:
            base = np.add.outer(np.arange(R) * 2, np.arange(Cc) * 3)
            a = np.stack([base + 5 * t + rng.integers(-1, 2, size=(R, Cc)) * (rng.random((R, Cc)) < 0.05) for t in range(T)])
        elif kind == 2:
            a = np.stack([(rng.integers(0, 3, size=((R + 7) // 8, (Cc + 7) // 8)).repeat(8, 0).repeat(8, 1))[:R, :Cc] * 1000 for _ in range(T)])
        else:
            a = rng.integers(-(2 ** 28), 2 ** 28, size=(T, R, Cc))
            a[1:] = a[0] + rng.integers(-40000, 40000, size=(T - 1, R, Cc))
        dt = [np.int32, np.int64, np.float32, np.float64][int(rng.integers(0, 4))]
        fb = 0
        if np.dtype(dt).kind == "f":
            fb = int(rng.integers(0, 5))
            a = np.clip(a, -(2 ** 21), 2 ** 21)
            a = (a / float(1 << fb)).astype(dt)
            if rng.random() < 0.3:
                a[rng.integers(0, T), rng.integers(0, R), rng.integers(0, Cc)] = np.nan
        else:
            a = a.astype(dt)
        tiles.append(np.ascontiguousarray(a))
        fbs.append(fb)
    out = dc.build_batch(tiles, fractional_bits=fbs)
    bad = 0
    for i, (a, fb, o) in enumerate(zip(tiles, fbs, out)):
        ref = O.chunk_build(a, fractional_bits=fb)
        if isinstance(o, Exception) or o.data.write_to() != ref:
            bad += 1
            print("MISMATCH tile", i, a.shape, a.dtype, fb, o if isinstance(o, Exception) else "bytes differ")
    for i in rng.integers(0, len(tiles), 20):
        a, o = tiles[i], out[i]
        T, R, Cc = a.shape
        w = o.data.fill_window(dc.Cube(0, T, 0, R, 0, Cc), dtype=np.int64)  # stored values (fixed point for float tiles)
        oc = O.Chunk(o.data.write_to())
        assert (np.asarray(w) == oc.fill_window(0, T, 0, R, 0, Cc, dtype=np.int64)).all(), "window mismatch tile %d" % i
    print("stress: %d tiles, %d mismatches" % (len(tiles), bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
