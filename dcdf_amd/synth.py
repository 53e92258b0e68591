"""Deterministic synthetic raster model.  Counter-based so that the numpy version
(this file; used by tests and by bench.py's CPU-baseline sample) and the device generator
(csrc/k2r_synth.hip) produce identical cells without any files.

    v(t,r,c) = base(r,c) + season(t) + event(t, r>>4, c>>4) + speckle(t,r,c)

  base    : three integer cosine-table waves, |base| <= 2048
  season  : 64 * tri(t / 365)
  event   : with prob 1/8 per (t, 16x16 block) a constant in [-256, 256], else 0
  speckle : with prob 1/64 per cell a value in [-8, 8], else 0
  plus    : 5 % of 64x64 blocks are forced constant per instant (exercises elision) and every 50th
            instant repeats its predecessor (exercises eqB / tiny logs)
  rng     : splitmix64(seed ^ (t << 40 ^ r << 20 ^ c) ^ salt)
"""
import numpy as np

MASK = (1 << 64) - 1
_COS = np.round(1024.0 * np.cos(2.0 * np.pi * np.arange(1024) / 1024.0)).astype(np.int64)  # table, |.| <= 1024


def splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _rng(seed, t, r, c, salt):
    key = (np.uint64(seed) ^ (t.astype(np.uint64) << np.uint64(40)) ^ (r.astype(np.uint64) << np.uint64(20)) ^
           c.astype(np.uint64) ^ np.uint64(salt))
    return splitmix64(key)


def cells(seed, t0, t1, r0, r1, c0, c1, dtype=np.int32):
    """Cells [t0:t1, r0:r1, c0:c1] of the raster with the given seed (global coordinates)."""
    with np.errstate(over="ignore"):
        t = np.arange(t0, t1, dtype=np.int64)
        # every 50th instant repeats its predecessor
        te = np.where((t % 50 == 49) & (t > 0), t - 1, t)
        T, R, Cc = np.meshgrid(te, np.arange(r0, r1, dtype=np.int64), np.arange(c0, c1, dtype=np.int64), indexing="ij")
        base = (_COS[(3 * R + 5 * Cc) & 1023] + (_COS[(7 * R - 2 * Cc) & 1023] >> 1) + (_COS[(R + 11 * Cc) & 1023] >> 1))
        ph = T % 365
        season = np.where(ph < 183, ph, 365 - ph) * 64 // 183
        h1 = _rng(seed, T, R >> 4, Cc >> 4, 0x1111)
        event = np.where((h1 & np.uint64(7)) == 0, ((h1 >> np.uint64(8)) % np.uint64(513)).astype(np.int64) - 256, 0)
        h2 = _rng(seed, T, R, Cc, 0x2222)
        speckle = np.where((h2 & np.uint64(63)) == 0, ((h2 >> np.uint64(8)) % np.uint64(17)).astype(np.int64) - 8, 0)
        v = base + season + event + speckle
        h3 = _rng(seed, T, R >> 6, Cc >> 6, 0x3333)
        flat = (h3 % np.uint64(20)) == 0
        flatv = ((h3 >> np.uint64(16)) % np.uint64(4097)).astype(np.int64) - 2048
        v = np.where(flat, flatv, v)
    if np.dtype(dtype) == np.dtype(np.int64):
        return (v * 2 + 1).astype(np.int64)  # "fixed point with the NaN tag bit applied": odd values
    return v.astype(dtype)
