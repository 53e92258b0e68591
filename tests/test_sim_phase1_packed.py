"""The PACKED phase 1 of the fused encoder (k2r_encode.h: 16-bit pair arithmetic against the snapshot copy, one stash
allocation per wave and sub-block) in the sequential simulator, bit for bit against the oracle, together with the hand-over
to the scalar phase 1: every level of the tree deciding, empty / single-node logs, equal and uniform regions, second bytes
at every level, the int16 limits of the differences, snapshot ranges beyond 16 bits, blocks closing in the middle of a chunk,
stash overflow, the 254-log cap, float tiles.  `packed` below = instants analysed by the packed form (sim_last_fast_logs)."""
import ctypes

import numpy as np
import pytest

import oracle_lib as O
import sim_lib as S
from dcdf_amd import synth


def packed():
    S.lib().sim_last_fast_logs.restype = ctypes.c_uint32
    return S.lib().sim_last_fast_logs()


def check(a, **kw):
    okw = {k: v for k, v in kw.items() if k in ("fractional_bits", "round_")}
    ref, rs, rl, _ = O.chunk_build(a, want_snapshots=True, **okw)
    st, data, ns, nl, mm = S.encode(a, want_minmax=True, **kw)
    assert st == 0
    assert (ns, nl) == (rs, rl)
    if data != ref:
        n = min(len(data), len(ref))
        first = next((i for i in range(n) if data[i] != ref[i]), n)
        raise AssertionError("bytes differ: len %d vs %d, first diff at %d" % (len(data), len(ref), first))
    fin = np.asarray(a).reshape(a.shape[0], -1)
    if np.asarray(a).dtype.kind == "i":
        assert (mm[:, 0] == fin.min(1)).all() and (mm[:, 1] == fin.max(1)).all()
    return nl, packed()


@pytest.mark.parametrize("side", [16, 32, 64, 128, 256])
@pytest.mark.parametrize("dtype", [np.int32, np.int64])
def test_synthetic_all_instants_packed(side, dtype):
    a = synth.cells(0xDCDF0002, 0, 6, 0, side, 0, side, dtype)
    nl, nf = check(a)
    # every instant but the first (logs, and the ones that turn out to be snapshots); 8-byte rows stay with the scalar form
    assert nf == (5 if dtype == np.int32 else 0)


def test_scalar_form_gives_the_same_bytes(monkeypatch):
    a = synth.cells(0xDCDF0003, 0, 8, 0, 256, 0, 256, np.int32)
    st, d1, _, _, _ = S.encode(a, want_minmax=True)
    assert packed() == 7
    monkeypatch.setenv("K2R_SIM_NO_FAST", "1")
    st, d2, _, _, _ = S.encode(a, want_minmax=True)
    assert packed() == 0 and d1 == d2


@pytest.mark.parametrize("side", [64, 256])
def test_structured_cases(side):
    rng = np.random.default_rng(side)
    base = rng.integers(-300, 300, size=(side, side)).astype(np.int32)
    inst = [base]
    inst.append(base.copy())                                   # identical: single-node log, eqB = 1 at the root
    inst.append(base + 7)                                      # equal everywhere with a non-zero diff
    inst.append(np.full_like(base, 12))                        # uniform instant: single-node log, eqB = 0
    b = base.copy(); b[0, 0] += 1; inst.append(b)              # one cell: one path root -> cell
    b = base.copy(); b[side - 1, side - 1] -= 300; inst.append(b)   # last cell, second byte on the way
    b = base.copy(); b[: side // 2] += 5; inst.append(b)       # a height-(H-1) split: top levels decide
    b = base.copy(); b[:, : side // 4] = 3; inst.append(b)     # uniform quarter next to untouched rest
    b = base.copy(); b[8:16, 8:16] += 1000; inst.append(b)     # one 8x8 block equal with a long diff
    b = base.copy(); b[4:8, 4:8] = 0; inst.append(b)           # a uniform height-2 node
    b = base.copy(); b[32:64, 0:32] += rng.integers(-2, 3, size=(32, 32)); inst.append(b)  # one dense height-5 node
    b = base + rng.integers(-1, 2, size=base.shape) * 200; inst.append(b)   # dense, many second bytes
    a = np.stack(inst).astype(np.int32)
    nl, nf = check(a)
    assert nf == len(inst) - 1
    check(a.astype(np.int64))


def test_int16_limits_of_the_differences():
    side = 64
    base = np.zeros((side, side), dtype=np.int32) + 100
    base[::2] += 3
    ok1 = base.copy(); ok1[5, 5] += 32767 - 3                  # largest difference a block bound of 32767 admits
    ok2 = base.copy(); ok2[6, 6] -= 32768 - 3
    wide1 = base.copy(); wide1[7, 7] += 32768                  # beyond int16: the scalar form takes the instant
    wide2 = base.copy(); wide2[9, 9] -= 32769
    a = np.stack([base, ok1, ok2, wide1, wide2, base + 1, base + 2, base + 3, base]).astype(np.int32)
    nl, nf = check(a)
    # instants 1, 2 packed; 3 fails (skip 1 -> instant 4 scalar); 5 packed ...
    assert nf == 6
    # cells far from the snapshot's range (differences beyond int16 everywhere): always handed over, never wrong
    far = base + 40000
    a = np.stack([base, far, base + 2, far - 80000])
    nl, nf = check(a)
    assert nf <= 1
    # snapshot range itself beyond 16 bits: no snapshot copy at all
    big = base.copy(); big[0, 0] = 70000
    nl, nf = check(np.stack([big, big + 1, big]))
    assert nf == 0


def test_blocks_closing_mid_chunk_and_resuming():
    side = 64
    rng = np.random.default_rng(3)
    a = [rng.integers(0, 50, size=(side, side))]
    for i in range(1, 14):
        if i in (4, 9):
            a.append(rng.integers(0, 5000, size=(side, side)))   # unrelated instant: becomes a Snapshot
        else:
            b = a[-1].copy()
            b[rng.integers(side), rng.integers(side)] += 3
            a.append(b)
    a = np.stack(a).astype(np.int32)
    ref, ns, nl, snaps = O.chunk_build(a, want_snapshots=True)
    assert ns == 3
    nl2, nf = check(a)
    assert nf == 13


def test_noise_every_instant_a_snapshot():
    rng = np.random.default_rng(9)
    a = rng.integers(0, 60000, size=(12, 64, 64)).astype(np.int32)
    nl, nf = check(a)
    assert nl == 0


def test_stash_overflow_falls_back(monkeypatch):
    a = synth.cells(0xDCDF0002, 0, 4, 0, 256, 0, 256, np.int32)
    nl, nf = check(a)
    assert nl == 3 and nf == 3
    for words in ("1", "7", "3000"):
        monkeypatch.setenv("K2R_SIM_STASH_WORDS", words)   # far too small for an instant of this raster
        nl, nf = check(a)
        assert nl == 3 and nf == 3


@pytest.mark.parametrize("ftype", [np.float32, np.float64])
def test_float_tiles(ftype):
    a = synth.cells(0xDCDF0004, 0, 5, 0, 64, 0, 64, np.int32)
    f = (a / 8.0).astype(ftype)
    f[2, 10, 10] = np.nan
    nl, nf = check(f, fractional_bits=3)
    assert nf == (4 if ftype == np.float32 else 0)
    g = f.copy()
    g[3, 1, 1] = ftype(0.01)  # not representable with 3 fractional bits: the conversion error is the reference's
    assert S.encode(g, fractional_bits=3)[0] == -3


def test_254_cap():
    base = (np.add.outer(np.arange(64), np.arange(64)) % 5).astype(np.int32)
    a = np.stack([base.copy() for _ in range(260)])
    for i in range(1, 260):
        a[i, i % 64, (i * 3) % 64] += 1
    nl, nf = check(a)
    assert nf == 259


def test_strided_unaligned_full_tiles():
    big = synth.cells(0xDCDF0005, 0, 5, 0, 130, 0, 131, np.int32)
    a = big[:, 1:129, 3:131]  # 128 x 128 view: generic loads, no padding
    nl, nf = check(a)
    assert nf == 4


@pytest.mark.parametrize("seed", range(8))
def test_random_mixtures(seed):
    rng = np.random.default_rng(100 + seed)
    side = [64, 128, 64, 256, 64, 128, 32, 16][seed]
    T = 6
    base = rng.integers(-1000, 1000, size=(side, side))
    a = [base]
    for i in range(1, T):
        b = a[0].copy() if rng.random() < 0.7 else a[-1].copy()
        for _ in range(int(rng.integers(1, 6))):
            h = int(2 ** rng.integers(0, 7)); w = int(2 ** rng.integers(0, 7))
            r = int(rng.integers(0, side - min(h, side) + 1)); c = int(rng.integers(0, side - min(w, side) + 1))
            kind = rng.integers(4)
            if kind == 0:
                b[r:r + h, c:c + w] += int(rng.integers(-600, 600))
            elif kind == 1:
                b[r:r + h, c:c + w] = int(rng.integers(-1000, 1000))
            elif kind == 2:
                b[r:r + h, c:c + w] += rng.integers(-200, 200, size=b[r:r + h, c:c + w].shape)
            else:
                b[r:r + h, c:c + w] += int(rng.integers(-40000, 40000))  # may leave int16: the scalar form's business
        a.append(b)
    a = np.stack(a).astype(np.int64 if seed % 2 else np.int32)
    check(a)
