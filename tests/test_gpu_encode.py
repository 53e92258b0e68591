"""Parity tests proper: the HIP encoder, called through the C ABI, against the CPU oracle on the same
seeded inputs, bit for bit.  Run on a real MI355X: pytest -m gpu."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_vectors.json")) as f:
    G = json.load(f)


@pytest.fixture(scope="module")
def dc():
    import dcdf_amd
    from dcdf_amd import _lib
    assert _lib.lib().dcdf_device_name(), "no GPU"
    return dcdf_amd


def farr(name, dtype):
    return np.array([[[float("nan") if v == "nan" else v for v in row] for row in inst] for inst in G[name]], dtype=dtype)


def array_n(n, T=12):
    a8 = np.array(G["array8"], dtype=np.int64)
    a = np.stack([a8[i % 3] for i in range(T)])
    idx = np.arange(n) % 8
    return np.ascontiguousarray(a[:, idx][:, :, idx])


def assert_same(dc, arrays, **kw):
    res = dc.build_batch(arrays, **kw)
    okw = {"fractional_bits": kw.get("fractional_bits", 0), "round_": kw.get("round", False)}
    for a, r in zip(arrays, res):
        assert not isinstance(r, Exception), r
        ref, rs, rl, _ = O.chunk_build(a, want_snapshots=True, **okw)
        data = r.data.write_to()
        assert (r.snapshots, r.logs) == (rs, rl)
        if data != ref:
            n = min(len(data), len(ref))
            first = next((i for i in range(n) if data[i] != ref[i]), n)
            raise AssertionError("bytes differ: len %d vs %d, first diff at %d, shape %s" % (len(data), len(ref), first, a.shape))
        a3 = np.asarray(a)
        fin = a3.reshape(a3.shape[0], -1)
        if a3.dtype.kind == "i":
            assert (r.minmax[:, 0] == fin.min(1)).all() and (r.minmax[:, 1] == fin.max(1)).all()
        r.data.close()


def test_reference_fixtures(dc):  # testing.rs:200-249 (array8 cycled / tiled), both int encodings
    arrays = []
    for n in (8, 16, 32, 64):
        arrays += [array_n(n), array_n(n).astype(np.int32)]
    arrays.append(np.array(G["array9"], dtype=np.int64))
    pad = np.zeros((3, 9, 9), dtype=np.int64) + 5  # snapshot.rs:560-572, log.rs:939-955
    pad[:, :8, :8] = np.array(G["array8"], dtype=np.int64)
    pad[0] = np.array(G["array9"], dtype=np.int64)[0]
    arrays.append(pad)
    assert_same(dc, arrays)


def test_random_kinds_all_sidelens(dc):
    arrays = []
    for shape in [(5, 8, 8), (7, 16, 16), (6, 32, 32), (5, 64, 64), (4, 128, 128), (3, 256, 256)]:
        for kind in ["small", "wide", "noise", "const", "sparse"]:
            rng = np.random.default_rng(hash((shape, kind)) & 0xFFFF)
            T, R, Cc = shape
            if kind == "small":
                a = rng.integers(-3, 4, size=shape)
            elif kind == "wide":
                a = rng.integers(-(2 ** 29), 2 ** 29, size=shape)
                a[1] = a[0] + rng.integers(-300, 300, size=(R, Cc))
            elif kind == "noise":
                a = rng.integers(0, 70000, size=shape)
            elif kind == "const":
                a = np.zeros(shape, dtype=np.int64) + 5
                a[2:] += 1
            else:
                base = rng.integers(-100, 100, size=(R, Cc))
                a = np.stack([base.copy() for _ in range(T)])
                for i in range(1, T):
                    for _ in range(3):
                        a[i, rng.integers(R), rng.integers(Cc)] += rng.integers(-500, 500)
                    if i == 3 % T:
                        a[i, : R // 2, : Cc // 2] += 7
            arrays += [a.astype(np.int64), a.astype(np.int32)]
    assert_same(dc, arrays)


def test_padded_and_ragged(dc):  # superchunk.rs:129-142 edge tiles; snapshot.rs:453-457 None cells
    arrays = []
    for rows, cols in [(5, 3), (8, 7), (9, 9), (16, 9), (17, 4), (13, 31), (33, 20), (64, 1), (100, 77), (129, 130),
                       (7, 8), (200, 256), (256, 255)]:
        rng = np.random.default_rng(rows * 1000 + cols)
        a = rng.integers(-9, 9, size=(5, rows, cols)).astype(np.int64)
        a[2] = a[1]
        a[3] = a[1] + 4
        a[4, : rows // 2] = a[1, : rows // 2]
        b = rng.integers(-40000, 40000, size=(4, rows, cols)).astype(np.int32)
        b[2, :, : max(1, cols // 2)] = b[0, :, : max(1, cols // 2)] - 3
        arrays += [a, b]
    assert_same(dc, arrays)


def test_strided_views(dc):
    from dcdf_amd import synth
    big = synth.cells(7, 0, 4, 0, 96, 0, 160, np.int32)
    tile = big[:, 32:96, 64:128]
    assert_same(dc, [tile, tile[:, :, ::2][:, :32, :32], big.transpose(0, 2, 1)[:, :64, :64]])


def test_float_chunks(dc):  # mmarray.rs:1285,1403; fixed.rs quirks
    arrays32, arrays64 = [], []
    for dtype, lst in ((np.float32, arrays32), (np.float64, arrays64)):
        f8 = farr("farray8", dtype)
        a = np.stack([f8[i % 6] for i in range(12)])
        idx = np.arange(16) % 8
        lst.append(np.ascontiguousarray(a[:, idx][:, :, idx]))
    assert_same(dc, arrays32 + arrays64, fractional_bits=3)
    assert_same(dc, arrays32 + arrays64, fractional_bits=2, round=True)
    r = dc.build_batch(arrays32, fractional_bits=2, round=False)[0]
    assert isinstance(r, Exception) and r.code == -3  # fixed.rs:47-59
    b = arrays64[0].copy()
    b[3, 2, 2] = -np.inf
    r = dc.build_batch([b], fractional_bits=3)[0]
    assert isinstance(r, Exception) and r.code == -2  # fixed.rs:39-41
    neg = -np.abs(arrays64[0])  # negative non-integers: never rounded, never rejected (appendix A.15)
    assert_same(dc, [neg], fractional_bits=1)


def test_254_log_cap(dc):  # chunk.rs:62, block.rs:27
    base = (np.add.outer(np.arange(8), np.arange(8)) % 5).astype(np.int64)
    a = np.stack([base.copy() for _ in range(300)])
    for i in range(1, 300):
        a[i, i % 8, (i * 3) % 8] += 1
    assert_same(dc, [a])


def test_inputs_outside_the_fused_kernel_take_the_universal_one(dc):
    """|v| >= 2^30, sidelen < 8 or > 256, k = 3: declined by the fused kernel, encoded by k2r_generic.hip (more in
    tests/test_gpu_generic.py)."""
    a = np.zeros((2, 8, 8), dtype=np.int64)
    a[1, 3, 3] = 2 ** 30
    assert_same(dc, [a, np.zeros((2, 4, 4), dtype=np.int32), np.zeros((1, 300, 8), dtype=np.int32)])
    b = np.arange(2 * 9 * 9, dtype=np.int32).reshape(2, 9, 9) % 7
    r = dc.Chunk.build(b, k=3)
    assert r.data.write_to() == O.chunk_build(b, k=3)


def test_config2_sample_synthetic_256(dc):
    """BASELINE config 2 shape ([32,256,256], seed 0xDCDF0002 + c), a few chunks, int32 and int64."""
    from dcdf_amd import synth
    arrays = [synth.cells(0xDCDF0002 + c, 0, 32, 0, 256, 0, 256, np.int32) for c in range(3)]
    arrays.append(synth.cells(0xDCDF0002 + 3, 0, 32, 0, 256, 0, 256, np.int64))
    arrays.append(synth.cells(0xDCDF0003, 352, 365, 1024, 1280, 2048, 2304, np.int32))  # ragged last segment (13)
    assert_same(dc, arrays)


def test_adversarial_sets(dc):  # iid noise (every instant a snapshot) and all-constant
    rng = np.random.default_rng(9)
    noise = rng.integers(0, 2 ** 30 - 1, size=(6, 256, 256)).astype(np.int32)
    const = np.zeros((6, 256, 256), dtype=np.int32) + 1234
    assert_same(dc, [noise, const])


def test_float32_full_size_chunks(dc):
    """[32,256,256] float32 chunks (configs[1] as floats): aligned rows go through the converting 16-byte row loader."""
    from dcdf_amd import synth
    arrays = []
    for c in range(2):
        a = synth.cells(0xDCDF0002 + c, 0, 32, 0, 256, 0, 256, np.int32)
        f = (a / 8.0).astype(np.float32)
        f[5, 100, 7] = np.nan
        arrays.append(f)
    assert_same(dc, arrays, fractional_bits=3)
    g = arrays[0][:6].copy()
    g[2, 9, 9] = np.float32(5.0 + 1 / 64.0)
    r = dc.build_batch([g], fractional_bits=3)[0]
    assert isinstance(r, Exception) and r.code == -3
    assert_same(dc, [g], fractional_bits=3, round=True)
    g[2, 9, 9] = np.float32(3e8)  # stored value beyond 2^30: the universal kernel takes the tile
    assert_same(dc, [g], fractional_bits=3, round=True)


def test_wide_model_logs(dc):
    """bench.py --dataset wide in small: the model raster times 300.  Log differences need three bytes (the stash does not take
    them), the snapshot's range exceeds 16 bits (no compact copy), and logs still win chunk.rs:62: the LOG side of the general path."""
    from dcdf_amd import synth
    a = (synth.cells(0xDCDF0003, 0, 8, 256, 512, 512, 768, np.int32).astype(np.int64) * 300).astype(np.int32)
    res = dc.build_batch([a])
    assert res[0].snapshots == 1 and res[0].logs == 7
    assert_same(dc, [a])
    assert_same(dc, [a.astype(np.int64)])


def test_int32_rows_value_range_contract(dc):
    a = np.zeros((2, 16, 16), dtype=np.int32)
    a[1, 9, 9] = 2 ** 30  # outside the fused kernel's contract: re-routed, same bytes as the oracle
    assert_same(dc, [a])
    a[1, 9, 9] = 2 ** 30 - 1
    a[0, 1, 1] = -(2 ** 30)
    assert_same(dc, [a])


@pytest.mark.parametrize("dtype", [np.int32, np.int64, np.float32, np.float64])
def test_row_loader_selection_aligned_and_not(dc, dtype):
    """Every element type has a 16-byte row loader for aligned, unit-stride tiles and falls back to the generic loader
    otherwise (misaligned base, odd row stride, transposed view): all of them must give the oracle's bytes."""
    from dcdf_amd import synth
    fb = 3 if np.dtype(dtype).kind == "f" else 0
    big = synth.cells(11, 0, 5, 0, 66, 0, 70, np.int32)
    big = (big / 8.0).astype(dtype) if fb else big.astype(dtype)
    aligned = np.ascontiguousarray(big[:, :64, :64])
    assert aligned.ctypes.data % 16 == 0
    shifted = big[:, 1:65, 1:65]           # base not 16-byte aligned, row stride 70
    odd = np.ascontiguousarray(big[:, :64, :67])[:, :, :64]  # row stride 67 elements
    transposed = np.ascontiguousarray(big[:, :64, :64].transpose(0, 2, 1)).transpose(0, 2, 1)  # column stride != 1
    assert_same(dc, [aligned, shifted, odd, transposed], fractional_bits=fb)


_SHA_SCRIPT = r"""
import hashlib, sys
import numpy as np
import torch  # before the library: one HIP runtime per process, torch's (as in bench.py)
sys.path.insert(0, %r)
from dcdf_amd import _lib as L, synth
from dcdf_amd.encoder import Encoder
rng = np.random.default_rng(9)
arrays = [rng.integers(-50, 50, size=(t, s, s)).astype(np.int32) for t, s in ((1, 8), (2, 8), (3, 16), (5, 32), (7, 64), (4, 128))]
arrays += [np.zeros((1, 8, 8), dtype=np.int32), synth.cells(0xDCDF0005, 0, 6, 0, 256, 0, 256, np.int32)]
wide = np.zeros((2, 8, 8), dtype=np.int32)
wide[1, 1, 1] = 2 ** 30  # outside the fused kernel's contract: encoded by the universal kernel, hashed like the others
arrays.append(wide)
arrays.append(np.zeros((1, 1500, 8), dtype=np.int32))  # sidelen 2048: refused, digest stays zero
dev = [torch.from_numpy(a).cuda() for a in arrays]
enc = Encoder([(d.data_ptr(), L.DCDF_I32, tuple(s // 4 for s in a.strides), a.shape) for d, a in zip(dev, arrays)], k=2)
enc.run()
dig, _ = enc.object_sha256()
cids = enc.object_cids()
hdr = bytes([0xDC, 0xE0, 0, 0, 0, 1, 2, 4])
lens = set()
for i in range(len(arrays) - 1):
    data = enc.fetch(i)
    lens.add(len(data) %% 64)
    want = hashlib.sha256(hdr + data).digest()
    assert dig[i].tobytes() == want, i
    assert cids[i] == bytes([1, 0x12, 0x12, 0x20]) + want
assert not dig[-1].any() and cids[-1] is None
assert len(lens) > 3  # different tail lengths exercised
print("sha256 ok")
"""


def test_object_sha256_on_device(dc):
    """Content addressing (resolver.rs:126-138 framing, testing.rs:172-183 CID): device SHA-256 of header + chunk bytes
    against hashlib, for chunk lengths around the block boundaries and for a failed tile.  Runs in a child process:
    the device-resident session needs torch for device memory, and torch must initialise HIP before the library does."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _SHA_SCRIPT % root], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sha256 ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("parts", ["2", "4", "8"])
def test_speculative_parts_splice(dc, monkeypatch, parts):
    """K2R_SPLIT=all: every tile is encoded as several work items ([0, T/2), then continuations of decreasing length that assume
    the first block is still open and read instant 0's compact copy from the buffer the first part published) and spliced on
    the device by k_stitch.  Bytes, counters and per-instant (min, max) must equal the sequential encode's -- including tiles
    where a block closes early (the assumption fails: re-encoded whole), padded tiles, other element types and chunks too
    short to split."""
    from dcdf_amd import synth
    monkeypatch.setenv("K2R_SPLIT", "all")
    monkeypatch.setenv("K2R_PARTS", parts)
    arrays = [synth.cells(0xDCDF0003, 32 * s, 32 * s + 32, 256 * i, 256 * i + 256, 0, 256, np.int32) for s, i in [(0, 0), (1, 3), (2, 5)]]
    arrays.append(synth.cells(0xDCDF0003, 352, 365, 0, 256, 256, 512, np.int32))          # the 13-instant last segment
    arrays.append(synth.cells(0xDCDF0002, 0, 9, 0, 200, 0, 256, np.int32))                 # padded
    arrays.append(synth.cells(0xDCDF0002, 0, 8, 0, 128, 0, 128, np.int64))
    noisy = arrays[0].copy()
    noisy[5] = np.random.default_rng(5).integers(0, 1 << 20, size=noisy[5].shape)         # a snapshot inside the first half
    arrays.append(noisy)
    late = arrays[1].copy()
    late[20:] = np.random.default_rng(6).integers(0, 1 << 20, size=late[20:].shape)       # block boundaries in the second half only
    arrays.append(late)
    arrays.append(arrays[2][:3].copy())                                                    # 3 instants: not split
    wide = arrays[0].copy()
    wide[0, 7, 7] = 300000                                                                 # instant 0 beyond 16 bits: no compact copy to share
    arrays.append(wide)
    arrays += [synth.cells(0xDCDF0003, 32 * s, 32 * s + 32, 0, 256, 256 * j, 256 * j + 256, np.int32) for s in range(3) for j in range(8)]
    assert_same(dc, arrays)
    f = (arrays[5][:, :64, :64] // 2 / 8.0).astype(np.float32)
    assert_same(dc, [f], fractional_bits=3)
