"""The universal encoder (dcdf_amd/csrc/k2r_generic.hip) against the oracle: everything the fused kernel declines --
k = 3, sidelen below 8 and above 256, stored values beyond 2^30 (all 8 Dac planes) -- plus the reference's own real-data
fixture (py-dcdf/tests/test_dcdf.py:340-365) and its superchunk fixture shapes (mmstruct.rs:463-479)."""
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


def assert_same(dc, arrays, k=2, fractional_bits=0, round=False):
    fb = fractional_bits if isinstance(fractional_bits, (list, tuple)) else [fractional_bits] * len(arrays)
    rd = round if isinstance(round, (list, tuple)) else [round] * len(arrays)
    res = []
    # (round is per call in the mirror: group by it)
    for flag in (False, True):
        idx = [i for i in range(len(arrays)) if bool(rd[i]) == flag]
        if idx:
            out = dc.build_batch([arrays[i] for i in idx], k=k, fractional_bits=[fb[i] for i in idx], round=flag)
            res += list(zip(idx, out))
    for i, r in res:
        a = arrays[i]
        assert not isinstance(r, Exception), (i, a.shape, r)
        ref, rs, rl, _ = O.chunk_build(a, k=k, fractional_bits=fb[i], round_=bool(rd[i]), want_snapshots=True)
        data = r.data.write_to()
        assert (r.snapshots, r.logs) == (rs, rl), (i, a.shape)
        if data != ref:
            n = min(len(data), len(ref))
            first = next((j for j in range(n) if data[j] != ref[j]), n)
            raise AssertionError("tile %d %s: bytes differ: len %d vs %d, first diff at %d" % (i, a.shape, len(data), len(ref), first))
        if a.dtype.kind == "i":
            fin = a.reshape(a.shape[0], -1)
            assert (r.minmax[:, 0] == fin.min(1)).all() and (r.minmax[:, 1] == fin.max(1)).all()


def array_n(n, T=100):
    a8 = np.array(G["array8"], dtype=np.int64)
    a = np.stack([a8[i % 3] for i in range(T)])
    idx = np.arange(n) % 8
    return np.ascontiguousarray(a[:, idx][:, :, idx])


def test_small_and_large_sidelens_k2(dc):
    rng = np.random.default_rng(1)
    arrays = []
    for rows, cols in [(1, 1), (1, 2), (2, 2), (3, 1), (4, 4), (4, 1), (1, 4), (5, 7), (7, 7)]:
        a = rng.integers(-5, 6, size=(9, rows, cols)).astype(np.int64)
        a[3] = a[2]
        a[4] = a[2] + 1
        arrays += [a, a.astype(np.int32)]
    for rows, cols in [(300, 8), (257, 257), (512, 512), (400, 1000)]:
        b = rng.integers(-40, 40, size=(3, rows, cols)).astype(np.int32)
        b[1] = b[0]
        b[1, 5:9, 2:7] += 3
        b[2, rows // 2:, : cols // 2] = 7
        arrays.append(b)
    assert_same(dc, arrays)


def test_wide_values_all_dac_planes(dc):
    rng = np.random.default_rng(2)
    arrays = []
    for side in (8, 16, 64):
        a = rng.integers(-(2 ** 62), 2 ** 62, size=(4, side, side)).astype(np.int64)
        a[2] = a[1] + rng.integers(-3, 4, size=(side, side))
        arrays.append(a)
        b = rng.integers(-100, 100, size=(5, side, side)).astype(np.int64)
        b[1, 2, 3] = 2 ** 30          # just beyond the fused kernel's contract
        b[2, 1, 1] = -(2 ** 45)
        b[3] = b[2]
        b[4, 0, 0] = 2 ** 62
        arrays.append(b)
    c = np.zeros((3, 256, 256), dtype=np.int64)
    c[1, 100, 100] = 2 ** 40
    c[2, 7, 9] = -(2 ** 33)
    arrays.append(c)
    f = (rng.integers(-1000, 1000, size=(3, 32, 32)) / 1024.0).astype(np.float64)  # 10 fractional bits, then 40: wide stored values
    assert_same(dc, arrays)
    assert_same(dc, [f], fractional_bits=40)


@pytest.mark.parametrize("k", [3, 4, 5])
def test_other_k(dc, k):
    rng = np.random.default_rng(k)
    arrays = [np.array(G["array9"], dtype=np.int64), array_n(8, T=12), array_n(16, T=12)]
    for rows, cols in [(9, 9), (27, 27), (10, 31), (81, 80), (1, 1), (2, 5)]:
        a = rng.integers(-9, 10, size=(6, rows, cols)).astype(np.int64)
        a[2] = a[1]
        a[3, : rows // 2] = a[1, : rows // 2]
        a[5] = a[4] + 2
        arrays.append(a)
    assert_same(dc, arrays, k=k)
    # queries on a k = 3 chunk BUILT on the GPU (chunk.rs:426-565 uses k = 3 builds too)
    a = array_n(9, T=8)[:, :9, :9]
    c = dc.Chunk.build(a, k=3).data
    assert c.write_to() == O.chunk_build(a, k=3)
    np.testing.assert_array_equal(c.fill_window(dc.Cube(0, 8, 0, 9, 0, 9)), a)
    got = set(map(tuple, c.iter_search(dc.Cube(1, 7, 2, 9, 1, 8), 4, 7).tolist()))
    sub = a[1:7, 2:9, 1:8]
    want = set((int(i) + 1, int(y) + 2, int(x) + 1) for i, y, x in zip(*np.nonzero((sub >= 4) & (sub <= 7))))
    assert got == want


def test_depth_follows_the_reference_float_formula(dc):
    """snapshot.rs:118-119 takes ceil(ln(side)/ln(k)) in f64: a 125-wide tile with k = 5 gets FOUR levels (sidelen 625), not
    the three an exact power search finds (ln 125 / ln 5 = 3.0000000000000004).  Encode, open and query must agree with it."""
    rng = np.random.default_rng(125)
    a = rng.integers(-50, 50, size=(5, 125, 125)).astype(np.int64)
    a[2] = a[1]
    a[3, :60] = a[1, :60] + 3
    ref = O.chunk_build(a, k=5)
    assert ref[6 + 1 + 1 + 8:6 + 1 + 1 + 12] == (625).to_bytes(4, "big")  # the first Snapshot's sidelen field (snapshot.rs:48-58)
    assert_same(dc, [a, a[:, :125, :100], a[:, :124, :124]], k=5)
    c = dc.Chunk.build(a, k=5).data  # Chunk::build on the GPU, then dcdf_chunk_open of those bytes
    assert c.write_to() == ref
    np.testing.assert_array_equal(c.fill_window(dc.Cube(0, 5, 0, 125, 0, 125)), a)
    oc = O.Chunk(ref)
    assert c.get(3, 124, 124) == oc.get(3, 124, 124) == int(a[3, 124, 124])
    got = set(map(tuple, c.iter_search(dc.Cube(1, 5, 3, 120, 7, 125), -5, 9).tolist()))
    assert got == set(map(tuple, oc.search(1, 5, 3, 120, 7, 125, -5, 9).tolist()))
    r = dc.build_batch([np.zeros((1, 216, 216), dtype=np.int32)], k=6)[0]  # same formula: 4 levels, sidelen 1296 > 1024
    assert isinstance(r, Exception) and r.code == -8


def test_superchunk_fixture_subchunks(dc):
    """mmstruct.rs:463-479: testing::array(17) as [100,17,17], levels [1,2,2], k = 2 -> superchunk.rs:119-181 cuts it into
    4x4 sub-chunks, ragged 4x1 / 1x4 ones along the edges and a 1x1 corner; every one of them must encode."""
    data = array_n(17)
    assert data.shape == (100, 17, 17)
    tiles = []
    for top, bottom, left, right in [(0, 16, 0, 16), (0, 16, 16, 17), (16, 17, 0, 16), (16, 17, 16, 17)]:
        rows, cols = bottom - top, right - left
        if max(rows, cols) <= 4:  # needed_levels <= sublevels[0]: a Chunk (superchunk.rs:155-165)
            tiles.append(data[:, top:bottom, left:right])
            continue
        for r in range(top, bottom, 4):
            for c in range(left, right, 4):
                tiles.append(data[:, r:min(r + 4, bottom), c:min(c + 4, right)])
    assert sorted(set(t.shape[1:] for t in tiles)) == [(1, 1), (1, 4), (4, 1), (4, 4)] and len(tiles) == 25
    assert_same(dc, tiles)


def test_cpc_precip_real_data_fixture(dc):
    """py-dcdf/tests/test_dcdf.py:340-365: one 360 x 720 float32 day of CPC precipitation, k2_levels [4, 6] -> 64 x 64 sub-chunks
    (ragged at the bottom / right), per-tile fractional bits (superchunk.rs:167 -> mmbuffer.rs:596-613, on the GPU here).
    Stored values reach 2^38: no tile may be refused, every tile's bytes == oracle."""
    day = np.load(os.path.join(HERE, "golden", "cpc_precip_day.npz"))["precip"]
    assert day.shape == (360, 720) and day.dtype == np.float32
    cube = day.reshape(1, 360, 720)
    tiles, fbs, rounds = [], [], []
    for r in range(0, 360, 64):
        for c in range(0, 720, 64):
            t = cube[:, r:r + 64, c:c + 64]
            kind, bits = dc.suggest_fraction(t)
            want = O.suggest_fraction(np.ascontiguousarray(t), "f32")
            assert (kind == "round", bits) == (bool(want[0]), want[1])
            tiles.append(t)
            fbs.append(bits)
            rounds.append(kind == "round")
    assert len(tiles) == 72 and max(fbs) >= 20
    assert_same(dc, tiles, fractional_bits=fbs, round=rounds)
    # and three instants of it (the day, the day shifted, the day again): logs with wide values
    three = np.stack([day, np.roll(day, 3, axis=1), day]).astype(np.float32)
    t3 = [three[:, r:r + 64, c:c + 64] for r in (0, 128, 320) for c in (0, 256, 704)]
    fb3 = []
    for t in t3:
        kind, bits = dc.suggest_fraction(t)
        fb3.append(bits)
        assert kind == "precise"
    assert_same(dc, t3, fractional_bits=fb3)


def test_too_large_for_the_universal_kernel_is_reported(dc):
    r = dc.build_batch([np.zeros((1, 1500, 8), dtype=np.int32)])[0]
    assert isinstance(r, Exception) and r.code == -8
