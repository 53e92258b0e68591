"""Exhaustive round-trip sweeps on the oracle, modelled on the reference's own unit tests
(snapshot.rs:574-870, log.rs:957-1616, block.rs:201-304, chunk.rs:426-565, mmstruct.rs:452-459,
mmarray.rs:1050-1502).  Truth is the raw array / brute-force search (testing.rs:38-87)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_vectors.json")) as f:
    G = json.load(f)


def arr(name, dtype=np.int64):
    return np.array([[[float("nan") if v == "nan" else v for v in row] for row in inst] for inst in G[name]], dtype=dtype)


def array8_100():  # testing.rs:200-240
    a = arr("array8")
    return np.stack([a[i % 3] for i in range(100)])


def array_n(n):  # testing.rs:242-249
    a = array8_100()
    idx = np.arange(n) % 8
    return a[:, idx][:, :, idx]


def farray_n(n):  # testing.rs:251-339
    a = arr("farray8", np.float32)
    a = np.stack([a[i % 6] for i in range(100)])
    idx = np.arange(n) % 8
    return a[:, idx][:, :, idx]


@pytest.mark.parametrize("name,k", [("array8", 2), ("array9", 2), ("array9", 3), ("array8", 3)])
def test_snapshot_exhaustive(name, k):  # snapshot.rs:574-870
    a = arr(name)
    for i in range(3):
        r = O.sl_exhaustive(a[i], a[i], 0, k=k, lo=4, hi=9)
        assert r["windows"] > 0
        assert r["bad_gets"] == 0 and r["bad_window_cells"] == 0 and r["bad_searches"] == 0


@pytest.mark.parametrize("name,k", [("array8", 2), ("array9", 2), ("array9", 3)])
def test_log_exhaustive(name, k):  # log.rs:957-1616
    a = arr(name)
    for s, t in [(0, 1), (0, 2), (1, 2), (0, 0)]:
        r = O.sl_exhaustive(a[s], a[t], 1, k=k, lo=4, hi=9)
        assert r["bad_gets"] == 0 and r["bad_window_cells"] == 0 and r["bad_searches"] == 0


def test_log_single_node_trees():  # log.rs single-node special cases (log.rs:181-187, 318-327)
    s = np.zeros((8, 8), dtype=np.int64) + 20
    t = np.zeros((8, 8), dtype=np.int64) + 21
    a8 = arr("array8")
    for ss, tt in [(s, t), (s, a8[0]), (a8[0], a8[0] + 3)]:
        r = O.sl_exhaustive(ss, tt, 1, k=2, lo=int(tt.min()) - 1, hi=int(tt.min()) + 3)
        assert r["bad_gets"] == 0 and r["bad_window_cells"] == 0 and r["bad_searches"] == 0
    # single-node (uniform) log over a multi-node snapshot: get/window are exact ...
    r = O.sl_exhaustive(a8[0], t, 1, k=2, lo=40, hi=41)  # range misses every value, like log.rs:1521-1557
    assert r["bad_gets"] == 0 and r["bad_window_cells"] == 0 and r["bad_searches"] == 0


def test_log_search_uniform_log_reference_quirk():
    """REFERENCE QUIRK (restated faithfully, DESIGN.md "quirks"): Log::search_window (log.rs:519-551) has no
    special case for a uniform single-node log (T=[0], eq=[0]) over a multi-node snapshot, unlike get
    (log.rs:184) and fill_window (log.rs:318).  It descends the snapshot as if eq were 1, i.e. it reports the
    cells where snapshot + (max_t - max_s) is in range.  The reference's own test (log.rs:1521-1557) only
    searches ranges that miss every value, so it passes there.  The oracle reproduces the reference."""
    a8 = arr("array8")
    t = np.zeros((8, 8), dtype=np.int64) + 21
    c = O.Chunk(O.chunk_build_forced(np.stack([a8[0], t]), 2, 2))
    q = a8[0] + (21 - int(a8[0].max()))
    for lo in range(12, 24):
        for hi in range(lo, 24):
            got = set(map(tuple, c.search(1, 2, 1, 7, 2, 8, lo, hi)))
            want = set((1, int(y), int(x)) for y, x in zip(*np.nonzero((q >= lo) & (q <= hi))) if 1 <= y < 7 and 2 <= x < 8)
            assert got == want


def _check_chunk(data, raw, truth, dtype):
    c = O.Chunk(data)
    assert c.size == len(data)  # chunk.rs:572-574
    assert c.serialize() == data
    assert c.shape == truth.shape
    T, R, Cc = truth.shape
    rng = np.random.default_rng(0)
    # get
    for _ in range(200):
        i, r, cc = int(rng.integers(T)), int(rng.integers(R)), int(rng.integers(Cc))
        assert c.get(i, r, cc) == raw[i, r, cc]
    # cell
    for _ in range(20):
        r, cc = int(rng.integers(R)), int(rng.integers(Cc))
        s = int(rng.integers(T))
        e = int(rng.integers(s, T + 1))
        np.testing.assert_array_equal(c.fill_cell(s, e, r, cc), raw[s:e, r, cc])
    # window (typed output)
    for _ in range(40):
        s = int(rng.integers(T)); e = int(rng.integers(s + 1, T + 1))
        t = int(rng.integers(R)); b = int(rng.integers(t + 1, R + 1))
        l = int(rng.integers(Cc)); rr = int(rng.integers(l + 1, Cc + 1))
        w = c.fill_window(s, e, t, b, l, rr, dtype=dtype)
        np.testing.assert_array_equal(w, truth[s:e, t:b, l:rr])
    return c


def test_chunk_i64_fixture():  # mmstruct.rs:452-459, mmarray.rs:1168
    a = array_n(16)
    data, ns, nl, si = O.chunk_build(a, want_snapshots=True)
    assert ns + nl == 100 and si[0] == 0
    c = _check_chunk(data, a, a, np.int64)
    assert sum(c.block_lengths()) == 100
    # search vs brute force (testing.rs:63-87), results compared as sets
    rng = np.random.default_rng(1)
    for _ in range(30):
        s = int(rng.integers(100)); e = int(rng.integers(s + 1, 101))
        t = int(rng.integers(16)); b = int(rng.integers(t + 1, 17))
        l = int(rng.integers(16)); r = int(rng.integers(l + 1, 17))
        lo = int(rng.integers(2, 10)); hi = int(rng.integers(lo, 10))
        got = set(map(tuple, c.search(s, e, t, b, l, r, lo, hi)))
        sub = a[s:e, t:b, l:r]
        want = set((int(i) + s, int(y) + t, int(x) + l) for i, y, x in zip(*np.nonzero((sub >= lo) & (sub <= hi))))
        assert got == want
    # lower/upper auto-swap (chunk.rs:214)
    assert set(map(tuple, c.search(0, 3, 0, 16, 0, 16, 7, 5))) == set(map(tuple, c.search(0, 3, 0, 16, 0, 16, 5, 7)))


def test_chunk_i32_fixture():  # mmarray.rs:1050
    a = array_n(16).astype(np.int32)
    data = O.chunk_build(a)
    assert data[0] == 4 and data[1] == 0
    _check_chunk(data, a.astype(np.int64), a, np.int32)


@pytest.mark.parametrize("dtype,enc", [(np.float32, 32), (np.float64, 64)])
def test_chunk_float_fixture(dtype, enc):  # mmarray.rs:1285,1403 (farray(16), 3 fractional bits, Precise)
    a = farray_n(16).astype(dtype)
    rnd, bits = O.suggest_fraction(a, "f32" if dtype == np.float32 else "f64")
    assert (rnd, bits) == (False, 3)
    data = O.chunk_build(a, fractional_bits=bits, round_=False)
    assert data[0] == enc and data[1] == 3
    c = O.Chunk(data)
    w = c.fill_window(0, 100, 0, 16, 0, 16, dtype=dtype)
    np.testing.assert_array_equal(np.isnan(w), np.isnan(a))
    np.testing.assert_array_equal(w[~np.isnan(a)], a[~np.isnan(a)])
    # stored fixed-point values: NaN -> 0, finite -> odd (fixed.rs:8-10)
    raw0 = c.get(3, 0, 0)
    assert raw0 == 0
    assert c.get(0, 0, 0) == int(9.5 * 8) * 2 + 1


def test_chunk_forced_blocks_like_reference_helper():  # chunk.rs:397-424
    a = array8_100()
    data = O.chunk_build_forced(a, k=2, block_len=4)
    c = O.Chunk(data)
    assert c.n_blocks == 25 and c.shape == (100, 8, 8)
    _check_chunk(data, a, a, np.int64)


def test_chunk_non_contiguous_input_view():  # mmbuffer.rs:517-522: tile slices of a bigger array
    big = np.random.default_rng(5).integers(-50, 50, size=(6, 40, 48)).astype(np.int32)
    tile = big[:, 8:24, 16:32]
    assert not tile.flags["C_CONTIGUOUS"]
    assert O.chunk_build(tile) == O.chunk_build(np.ascontiguousarray(tile))


def test_chunk_heuristic_extremes():
    rng = np.random.default_rng(2)
    # iid noise: every instant becomes a snapshot (snapshot.size() <= log.size())
    a = rng.integers(0, 2 ** 31 - 1, size=(6, 16, 16)).astype(np.int64)
    data, ns, nl, si = O.chunk_build(a, want_snapshots=True)
    assert (ns, nl) == (6, 0) and si == list(range(6))
    # constant: snapshot of a uniform tile is tiny (single node) -> new snapshot each time too (<=)
    a = np.zeros((5, 16, 16), dtype=np.int64) + 7
    data, ns, nl, si = O.chunk_build(a, want_snapshots=True)
    _check_chunk(data, a, a, np.int64)
    # smooth base + sparse changes: logs win
    base = (np.add.outer(np.arange(32), np.arange(32)) % 17).astype(np.int64)
    a = np.stack([base.copy() for _ in range(10)])
    for i in range(1, 10):
        a[i, i, i] += 3
    data, ns, nl, si = O.chunk_build(a, want_snapshots=True)
    assert ns == 1 and nl == 9
    _check_chunk(data, a, a, np.int64)


def test_chunk_254_log_cap():  # chunk.rs:62, block.rs:27
    base = (np.add.outer(np.arange(8), np.arange(8)) % 5).astype(np.int64)
    a = np.stack([base.copy() for _ in range(300)])
    for i in range(1, 300):
        a[i, i % 8, (i * 3) % 8] += 1
    data, ns, nl, si = O.chunk_build(a, want_snapshots=True)
    c = O.Chunk(data)
    assert max(c.block_lengths()) <= 255
    assert c.block_lengths()[0] == 255 and si[1] == 255
    _check_chunk(data, a, a, np.int64)


def test_ragged_shapes_roundtrip():
    rng = np.random.default_rng(3)
    for rows, cols in [(1, 1), (1, 7), (5, 3), (9, 9), (17, 4), (13, 31), (33, 20)]:
        a = rng.integers(-9, 9, size=(5, rows, cols)).astype(np.int64)
        a[2] = a[1]
        if rows * cols == 1:
            # a 1x1 tile has sidelen 1 -> empty T; the reference panics on query (snapshot.rs:166)
            data = O.chunk_build(a)
            assert len(data) > 0
            continue
        data = O.chunk_build(a)
        _check_chunk(data, a, a, np.int64)
