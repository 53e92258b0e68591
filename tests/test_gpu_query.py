"""Query-path parity on the GPU: get / fill_cell / fill_window / search through the C ABI against the oracle
(and against the raw arrays), modelled on chunk.rs:426-565, block.rs:201-304, snapshot.rs / log.rs sweeps."""
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


def arr(name, dtype=np.int64):
    return np.array([[[float("nan") if v == "nan" else v for v in row] for row in inst] for inst in G[name]], dtype=dtype)


def array_n(n, T=100):
    a8 = arr("array8")
    a = np.stack([a8[i % 3] for i in range(T)])
    idx = np.arange(n) % 8
    return np.ascontiguousarray(a[:, idx][:, :, idx])


def brute(a, s, e, t, b, l, r, lo, hi):
    sub = a[s:e, t:b, l:r]
    return set((int(i) + s, int(y) + t, int(x) + l) for i, y, x in zip(*np.nonzero((sub >= lo) & (sub <= hi))))


def test_forced_blocks_exhaustive_8x8(dc):
    """chunk.rs:397-565 helper layout (blocks of 4): every window, every [lower,upper] in 4..9, k = 2 and 3.
    The encoded bytes come from the oracle (forced blocks / k=3 are not what Chunk::build emits)."""
    a = np.stack([arr("array8")[i % 3] for i in range(8)])
    for k in (2, 3):
        data = O.chunk_build_forced(a, k=k, block_len=4)
        c = dc.Chunk(data)
        oc = O.Chunk(data)
        assert c.shape() == [8, 8, 8]
        for i in range(8):
            for r in range(8):
                for cc in range(8):
                    assert c.get(i, r, cc) == a[i, r, cc]
        for top in range(0, 8, 1):
            for bottom in range(top + 1, 9, 2):
                for left in range(0, 8, 1):
                    for right in range(left + 1, 9, 3):
                        w = c.fill_window(dc.Cube(0, 8, top, bottom, left, right))
                        assert (w == a[:, top:bottom, left:right]).all()
                        for lo, hi in [(4, 9), (5, 5), (6, 8), (3, 4), (9, 12)]:
                            got = set(map(tuple, c.iter_search(dc.Cube(1, 7, top, bottom, left, right), lo, hi).tolist()))
                            assert got == brute(a, 1, 7, top, bottom, left, right, lo, hi)
                            assert got == set(map(tuple, oc.search(1, 7, top, bottom, left, right, lo, hi).tolist()))
        c.close()


def test_array9_padding_k2_k3(dc):  # snapshot.rs:602-625, log.rs:1039-1077
    a = arr("array9")
    for k in (2, 3):
        data = O.chunk_build_forced(a, k=k, block_len=3)
        c = dc.Chunk(data)
        w = c.fill_window(dc.Cube(0, 3, 0, 9, 0, 9))
        assert (w == a).all()
        for lo in range(1, 11):
            got = set(map(tuple, c.iter_search(dc.Cube(0, 3, 2, 9, 1, 9), lo, lo + 2).tolist()))
            assert got == brute(a, 0, 3, 2, 9, 1, 9, lo, lo + 2)
        c.close()


def test_chunk_fixture_i64_built_on_gpu(dc):  # mmstruct.rs:452-459: Chunk::build([100,16,16])
    a = array_n(16)
    b = dc.Chunk.build(a)
    c = b.data
    assert b.snapshots + b.logs == 100 and c.shape() == [100, 16, 16]
    rng = np.random.default_rng(0)
    for _ in range(100):
        i, r, cc = int(rng.integers(100)), int(rng.integers(16)), int(rng.integers(16))
        assert c.get(i, r, cc) == a[i, r, cc]
    for _ in range(20):
        r, cc = int(rng.integers(16)), int(rng.integers(16))
        s = int(rng.integers(100)); e = int(rng.integers(s, 101))
        np.testing.assert_array_equal(c.fill_cell(s, e, r, cc), a[s:e, r, cc])
    oc = O.Chunk(c.write_to())
    for _ in range(40):
        s = int(rng.integers(100)); e = int(rng.integers(s + 1, 101))
        t = int(rng.integers(16)); bo = int(rng.integers(t + 1, 17))
        l = int(rng.integers(16)); rr = int(rng.integers(l + 1, 17))
        np.testing.assert_array_equal(c.fill_window(dc.Cube(s, e, t, bo, l, rr)), a[s:e, t:bo, l:rr])
        lo = int(rng.integers(2, 10)); hi = int(rng.integers(lo, 10))
        got = set(map(tuple, c.iter_search(dc.Cube(s, e, t, bo, l, rr), lo, hi).tolist()))
        assert got == brute(a, s, e, t, bo, l, rr, lo, hi)
        assert got == set(map(tuple, oc.search(s, e, t, bo, l, rr, lo, hi).tolist()))
    # reversed bounds are swapped like geom::Cube::new / chunk.rs:214
    got = c.iter_search(dc.Cube(5, 2, 9, 1, 12, 3), 7, 5)
    assert set(map(tuple, got.tolist())) == brute(a, 2, 5, 1, 9, 3, 12, 5, 7)
    res = got.tolist()
    assert res == sorted(res)  # deterministic order: (instant,row,col) ascending
    with pytest.raises(dc.DcdfError):
        c.get(100, 0, 0)  # mmarray.rs:218-229 bounds panic -> DCDF_ERR_BOUNDS


def test_typed_windows(dc):  # mmbuffer.rs:505,525,560,622 output conversions
    a32 = array_n(16, T=10).astype(np.int32)
    c = dc.Chunk.build(a32).data
    assert c.encoding == 4
    np.testing.assert_array_equal(c.fill_window(dc.Cube(0, 10, 0, 16, 0, 16)), a32)
    out = np.zeros((10, 20, 24), dtype=np.int32)[:, 2:18, 4:20]  # strided caller buffer
    c.fill_window(dc.Cube(0, 10, 0, 16, 0, 16), out=out)
    np.testing.assert_array_equal(out, a32)
    for dtype in (np.float32, np.float64):
        f8 = arr("farray8", dtype)
        a = np.stack([f8[i % 6] for i in range(12)])
        idx = np.arange(16) % 8
        a = np.ascontiguousarray(a[:, idx][:, :, idx])
        c = dc.Chunk.build(a, fractional_bits=3).data
        w = c.fill_window(dc.Cube(0, 12, 0, 16, 0, 16))
        assert w.dtype == dtype
        np.testing.assert_array_equal(np.isnan(w), np.isnan(a))
        np.testing.assert_array_equal(w[~np.isnan(a)], a[~np.isnan(a)])
        assert c.get(3, 0, 0) == 0 and c.get(0, 0, 0) == int(9.5 * 8) * 2 + 1  # NaN -> 0, finite -> odd


def test_log_search_reference_quirk_reproduced(dc):
    """Uniform single-node log over a multi-node snapshot: the reference's search descends as if eqB were 1
    (see tests/test_oracle_roundtrip.py::test_log_search_uniform_log_reference_quirk).  Drop-in => same set."""
    a8 = arr("array8")
    for tv in (21, 5):
        t = np.zeros((8, 8), dtype=np.int64) + tv
        data = O.chunk_build_forced(np.stack([a8[0], t]), 2, 2)
        c, oc = dc.Chunk(data), O.Chunk(data)
        for lo in range(-2, 24):
            for hi in range(lo, 24, 3):
                got = set(map(tuple, c.iter_search(dc.Cube(1, 2, 1, 7, 2, 8), lo, hi).tolist()))
                assert got == set(map(tuple, oc.search(1, 2, 1, 7, 2, 8, lo, hi).tolist()))
    # the same shape on a tile that HAS side-16 tables (sidelen 64): the quirk instant must bypass them, the others use them
    rng = np.random.default_rng(64)
    s64 = rng.integers(0, 40, size=(64, 64)).astype(np.int64)
    for tv in (55, 17, -3):
        data = O.chunk_build_forced(np.stack([s64, np.zeros((64, 64), dtype=np.int64) + tv, s64 + 1]), 2, 3)
        c, oc = dc.Chunk(data), O.Chunk(data)
        for lo, hi in [(-10, 100), (0, 39), (10, 20), (tv, tv), (tv - 5, tv + 5), (40, 60), (-8, -1), (16, 18), (39, 56)]:
            for cube in [(0, 3, 0, 64, 0, 64), (1, 2, 5, 40, 17, 64), (1, 3, 30, 34, 0, 7)]:
                got = set(map(tuple, c.iter_search(dc.Cube(*cube), lo, hi).tolist()))
                assert got == set(map(tuple, oc.search(*cube, lo, hi).tolist())), (tv, lo, hi, cube)


def test_synthetic_256_roundtrip_and_batches(dc):
    import ctypes as C
    from dcdf_amd import synth, _lib as L
    a = synth.cells(0xDCDF0001, 0, 16, 0, 256, 0, 256, np.int32)  # BASELINE config 1 shape
    b = dc.Chunk.build(a)
    c = b.data
    np.testing.assert_array_equal(c.fill_window(dc.Cube(0, 16, 0, 256, 0, 256)), a)
    rng = np.random.default_rng(1)
    nq = 100
    cubes = (L.Cube * nq)()
    offs = np.zeros(nq, dtype=np.uint64)
    total = 0
    spec = []
    for q in range(nq):
        s = int(rng.integers(16)); e = min(16, s + int(rng.integers(1, 9)))
        t = int(rng.integers(256)); bo = min(256, t + int(rng.integers(1, 65)))
        l = int(rng.integers(256)); r = min(256, l + int(rng.integers(1, 65)))
        cubes[q] = L.Cube(s, e, t, bo, l, r)
        offs[q] = total
        total += (e - s) * (bo - t) * (r - l)
        spec.append((s, e, t, bo, l, r))
    handles = (C.c_void_p * nq)(*[c._h for _ in range(nq)])
    out = np.zeros(total, dtype=np.int64)
    ms = C.c_float()
    L.check(L.lib().dcdf_query_fill_window_batch(handles, cubes, C.c_size_t(nq), C.c_void_p(out.ctypes.data),
                                                 C.c_void_p(offs.ctypes.data), C.byref(ms)))
    for q, (s, e, t, bo, l, r) in enumerate(spec):
        n = (e - s) * (bo - t) * (r - l)
        np.testing.assert_array_equal(out[int(offs[q]):int(offs[q]) + n].reshape(e - s, bo - t, r - l), a[s:e, t:bo, l:r])
    lo_v, hi_v = int(a.min()), int(a.max())
    band = max(1, (hi_v - lo_v) // 10)
    lower = np.array([lo_v + int(rng.integers(0, hi_v - lo_v - band)) for _ in range(nq)], dtype=np.int64)
    upper = lower + band
    counts = np.zeros(nq, dtype=np.uint64)
    soffs = np.zeros(nq, dtype=np.uint64)
    cap = total
    res = np.zeros((cap, 3), dtype=np.uint32)
    L.check(L.lib().dcdf_query_search_batch(handles, cubes, C.c_void_p(lower.ctypes.data), C.c_void_p(upper.ctypes.data),
                                            C.c_size_t(nq), C.c_void_p(res.ctypes.data), C.c_size_t(cap),
                                            C.c_void_p(counts.ctypes.data), C.c_void_p(soffs.ctypes.data), C.byref(ms)))
    for q, (s, e, t, bo, l, r) in enumerate(spec):
        got = set(map(tuple, res[int(soffs[q]):int(soffs[q] + counts[q])].tolist()))
        assert got == brute(a, s, e, t, bo, l, r, int(lower[q]), int(upper[q]))


@pytest.mark.parametrize("kind", ["beyond_2_30", "beyond_int32", "int64_extremes"])
def test_wide_value_chunks_take_the_64_bit_walk(dc, kind):
    """The query walks run on 32-bit values only for chunks whose values lie in [-2^30, 2^30) and start from the per-instant
    top table only when its entries fit int32: chunks outside either range take the 64-bit instantiation / the walk from the
    root, with the same results (windows == the array, search == brute force, both against the oracle's decode too)."""
    rng = np.random.default_rng({"beyond_2_30": 1, "beyond_int32": 2, "int64_extremes": 3}[kind])
    side = 64
    a = rng.integers(-500, 500, size=(6, side, side)).astype(np.int64)
    a[2] = a[1]
    a[3] = a[1] + 9
    a[4, :40, :24] = a[1, :40, :24]
    if kind == "beyond_2_30":
        a[:, 5, 7] += 2 ** 30 + 12345          # narrow32 false, the table's entries still fit int32
    elif kind == "beyond_int32":
        a[:, 5, 7] += 2 ** 33
        a[3, 40, 41] = -(2 ** 35)              # no top table for this chunk
    else:
        a[0, 0, 0] = 2 ** 62
        a[5, 63, 63] = -(2 ** 62)
        a[4, 10:20, 10:20] += 2 ** 50
    data = O.chunk_build(a)
    c = dc.Chunk(data)
    assert c.write_to() == data
    np.testing.assert_array_equal(c.fill_window(dc.Cube(0, 6, 0, side, 0, side)), a)
    for _ in range(40):
        s = int(rng.integers(6)); e = min(6, s + int(rng.integers(1, 4)))
        t = int(rng.integers(side)); bo = min(side, t + int(rng.integers(1, 50)))
        l = int(rng.integers(side)); r = min(side, l + int(rng.integers(1, 50)))
        np.testing.assert_array_equal(c.fill_window(dc.Cube(s, e, t, bo, l, r)), a[s:e, t:bo, l:r])
        lo = int(rng.integers(-600, 600)); hi = lo + int(rng.integers(0, 400))
        if rng.random() < 0.3:
            hi = 2 ** 62
        got = set(map(tuple, c.iter_search(dc.Cube(s, e, t, bo, l, r), lo, hi).tolist()))
        assert got == brute(a, s, e, t, bo, l, r, lo, hi)


def test_windows_larger_than_one_piece(dc):
    """Windows beyond 64 x 64 cells are cut into pieces from their origin (one wave item each); search results of the pieces
    are stitched in (instant, row, col) order."""
    from dcdf_amd import synth
    a = synth.cells(0xDCDF0007, 0, 5, 0, 256, 0, 256, np.int32)
    c = dc.Chunk.build(a).data
    rng = np.random.default_rng(11)
    for (t, bo, l, r) in [(0, 256, 0, 256), (3, 203, 17, 207), (60, 125, 1, 256), (0, 65, 190, 255), (100, 229, 64, 129)]:
        np.testing.assert_array_equal(c.fill_window(dc.Cube(1, 4, t, bo, l, r)), a[1:4, t:bo, l:r])
        lo = int(rng.integers(int(a.min()), int(a.max()) - 300)); hi = lo + 300
        got = c.iter_search(dc.Cube(1, 4, t, bo, l, r), lo, hi)
        want = sorted(brute(a, 1, 4, t, bo, l, r, lo, hi))
        assert [tuple(x) for x in got.tolist()] == want  # the reference's order: instant, row, col


def test_suggest_fraction_golden_and_random(dc):  # fixed.rs:96-159, tests fixed.rs:311-401
    import json, os
    G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))

    def f(x):
        return float("nan") if x == "nan" else x

    for data, ft, rnd, bits in G["suggest_fraction"]:
        dt = np.float32 if ft == "f32" else np.float64
        if data == "fixed_array":
            src = np.array([[[f(v) for v in row] for row in inst] for inst in G["fixed_array"]], dtype=np.float32)
            arr = np.stack([src[i % 3] for i in range(100)])
        else:
            arr = np.array([f(x) for x in data], dtype=dt).reshape(1, 1, -1)
        assert dc.suggest_fraction(arr) == ("round" if rnd else "precise", bits), (data, ft)
    rng = np.random.default_rng(3)
    for dt, ft in ((np.float32, "f32"), (np.float64, "f64")):
        for bits in (0, 1, 5, 11):
            a = (rng.integers(-5000, 5000, size=(7, 33, 20)) / float(1 << bits)).astype(dt)
            a[rng.random(a.shape) < 0.05] = np.nan
            want = O.suggest_fraction(a, ft)
            assert dc.suggest_fraction(a) == ("round" if want[0] else "precise", want[1])
            v = a[:, ::2, 1:15:3]  # strided view
            want = O.suggest_fraction(np.ascontiguousarray(v), ft)
            assert dc.suggest_fraction(v) == ("round" if want[0] else "precise", want[1])
    a = np.full((2, 4, 4), 0.1, dtype=np.float64)
    a[0, 0, 0] = 316.0  # fixed.rs:358-371: loss of precision -> Round(53)
    assert dc.suggest_fraction(a) == ("round", 53)
    assert dc.suggest_fraction(np.full((1, 2, 2), np.nan, dtype=np.float32)) == ("precise", 0)
    assert dc.suggest_fraction(-np.ones((1, 2, 2), dtype=np.float32)) == (lambda w: ("round" if w[0] else "precise", w[1]))(
        O.suggest_fraction(-np.ones((1, 2, 2), dtype=np.float32), "f32"))
    with pytest.raises(dc.DcdfError):
        dc.suggest_fraction(np.full((1, 1, 2), 1e300, dtype=np.float64))  # whole part needs more than 62 bits


def test_malformed_chunks_are_rejected_at_open(dc):
    """Chunks arrive by CID from an untrusted store: dcdf_chunk_open validates structure (sidelen a power of k, rank-index
    stride 4, every Dac level's length == popcount of the level below, Lmax / Lmin / eqB lengths vs. popcount(T)) and returns
    DCDF_ERR_FORMAT; a corrupted chunk never reaches the kernels."""
    from dcdf_amd import synth
    a = synth.cells(77, 0, 5, 0, 32, 0, 32, np.int32)
    good = bytearray(dc.Chunk.build(a).data.write_to())
    dc.Chunk(bytes(good)).close()  # sanity: the intact chunk opens
    rng = np.random.default_rng(11)
    rejected = opened = 0
    cases = []
    for cut in (5, 6, 7, 20, len(good) - 1, len(good) // 2):  # truncations
        cases.append(bytes(good[:cut]))
    t = bytearray(good); t[7] = 3; cases.append(bytes(t))          # k = 3 with sidelen 32: not a power of k
    t = bytearray(good); t[16:20] = (64).to_bytes(4, "big"); cases.append(bytes(t))   # sidelen 64 for a 32 x 32 tile
    t = bytearray(good); t[24:28] = (8).to_bytes(4, "big"); cases.append(bytes(t))    # T bitmap's rank-index stride
    t = bytearray(good); t[20:24] = (len(good) * 8).to_bytes(4, "big"); cases.append(bytes(t))  # absurd T length
    for _ in range(300):  # random byte flips anywhere
        t = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            t[int(rng.integers(0, len(t)))] ^= int(rng.integers(1, 256))
        cases.append(bytes(t))
    for data in cases:
        try:
            c = dc.Chunk(data)
        except dc.DcdfError as e:
            assert e.code in (-7, -1), e.code
            rejected += 1
            continue
        opened += 1  # a flip inside value bytes leaves a well-formed chunk: queries must run without faulting
        shp = c.shape()
        w = c.fill_window(dc.Cube(0, shp[0], 0, shp[1], 0, shp[2]), dtype=np.int64)
        assert w.shape == tuple(shp)
        c.iter_search(dc.Cube(0, shp[0], 0, shp[1], 0, shp[2]), -5, 5)
        c.close()
    assert rejected >= 10 and opened > 0


def test_malformed_chunks_are_rejected_at_device_open(dc):
    """dcdf_chunk_open_batch(DCDF_MEM_DEVICE) parses AND validates on the device (k_validate_insts: a wave per instant runs the
    host entry point's checks): the same corrupted streams as above, uploaded, are accepted or refused exactly as
    dcdf_chunk_open accepts or refuses them, chunk by chunk, and the intact neighbours in the same batch open and answer."""
    import ctypes as C
    from dcdf_amd import synth, _lib as L
    from dcdf_amd.encoder import DeviceBuffer
    a = synth.cells(77, 0, 5, 0, 32, 0, 32, np.int32)
    good = bytearray(dc.Chunk.build(a).data.write_to())
    rng = np.random.default_rng(12)
    cases = [bytes(good)]
    t = bytearray(good); t[28 + 4 * (int.from_bytes(good[20:24], "big") // 128)] ^= 0x80; cases.append(bytes(t))  # T[0] flipped
    t = bytearray(good); t[24:28] = (8).to_bytes(4, "big"); cases.append(bytes(t))    # T bitmap's rank-index stride
    t = bytearray(good); t[16:20] = (64).to_bytes(4, "big"); cases.append(bytes(t))   # sidelen 64 for a 32 x 32 tile
    for _ in range(120):  # random byte flips anywhere
        t = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            t[int(rng.integers(0, len(t)))] ^= int(rng.integers(1, 256))
        cases.append(bytes(t))
    cases.append(bytes(good))
    expect = []
    for data in cases:  # what the host entry point says
        try:
            dc.Chunk(data).close()
            expect.append(0)
        except dc.DcdfError as e:
            expect.append(e.code)
    n = len(cases)
    offs = np.cumsum([0] + [(len(c) + 255) & ~255 for c in cases])
    buf = DeviceBuffer(int(offs[-1]) + 256)
    for c, o in zip(cases, offs):
        buf.write(int(o), np.frombuffer(c, dtype=np.uint8))
    ptrs = (C.c_void_p * n)(*[buf.ptr + int(o) for o in offs[:-1]])
    lens = (C.c_uint64 * n)(*[len(c) for c in cases])
    hs = (C.c_void_p * n)()
    st = (C.c_int32 * n)()
    L.check(L.lib().dcdf_chunk_open_batch(ptrs, lens, C.c_size_t(n), L.MEM_DEVICE, hs, st), "chunk_open_batch")
    got = [int(x) for x in st]
    assert [g != 0 for g in got] == [e != 0 for e in expect], [(i, g, e) for i, (g, e) in enumerate(zip(got, expect)) if (g != 0) != (e != 0)]
    assert got[0] == 0 and got[-1] == 0 and got[1] != 0 and got[2] != 0 and got[3] != 0 and sum(g != 0 for g in got) >= 10
    cube = (C.c_uint32 * 6)(0, 5, 0, 32, 0, 32)
    for i in range(n):
        if got[i] == 0:  # a flip inside value bytes leaves a well-formed chunk: the walk must run without faulting
            assert hs[i]
            out = np.zeros((5, 32, 32), dtype=np.int64)
            L.check(L.lib().dcdf_chunk_fill_window(C.c_void_p(hs[i]), cube, C.c_void_p(out.ctypes.data), L.DCDF_I64, C.c_int64(32 * 32), C.c_int64(32),
                                                   C.c_int64(1)), "fill_window")
            if i in (0, n - 1):
                np.testing.assert_array_equal(out, a)
            L.lib().dcdf_chunk_close(C.c_void_p(hs[i]))
        else:
            assert not hs[i]
    buf.free()


def test_batched_points_typed_windows_and_device_open(dc):
    """The query entry points that do not defeat the kernels (include/dcdf_k2r.h): chunks opened straight from an encoder
    session's device buffers (dcdf_chunk_open_batch: parsed on the device, all side-16 tables in one launch), many gets /
    cell series in one launch, typed and device-resident window results.  Everything against the oracle's Chunk and the
    raw cells."""
    import ctypes as C
    from dcdf_amd import synth, _lib as L
    from dcdf_amd.encoder import DeviceBuffer, Encoder
    arrays = [synth.cells(0xDCDF0005 + i, 0, T, 0, r, 0, c, np.int32) for i, (T, r, c) in enumerate([(9, 256, 256), (5, 200, 256), (7, 64, 64), (3, 16, 16)])]
    arrays.append(np.zeros((4, 64, 64), dtype=np.int32) + 5)                      # single-node instants (no table walk)
    arrays.append((synth.cells(0xDCDF0007, 0, 4, 0, 64, 0, 64, np.int64)))         # int64 chunk
    bufs, descs = [], []
    for a in arrays:
        b = DeviceBuffer(a.nbytes)
        b.write(0, a)
        bufs.append(b)
        descs.append((b.ptr, L.DCDF_I32 if a.dtype == np.int32 else L.DCDF_I64, tuple(s // a.itemsize for s in a.strides), a.shape))
    enc = Encoder(descs, k=2)
    enc.run()
    chunks = enc.open_chunks()
    refs = [O.chunk_build(a) for a in arrays]
    rng = np.random.default_rng(11)
    for c, a, ref in zip(chunks, arrays, refs):
        assert c.shape() == list(a.shape) and c.write_to() == ref
        np.testing.assert_array_equal(c.fill_window(dc.Cube(0, a.shape[0], 0, a.shape[1], 0, a.shape[2])), a)
        lo, hi = int(np.percentile(a, 30)), int(np.percentile(a, 45))
        got = set(map(tuple, c.iter_search(dc.Cube(0, a.shape[0], 1, a.shape[1], 0, a.shape[2] - 1), lo, hi).tolist()))
        want = set(map(tuple, O.Chunk(ref).search(0, a.shape[0], 1, a.shape[1], 0, a.shape[2] - 1, lo, hi).tolist()))
        assert got == want
    # many points over many chunks, one launch
    which = rng.integers(0, len(arrays), size=500)
    pts = np.array([[rng.integers(0, arrays[w].shape[0]), rng.integers(0, arrays[w].shape[1]), rng.integers(0, arrays[w].shape[2])] for w in which])
    got = dc.get_batch([chunks[w] for w in which], pts)
    want = np.array([int(arrays[w][t, r, c]) for w, (t, r, c) in zip(which, pts)])
    np.testing.assert_array_equal(got, want)
    assert chunks[0].get(3, 100, 7) == int(arrays[0][3, 100, 7])                   # single get: pinned page, no allocation
    cells = [(0, arrays[w].shape[0], int(r), int(c)) for w, (_, r, c) in zip(which[:40], pts[:40])]
    cells[3] = (cells[3][1], 1, cells[3][2], cells[3][3])                          # reversed bounds are swapped (chunk.rs:135)
    series = dc.fill_cell_batch([chunks[w] for w in which[:40]], cells)
    for w, (a0, a1, r, c), sr in zip(which[:40], cells, series):
        lo_, hi_ = min(a0, a1), max(a0, a1)
        np.testing.assert_array_equal(sr, arrays[w][lo_:hi_, r, c])
    # typed windows: int32 result of int32 chunks, float64 through from_fixed of an int chunk is not meaningful -> int only
    cubes = [dc.Cube(1, 4, 10, 90, 5, 70), dc.Cube(0, 5, 0, 200, 100, 256), dc.Cube(2, 3, 60, 64, 0, 64)]
    flat, off = dc.fill_window_batch([chunks[0], chunks[1], chunks[2]], cubes, dtype=np.int32)
    assert flat.dtype == np.int32
    for q, (w, cu) in enumerate(zip([0, 1, 2], cubes)):
        exp = arrays[w][cu.start:cu.end, cu.top:cu.bottom, cu.left:cu.right]
        np.testing.assert_array_equal(flat[int(off[q]):int(off[q]) + exp.size].reshape(exp.shape), exp)
    # the same windows decoded straight into a caller's DEVICE array (nothing crosses PCIe until we look)
    vol = [c.instants() * c.rows() * c.cols() for c in cubes]
    doff = np.array([7, 7 + vol[0] + 13, 7 + vol[0] + 13 + vol[1]], dtype=np.uint64)     # arbitrary element offsets
    dev = DeviceBuffer(int(doff[2] + vol[2]) * 4)
    dev.write(0, np.full(int(doff[2] + vol[2]), -1, dtype=np.int32))
    dc.fill_window_batch([chunks[0], chunks[1], chunks[2]], cubes, dtype=np.int32, out_device_ptr=dev.ptr, out_offset=doff)
    back = dev.read(0, int(doff[2] + vol[2]) * 4, np.int32)
    assert (back[:7] == -1).all() and (back[7 + vol[0]:7 + vol[0] + 13] == -1).all()
    for q, (w, cu) in enumerate(zip([0, 1, 2], cubes)):
        exp = arrays[w][cu.start:cu.end, cu.top:cu.bottom, cu.left:cu.right]
        np.testing.assert_array_equal(back[int(doff[q]):int(doff[q]) + exp.size].reshape(exp.shape), exp)
    for c in chunks:
        c.close()
    enc.close()


def test_open_batch_rejects_garbage_in_device_memory(dc):
    from dcdf_amd import _lib as L
    from dcdf_amd.encoder import DeviceBuffer
    import ctypes as C
    good = O.chunk_build(np.arange(2 * 16 * 16, dtype=np.int32).reshape(2, 16, 16))
    blobs = [good, good[:-3], bytes(40), good[:6] + bytes([9]) + good[7:]]
    bufs = []
    ptrs = (C.c_void_p * len(blobs))()
    lens = (C.c_uint64 * len(blobs))()
    for i, bts in enumerate(blobs):
        b = DeviceBuffer(len(bts) + 64)
        b.write(0, np.frombuffer(bts, dtype=np.uint8))
        bufs.append(b)
        ptrs[i], lens[i] = b.ptr, len(bts)
    hs = (C.c_void_p * len(blobs))()
    st = (C.c_int32 * len(blobs))()
    L.check(L.lib().dcdf_chunk_open_batch(ptrs, lens, C.c_size_t(len(blobs)), L.MEM_DEVICE, hs, st))
    assert list(st) == [0, -7, -7, -7] and hs[0] and not hs[1] and not hs[2] and not hs[3]
    L.lib().dcdf_chunk_close(C.c_void_p(hs[0]))
