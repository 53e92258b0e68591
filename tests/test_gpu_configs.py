"""BASELINE.json configs exercised at full size on the GPU (pytest -m gpu):

  configs[2]  every one of the 3072 chunks of the 4096x4096x365 raster, bytes == oracle (oracle on all host cores)
  configs[4]  window + search queries on a multi-tile, multi-segment encoded raster, split at tile/segment boundaries
              as superchunk.rs:589-633 / span.rs:190-216 do, against the raw raster AND the oracle's Chunk queries
  plus the session paths the advisor flagged: the ST_OUT_CAPACITY retry and non-contiguous fill_window_batch offsets.
"""
import ctypes as C
import hashlib
import json
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def dc():
    import dcdf_amd
    from dcdf_amd import _lib
    assert _lib.lib().dcdf_device_name(), "no GPU"
    return dcdf_amd


def _device_raster(seed, shape, origin=(0, 0, 0), tile=256, chunk_size=32, dtype_code=4):
    """The synthetic raster cut into chunks, generated in HBM: returns (DeviceBuffer, grid, descs, offsets in elements)."""
    from dcdf_amd.encoder import DeviceBuffer, synth_fill
    from dcdf_amd.raster import EncodedRaster
    grid = EncodedRaster.chunk_grid(shape, tile, chunk_size)
    esz = 4 if dtype_code == 4 else 8
    sizes = [(t1 - t0) * (r1 - r0) * (c1 - c0) for t0, t1, r0, r1, c0, c1 in grid]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    buf = DeviceBuffer(int(offs[-1]) * esz)
    descs = []
    for (t0, t1, r0, r1, c0, c1), o in zip(grid, offs):
        ptr = buf.ptr + int(o) * esz
        synth_fill(ptr, dtype_code, seed, origin[0] + t0, origin[0] + t1, origin[1] + r0, origin[1] + r1, origin[2] + c0, origin[2] + c1)
        descs.append((ptr, dtype_code, ((r1 - r0) * (c1 - c0), c1 - c0, 1), (t1 - t0, r1 - r0, c1 - c0)))
    return buf, grid, descs, offs


def test_config2_full_raster_every_chunk_vs_oracle(dc):
    """BASELINE configs[2] / SURVEY 8(d) config 3: the whole 4096x4096x365 int32 raster (seed 0xDCDF0003), 3072 chunks,
    encoded in one launch; every chunk's Chunk::write_to image compared with the CPU oracle (one chunk per task on the
    host cores), snapshots/logs counts too; SHA-256 of the concatenation in chunk order recorded and, when
    tests/golden/config2_sha256.json is present, compared with it."""
    from dcdf_amd.encoder import Encoder
    shape = (365, 4096, 4096)
    buf, grid, descs, offs = _device_raster(0xDCDF0003, shape)
    assert len(grid) == 3072
    enc = Encoder(descs, k=2)
    enc.run()
    packed, goffs, glens, mm = enc.gather()
    res = [enc.result(i) for i in range(len(grid))]
    assert all(r[0] == 0 for r in res)
    mmoff = np.concatenate([[0], np.cumsum([g[1] - g[0] for g in grid])])

    def check(i):
        t0, t1, r0, r1, c0, c1 = grid[i]
        host = buf.read(int(offs[i]) * 4, (t1 - t0) * 65536 * 4, np.int32).reshape(t1 - t0, 256, 256)
        ref, ns, nl, _ = O.chunk_build(host, want_snapshots=True)
        got = packed[int(goffs[i]):int(goffs[i]) + int(glens[i])]
        ok = len(ref) == int(glens[i]) and got.tobytes() == ref and (ns, nl) == (res[i][2], res[i][3])
        flat = host.reshape(t1 - t0, -1)
        m = mm[int(mmoff[i]):int(mmoff[i + 1])]
        ok = ok and (m[:, 0] == flat.min(1)).all() and (m[:, 1] == flat.max(1)).all()
        return ok, hashlib.sha256(ref).digest()

    nthr = max(1, min(16, len(os.sched_getaffinity(0))))
    with ThreadPoolExecutor(nthr) as ex:
        out = list(ex.map(check, range(len(grid))))
    bad = [i for i, (ok, _) in enumerate(out) if not ok]
    assert not bad, "chunks differing from the oracle: %s" % bad[:10]
    h = hashlib.sha256()
    for i in range(len(grid)):
        h.update(memoryview(packed[int(goffs[i]):int(goffs[i]) + int(glens[i])]))
    sha = h.hexdigest()
    rec = {"workload": "configs[2]: 4096x4096x365 int32, seed 0xDCDF0003, 3072 chunks", "chunks_compared": len(grid),
           "chunks_differing": 0, "encoded_bytes": int(glens.sum()), "sha256_of_concatenation_in_chunk_order": sha,
           "snapshots": sum(r[2] for r in res), "logs": sum(r[3] for r in res)}
    os.makedirs(os.path.join(os.path.dirname(HERE), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(HERE), "gpurun_out", "config2_parity.json"), "w") as f:
        json.dump(rec, f)
    print(json.dumps(rec))
    gold = os.path.join(HERE, "golden", "config2_sha256.json")
    if os.path.exists(gold):
        assert json.load(open(gold))["sha256_of_concatenation_in_chunk_order"] == sha
    enc.close()
    buf.free()


@pytest.mark.parametrize("dtype", [np.int32, np.int64])
def test_config1_all_1024_chunks_vs_oracle(dc, dtype):
    """BASELINE configs[1] / SURVEY 8(d) config 2: 1024 independent [32, 256, 256] chunks, chunk c from seed 0xDCDF0002 + c, as
    int32 and as int64 "fixed point" (odd values, 16 GiB), one launch each; EVERY chunk's bytes, counters and per-instant
    (min, max) against the CPU oracle (one chunk per task on the host cores)."""
    from dcdf_amd import _lib as L
    from dcdf_amd.encoder import DeviceBuffer, Encoder, synth_fill
    n, T, S = 1024, 32, 256
    es = np.dtype(dtype).itemsize
    code = L.DCDF_I32 if dtype == np.int32 else L.DCDF_I64
    buf = DeviceBuffer(n * T * S * S * es)
    descs = []
    for c in range(n):
        ptr = buf.ptr + c * T * S * S * es
        synth_fill(ptr, code, 0xDCDF0002 + c, 0, T, 0, S, 0, S)
        descs.append((ptr, code, (S * S, S, 1), (T, S, S)))
    enc = Encoder(descs, k=2)
    enc.run()
    packed, goffs, glens, mm = enc.gather()
    res = [enc.result(i) for i in range(n)]
    assert all(r[0] == 0 for r in res)

    def check(i):
        host = buf.read(i * T * S * S * es, T * S * S * es, dtype).reshape(T, S, S)
        ref, ns, nl, _ = O.chunk_build(host, want_snapshots=True)
        ok = len(ref) == int(glens[i]) and packed[int(goffs[i]):int(goffs[i]) + int(glens[i])].tobytes() == ref and (ns, nl) == (res[i][2], res[i][3])
        flat = host.reshape(T, -1)
        m = mm[i * T:(i + 1) * T]
        return ok and (m[:, 0] == flat.min(1)).all() and (m[:, 1] == flat.max(1)).all()

    nthr = max(1, min(16, len(os.sched_getaffinity(0))))
    with ThreadPoolExecutor(nthr) as ex:
        out = list(ex.map(check, range(n)))
    bad = [i for i, ok in enumerate(out) if not ok]
    assert not bad, "chunks differing from the oracle: %s" % bad[:10]
    enc.close()
    buf.free()


def test_config4_queries_split_at_tile_and_segment_boundaries(dc):
    """BASELINE configs[4] / SURVEY 8(d) config 5 on a sub-raster that has every kind of boundary: 70 x 768 x 768 cells of
    the configs[2] raster (3 time segments of 32/32/6 instants x 3 x 3 tiles = 27 chunks).  10 000 fill_window + 10 000
    search_window cubes (t0 in U, len_t in U[1,8], h, w in U[1,64], clipped), each split where Span / Superchunk would
    split it; answers checked against the raw raster and, for a sample, against the oracle's Chunk queries piece by piece."""
    from dcdf_amd import synth
    from dcdf_amd.encoder import Encoder
    from dcdf_amd.raster import EncodedRaster
    shape, origin = (70, 768, 768), (300, 1024, 2048)  # instants 300..369 wrap nothing: raster coordinates are global
    buf, grid, descs, offs = _device_raster(0xDCDF0003, shape, origin)
    assert len(grid) == 27
    enc = Encoder(descs, k=2)
    enc.run()
    data = [enc.fetch(i) for i in range(len(grid))]
    enc.close()
    buf.free()
    raw = synth.cells(0xDCDF0003, origin[0], origin[0] + shape[0], origin[1], origin[1] + shape[1], origin[2], origin[2] + shape[2], np.int32)
    er = EncodedRaster(shape, [dc.Chunk(d, lazy=True) for d in data])
    rng = np.random.default_rng(0xDCDF0005)

    def cubes(n):
        t0 = rng.integers(0, shape[0], n)
        t1 = np.minimum(shape[0], t0 + rng.integers(1, 9, n))
        r0 = rng.integers(0, shape[1], n)
        r1 = np.minimum(shape[1], r0 + rng.integers(1, 65, n))
        c0 = rng.integers(0, shape[2], n)
        c1 = np.minimum(shape[2], c0 + rng.integers(1, 65, n))
        return np.stack([t0, t1, r0, r1, c0, c1], axis=1)

    n = 10000
    # ---- fill_window
    q = cubes(n)
    # force the corner cases in: a cube spanning all three segments and all nine tiles, and 1-cell cubes on the seams
    q[0] = (0, 70, 200, 600, 250, 700)
    q[1] = (31, 33, 255, 257, 511, 513)
    q[2] = (64, 65, 767, 768, 0, 1)
    wins = er.fill_windows(q)
    nsplit = len(er.split(q))
    assert nsplit > n  # boundaries were crossed
    for c, w in zip(q, wins):
        assert (w == raw[c[0]:c[1], c[2]:c[3], c[4]:c[5]]).all()
    # ---- search_window: a random 10-percentile-wide band of the value range
    edges = np.percentile(raw[::7, ::5, ::5], np.arange(0, 101, 10)).astype(np.int64)
    q = cubes(n)
    q[0] = (0, 70, 200, 600, 250, 700)
    q[1] = (31, 33, 255, 257, 511, 513)
    band = rng.integers(0, 10, n)
    lo, hi = edges[band], edges[band + 1]
    hits = er.search(q, lo, hi)
    for c, l, h, got in zip(q, lo, hi, hits):
        sub = raw[c[0]:c[1], c[2]:c[3], c[4]:c[5]]
        want = np.argwhere((sub >= l) & (sub <= h)) + np.array([c[0], c[2], c[4]])
        assert len(got) == len(want)
        if len(want):
            assert (np.array(sorted(map(tuple, got.tolist()))) == want).all()
    # ---- the same pieces through the oracle's Chunk queries (chunk-level parity of the split sub-queries)
    och = {}
    sub, trip, soff, counts, _ = er.search_pieces(q[:300], lo[:300], hi[:300])
    wsub, wout, woff, vol, _ = er.window_pieces(q[:300])
    assert (wsub == sub).all()
    for k in range(len(sub)):
        qi, cid, a0, a1, b0, b1, d0, d1 = (int(x) for x in sub[k])
        oc = och.setdefault(cid, O.Chunk(data[cid]))
        want = oc.search(a0, a1, b0, b1, d0, d1, int(lo[qi]), int(hi[qi]))
        got = trip[int(soff[k]):int(soff[k]) + int(counts[k])]
        assert set(map(tuple, got.tolist())) == set(map(tuple, want.tolist()))
        ow = oc.fill_window(a0, a1, b0, b1, d0, d1)
        assert (wout[int(woff[k]):int(woff[k]) + int(vol[k])].reshape(ow.shape) == ow).all()
    for c in er.chunks:
        c.close()


def test_fill_window_batch_noncontiguous_offsets_touch_only_their_windows(dc):
    """include/dcdf_k2r.h: query q writes its window at out + out_offset[q] and nothing else."""
    from dcdf_amd import synth, _lib as L
    a = synth.cells(5, 0, 6, 0, 64, 0, 64, np.int32)
    c = dc.Chunk.build(a).data
    spec = [(0, 2, 3, 20, 5, 9), (4, 6, 0, 64, 0, 64), (1, 2, 63, 64, 63, 64), (2, 5, 10, 11, 0, 33)]
    vols = [(e - s) * (b - t) * (r - l) for s, e, t, b, l, r in spec]
    offs = np.array([7, 20000, 3, 9000], dtype=np.uint64)  # out of order, with gaps
    SENT = -0x5a5a5a5a5a5a5a5b
    out = np.full(20000 + vols[1] + 11, SENT, dtype=np.int64)
    cubes = (L.Cube * 4)(*[L.Cube(*s) for s in spec])
    handles = (C.c_void_p * 4)(*[c._h] * 4)
    ms = C.c_float()
    L.check(L.lib().dcdf_query_fill_window_batch(handles, cubes, C.c_size_t(4), C.c_void_p(out.ctypes.data),
                                                 C.c_void_p(offs.ctypes.data), C.byref(ms)))
    mask = np.zeros(out.size, dtype=bool)
    for (s, e, t, b, l, r), o, v in zip(spec, offs, vols):
        np.testing.assert_array_equal(out[int(o):int(o) + v].reshape(e - s, b - t, r - l), a[s:e, t:b, l:r])
        mask[int(o):int(o) + v] = True
    assert (out[~mask] == SENT).all()
    c.close()


def test_out_capacity_retry_and_session_reuse(dc):
    """k2r_capi_encode.hip: tiles whose output slot is too small are re-encoded into worst-case slots; fetch, gather,
    object_sha256 and a second run() must then still be right.  An iid-noise tile also overflows the DEFAULT slot."""
    from dcdf_amd import synth, _lib as L
    from dcdf_amd.encoder import DeviceBuffer, Encoder
    rng = np.random.default_rng(4)
    arrays = [synth.cells(31, 0, 5, 0, 64, 0, 64, np.int32), np.zeros((3, 16, 16), dtype=np.int32) + 7,
              rng.integers(-(2 ** 29), 2 ** 29, size=(4, 64, 64)).astype(np.int32), synth.cells(32, 0, 3, 0, 128, 0, 128, np.int32)]
    refs = [O.chunk_build(a) for a in arrays]
    hdr = bytes([0xDC, 0xE0, 0, 0, 0, 1, 2, 4])
    for cap in (256, 0):
        bufs = []
        descs = []
        for a in arrays:
            b = DeviceBuffer(a.nbytes)
            b.write(0, a)
            bufs.append(b)
            descs.append((b.ptr, L.DCDF_I32, tuple(s // 4 for s in a.strides), a.shape))
        enc = Encoder(descs, k=2, out_cap_per_tile=cap)
        for _ in range(2):
            enc.run()
            for i, ref in enumerate(refs):
                assert enc.fetch(i) == ref, (cap, i)
        packed, goffs, glens, mm = enc.gather()
        for i, ref in enumerate(refs):
            assert packed[int(goffs[i]):int(goffs[i]) + int(glens[i])].tobytes() == ref
        dig, _ = enc.object_sha256()
        for i, ref in enumerate(refs):
            assert dig[i].tobytes() == hashlib.sha256(hdr + ref).digest()
        if cap == 0:
            assert len(refs[2]) > arrays[2].size * 4 + 4096  # the noise tile really exceeded the default slot
        enc.close()
        for b in bufs:
            b.free()


def test_config4_one_million_queries_full_size(dc):
    """BASELINE configs[4] at full size: 1M random get_window + search_window queries against the encoded 4096x4096x365 raster
    (tools/bench_query.py: chunks opened from the encoder's device buffers, typed device-resident results; once through the chunk-level batch entry points with the routing in numpy and once through dcdf_raster_*), 80 reassembled
    answers per kind checked against the synthetic model cell by cell / hit by hit, and the totals against the size-independent
    facts of the workload (every query is answered; decoded cells == the cubes' volumes)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_query", os.path.join(os.path.dirname(HERE), "tools", "bench_query.py"))
    bq = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bq)
    res = bq.run(queries=1000000, batch=250000, segments=12, check=20, cpu_sample=0, host_results=False)
    assert res["config"]["chunks"] == 3072 and res["config"]["answers_checked_vs_model"] == 320
    assert res["fill_window"]["queries"] == 500000 and res["search_window"]["queries"] == 500000
    assert res["fill_window"]["cells"] > 2e9 and res["search_window"]["hits"] > 1e8
    assert res["open"]["seconds"] < 5.0


@pytest.mark.parametrize("host_built", [False, True])
def test_native_raster_routing_matches_the_python_routing(dc, monkeypatch, host_built):
    """dcdf_raster_fill_window_batch / dcdf_raster_search_batch (the split of Span::fill_window, span.rs:190-216, and
    Superchunk::subchunks_for, superchunk.rs:589-633, in C++; every piece decoded straight into its window) against the raw
    raster and against the Python routing over the per-chunk batch entry points."""
    from dcdf_amd import synth
    from dcdf_amd.encoder import DeviceBuffer, Encoder
    from dcdf_amd.raster import EncodedRaster
    if host_built:  # the search pieces / items built on the host and uploaded (the form other arities than k = 2 take)
        monkeypatch.setenv("K2R_RASTER_HOST", "1")
    shape = (70, 600, 520)  # ragged last segment (70 = 2 * 32 + 6), ragged tiles (600 = 2 * 256 + 88, 520 = 2 * 256 + 8)
    buf, grid, descs, offs = _device_raster(0xDCDF0009, shape)
    enc = Encoder(descs, k=2)
    enc.run()
    chunks = enc.open_chunks()
    R = EncodedRaster(shape, chunks)
    full = synth.cells(0xDCDF0009, 0, shape[0], 0, shape[1], 0, shape[2], np.int32)
    rng = np.random.default_rng(9)
    cubes = []
    for _ in range(300):
        t0 = int(rng.integers(0, shape[0])); t1 = min(shape[0], t0 + int(rng.integers(1, 40)))
        r0 = int(rng.integers(0, shape[1])); r1 = min(shape[1], r0 + int(rng.integers(1, 300)))
        c0 = int(rng.integers(0, shape[2])); c1 = min(shape[2], c0 + int(rng.integers(1, 300)))
        cubes.append((t0, t1, r0, r1, c0, c1))
    cubes.append((0, shape[0], 250, 262, 250, 262))     # crosses every segment and four tiles
    cubes.append((31, 33, 0, shape[1], 255, 257))
    cubes.append((5, 5, 10, 20, 10, 20))                 # empty
    cubes.append((0, 3, 100, 100, 0, 50))                # empty
    flat, off, _ = R.fill_windows_flat(cubes, dtype=np.int32)
    for q, c in enumerate(cubes):
        exp = full[c[0]:c[1], c[2]:c[3], c[4]:c[5]]
        np.testing.assert_array_equal(flat[int(off[q]):int(off[q]) + exp.size].reshape(exp.shape), exp)
    # device-resident result at arbitrary element offsets
    vol = [(c[1] - c[0]) * (c[3] - c[2]) * (c[5] - c[4]) for c in cubes[:20]]
    doff = np.cumsum([5] + [v + 3 for v in vol[:-1]]).astype(np.uint64)
    dev = DeviceBuffer(int(doff[-1] + vol[-1]) * 8)
    R.fill_windows_flat(cubes[:20], dtype=np.int64, out_device_ptr=dev.ptr, out_offset=doff)
    back = dev.read(0, int(doff[-1] + vol[-1]) * 8, np.int64)
    for q, c in enumerate(cubes[:20]):
        exp = full[c[0]:c[1], c[2]:c[3], c[4]:c[5]]
        np.testing.assert_array_equal(back[int(doff[q]):int(doff[q]) + exp.size].reshape(exp.shape), exp)
    # search: raster coordinates
    lo = rng.integers(-3000, 3000, size=len(cubes))
    hi = lo + rng.integers(0, 400, size=len(cubes))
    trip, soff, cnt, _ = R.search_flat(cubes, lo, hi)
    ne = [q for q, c in enumerate(cubes) if (c[1] - c[0]) * (c[3] - c[2]) * (c[5] - c[4]) > 0]
    ref = dict(zip(ne, R.search(np.array(cubes)[ne], lo[ne], hi[ne])))  # the Python routing over dcdf_query_search_batch
    assert int(cnt.sum()) > 1000
    for q, c in enumerate(cubes):
        got = trip[int(soff[q]):int(soff[q]) + int(cnt[q])]
        got = set(map(tuple, got.tolist()))
        assert len(got) == int(cnt[q])
        sub = full[c[0]:c[1], c[2]:c[3], c[4]:c[5]]
        want = set((int(a) + c[0], int(b) + c[2], int(d) + c[4]) for a, b, d in zip(*np.nonzero((sub >= lo[q]) & (sub <= hi[q]))))
        assert got == want, q
        if q in ref:
            assert got == set(map(tuple, ref[q].tolist())), q
    R.close()
    for ch in chunks:
        ch.close()
    enc.close()
    buf.free()


def test_raster_entry_points_reject_bad_input(dc):
    """dcdf_raster_*: a chunk table that does not match the grid, cubes beyond the raster, a result buffer too small -- error
    codes, not crashes (the reference panics at mmarray.rs:218-229)."""
    import ctypes as C
    from dcdf_amd import _lib as L, synth
    from dcdf_amd.raster import EncodedRaster
    a = synth.cells(0xDCDF0003, 0, 8, 0, 256, 0, 512, np.int32)
    chunks = [dc.Chunk.build(np.ascontiguousarray(a[:, :, 256 * j:256 * j + 256])).data for j in range(2)]
    hs = (C.c_void_p * 2)(*[c._h for c in chunks])
    h = C.c_void_p()
    shp = (C.c_uint32 * 3)(8, 256, 512)
    assert L.lib().dcdf_raster_create(hs, C.c_size_t(2), shp, 128, 8, C.byref(h)) == -1      # DCDF_ERR_BAD_ARG: 2 x 4 tiles of 128, not 2 chunks
    assert L.lib().dcdf_raster_create(hs, C.c_size_t(2), (C.c_uint32 * 3)(8, 256, 500), 256, 8, C.byref(h)) == -1  # second chunk is 256 wide, not 244
    R = EncodedRaster((8, 256, 512), chunks, tile=256, chunk_size=8)
    good = [(0, 8, 10, 20, 250, 262)]
    flat, off, _ = R.fill_windows_flat(good, dtype=np.int32)
    np.testing.assert_array_equal(flat.reshape(8, 10, 12), a[:, 10:20, 250:262])
    for bad in [(0, 9, 0, 1, 0, 1), (0, 1, 0, 257, 0, 1), (0, 1, 0, 1, 0, 513)]:
        with pytest.raises(L.DcdfError) as e:
            R.fill_windows_flat([bad], dtype=np.int32)
        assert e.value.code == -5  # DCDF_ERR_BOUNDS
        with pytest.raises(L.DcdfError) as e:
            R.search_flat([bad], [0], [1])
        assert e.value.code == -5
    with pytest.raises(L.DcdfError) as e:  # every cell matches, room for ten triples
        R.search_flat(good, [-10 ** 9], [10 ** 9], cap=10)
    assert e.value.code == -11  # DCDF_ERR_CAPACITY
    trip, soff, cnt, _ = R.search_flat([(8, 0, 20, 10, 262, 250)], [10 ** 9], [-10 ** 9])   # reversed bounds everywhere (geom.rs:83-103)
    assert int(cnt[0]) == 8 * 10 * 12
    R.close()
    for c in chunks:
        c.close()


@pytest.mark.parametrize("k,tile,cells", [(3, 81, False), (3, 81, True), (4, 64, False), (2, 300, False)])
def test_raster_entry_points_other_arities_and_tiles(dc, k, tile, cells, monkeypatch):
    """dcdf_raster_* over chunks that do not take the k = 2 node walk (k = 3, 4: the wave kernel on the chunk's 32-grid, search
    by the pruned wave walk `k_search_wave`, or -- `cells`: K2R_SEARCH_CELLS=1 -- by the decode-and-test kernel kept for
    k > 8) and over padded k = 2 tiles wider than one 64-piece: windows and matches against the raw raster."""
    if cells:
        monkeypatch.setenv("K2R_SEARCH_CELLS", "1")
    from dcdf_amd import build_batch
    from dcdf_amd.raster import EncodedRaster
    rng = np.random.default_rng(100 + k)
    shape = (11, 2 * tile + 17, tile + 40)
    cs = 4
    full = (rng.integers(0, 50, size=shape[1:])[None] + (rng.random(shape) < 0.05) * rng.integers(-200, 200, size=shape)).astype(np.int32)
    full[7] = full[6]
    grid = EncodedRaster.chunk_grid(shape, tile, cs)
    built = build_batch([np.ascontiguousarray(full[t0:t1, r0:r1, c0:c1]) for t0, t1, r0, r1, c0, c1 in grid], k=k)
    chunks = [b.data for b in built]
    R = EncodedRaster(shape, chunks, tile=tile, chunk_size=cs)
    cubes = [(0, shape[0], 0, shape[1], 0, shape[2])]
    for _ in range(60):
        t0 = int(rng.integers(0, shape[0])); t1 = min(shape[0], t0 + int(rng.integers(1, 8)))
        r0 = int(rng.integers(0, shape[1])); r1 = min(shape[1], r0 + int(rng.integers(1, 150)))
        c0 = int(rng.integers(0, shape[2])); c1 = min(shape[2], c0 + int(rng.integers(1, 150)))
        cubes.append((t0, t1, r0, r1, c0, c1))
    flat, off, _ = R.fill_windows_flat(cubes, dtype=np.int64)
    for q, c in enumerate(cubes):
        exp = full[c[0]:c[1], c[2]:c[3], c[4]:c[5]]
        np.testing.assert_array_equal(flat[int(off[q]):int(off[q]) + exp.size].reshape(exp.shape), exp)
    lo = rng.integers(-100, 100, size=len(cubes))
    hi = lo + rng.integers(0, 60, size=len(cubes))
    trip, soff, cnt, _ = R.search_flat(cubes, lo, hi)
    for q, c in enumerate(cubes):
        got = set(map(tuple, trip[int(soff[q]):int(soff[q]) + int(cnt[q])].tolist()))
        sub = full[c[0]:c[1], c[2]:c[3], c[4]:c[5]]
        want = set((int(a) + c[0], int(b) + c[2], int(d) + c[4]) for a, b, d in zip(*np.nonzero((sub >= lo[q]) & (sub <= hi[q]))))
        assert len(got) == int(cnt[q]) and got == want, q
    R.close()
    for ch in chunks:
        ch.close()


def test_calls_from_several_host_threads(dc):
    """the entry points may be called from several host threads (ctypes drops the GIL inside a call): four
    threads build chunks through the host-buffer entry point, open them, run windows / searches / points and assemble a
    superchunk at the same time; every result equals the one a single thread gets."""
    from concurrent.futures import ThreadPoolExecutor
    from dcdf_amd import synth, build_batch, Superchunk

    def work(seed):
        a = synth.cells(0xDCDF0100 + seed, 0, 12, 0, 256, 0, 256, np.int32)
        built = build_batch([a, a[:, :100, :77], a.astype(np.int64) * 2 + 1])
        datas = [b.data.write_to() for b in built]
        ch = built[0].data
        win = ch.fill_window(dc.Cube(2, 9, 10, 200, 30, 250))
        hits = sorted(map(tuple, np.asarray(ch.iter_search(dc.Cube(0, 12, 0, 256, 0, 256), 0, 40)).tolist()))
        pts = [ch.get(t, 5 * t, 250 - t) for t in range(12)]
        sc = Superchunk.build(np.ascontiguousarray(np.tile(a, (1, 2, 2))[:8]), [1, 8])
        for b in built:
            b.data.close()
        return datas, win.tobytes(), hits, pts, sorted(sc.objects.items()), sc.size

    seeds = list(range(8))
    ref = [work(s) for s in seeds[:4]]
    with ThreadPoolExecutor(4) as ex:
        got = list(ex.map(work, seeds[:4] * 3))
    for i, g in enumerate(got):
        assert g == ref[i % 4], i
    a0 = synth.cells(0xDCDF0100, 0, 12, 0, 256, 0, 256, np.int32)
    assert ref[0][0][0] == O.chunk_build(a0) and ref[0][1] == np.ascontiguousarray(a0[2:9, 10:200, 30:250]).tobytes()
