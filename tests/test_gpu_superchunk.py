"""Superchunk assembly (dcdf_superchunk_build, superchunk.rs:88-270) on the GPU against the oracle: every stored object --
framed sub-chunks, nested superchunks, Links, the Superchunk node itself -- byte-identical, CIDs equal, same de-duplication,
same MMStruct3Build counters, for the shapes of the reference's own fixtures (superchunk.rs:1006-1188)."""
import json
import os

import numpy as np
import pytest

import oracle_superchunk as OS

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


def array_n(n, T=100):
    a8 = np.array(G["array8"], dtype=np.int64)
    a = np.stack([a8[i % 3] for i in range(T)])
    idx = np.arange(n) % 8
    return np.ascontiguousarray(a[:, idx][:, :, idx])


def check(dc, a, levels, k=2, fractional_bits=0, round=False):
    store = OS.Store()
    obj, st = OS.superchunk_build(a, list(levels), k, store, fractional_bits, round)
    root = store.save(obj)
    b = dc.Superchunk.build(a, levels, k=k, fractional_bits=fractional_bits, round=round)
    assert b.cid == root and b.data == obj
    assert set(b.objects) == set(store), (len(b.objects), len(store))
    for cid, o in store.items():
        assert b.objects[cid] == o
    assert (b.size, b.elided, b.local, b.external, b.snapshots, b.logs) == (st["size"], st["elided"], 0, st["external"], st["snapshots"], st["logs"])
    return st


def test_reference_fixture_shapes(dc):
    assert len(check(dc, array_n(8), [3, 0])["references"]) == 64                      # no_subchunks
    check(dc, array_n(16), [4, 0])                                                      # no_subchunks_four_levels
    a = np.repeat(np.repeat(array_n(8), 2, axis=1), 2, axis=2)
    assert check(dc, a, [3, 1])["elided"] == 64                                         # no_subchunks_coarse
    st = check(dc, array_n(16), [2, 2])                                                 # external_subchunks: 16 refs, 4 links
    assert len(st["links"]) == 4 and st["elided"] == 0
    assert st["size_self"] == 3871  # Superchunk::size() term by term (tests/test_oracle_superchunk.py); check() compared b.size with it
    st = check(dc, array_n(17), [2, 3])                                                 # mixed_subchunks
    assert st["elided"] == 8 and sum(r is not None for r in st["references"]) == 8
    e = np.zeros((100, 16, 16), dtype=np.int64) + np.arange(100)[:, None, None]
    assert check(dc, e, [2, 2])["elided"] == 16                                         # elide_everything
    check(dc, array_n(17), [1, 2, 2])                                                   # nested_superchunks (mmstruct.rs:463-479)
    check(dc, array_n(17).astype(np.int32), [1, 2, 2])


def test_synthetic_and_float_rasters(dc):
    from dcdf_amd import synth
    a = synth.cells(0xDCDF0003, 0, 6, 0, 512, 0, 512, np.int32)                         # 2 x 2 tiles of 256: the fused kernel's tiles
    check(dc, a, [1, 8])
    check(dc, a[:, :300, :400], [1, 8])                                                  # ragged bottom / right tiles
    f = (synth.cells(0xDCDF0004, 0, 4, 0, 128, 0, 128, np.int32) / 8.0).astype(np.float32)
    f[1, 5, 5] = np.nan
    f[2, :64, :64] = np.nan                                                              # a whole tile NaN for one instant
    f[3, 64:, 64:] = 2.5                                                                 # uniform tile in one instant
    check(dc, f, [1, 6], fractional_bits=3)
    g = f.astype(np.float64)
    check(dc, g, [1, 6], fractional_bits=5, round=True)
    day = np.load(os.path.join(HERE, "golden", "cpc_precip_day.npz"))["precip"].reshape(1, 360, 720)
    check(dc, day, [4, 6], fractional_bits=24, round=True)                               # py-dcdf test_dcdf.py:340-365 (k2_levels [4, 6])


def test_wrong_levels_are_rejected(dc):
    with pytest.raises(dc.DcdfError):
        dc.Superchunk.build(array_n(16), [2, 3])  # superchunk.rs:104-110 panics


@pytest.mark.parametrize("band_mb", ["0", "1"])
def test_banded_assembly_of_a_host_view(dc, band_mb, monkeypatch):
    """A host view goes to HBM band of tile rows after band (k2r_superchunk.hip `Upload`), each band's chunks encoded and fetched while
    the next is on the link: K2R_SC_BAND_MB=1 forces one tile row per band on these small rasters, 0 turns the bands off.  Same
    objects either way -- dense, row-cropped (2-D copies) and transposed (gathered) views, ints and floats with an elided band."""
    from dcdf_amd import synth
    monkeypatch.setenv("K2R_SC_BAND_MB", band_mb)
    a = synth.cells(0xDCDF0003, 0, 9, 0, 1024, 0, 768, np.int32)                         # 4 x 4 grid of 256-tiles, the last column outside
    check(dc, a, [2, 8])
    check(dc, a[:, 100:900, 17:600], [2, 8])                                             # row pitch > cols: the 2-D copy per instant
    check(dc, a[:, :520, :300].transpose(0, 2, 1), [2, 8])                               # general strides: gathered rows
    b = a[:5, :700, :600].copy()
    b[:, 256:512, :] = 7                                                                 # a whole band of uniform tiles: elided
    check(dc, b, [2, 8])
    f = (synth.cells(0xDCDF0004, 0, 4, 0, 600, 0, 300, np.int32) / 4.0).astype(np.float32)
    f[1, 300, 5] = np.nan
    check(dc, f, [2, 8], fractional_bits=2)


def test_page_locked_upload_in_uneven_bands(dc, monkeypatch):
    """A host view of 64 MB or more is page-locked for the call (hipHostRegister) and its last band is halved: [8, 2048, 1024] int32 in
    bands of 16 MB against the oracle, and the same objects with the registration switched off and with one band."""
    from dcdf_amd import synth
    a = synth.cells(0xDCDF0003, 0, 8, 0, 2048, 0, 1024, np.int32)
    monkeypatch.setenv("K2R_SC_BAND_MB", "16")
    check(dc, a, [3, 8])
    ref = dc.Superchunk.build(a, [3, 8])
    monkeypatch.setenv("K2R_NO_HOST_REGISTER", "1")
    b = dc.Superchunk.build(a, [3, 8])
    monkeypatch.setenv("K2R_SC_BAND_MB", "0")
    c = dc.Superchunk.build(a, [3, 8])
    for other in (b, c):
        assert other.cid == ref.cid and set(other.objects) == set(ref.objects)
        assert all(bytes(other.objects[k]) == bytes(ref.objects[k]) for k in ref.objects)


def test_error_in_a_later_band_unwinds_the_pipeline(dc, monkeypatch):
    """A tile of the third band holds an infinity (to_fixed panics, fixed.rs:39-41): the call reports it while the upload and fetch
    threads of the earlier bands are still running, and the next call finds the rings and pools as it expects them."""
    from dcdf_amd import synth
    monkeypatch.setenv("K2R_SC_BAND_MB", "1")
    f = (synth.cells(0xDCDF0004, 0, 4, 0, 1024, 0, 512, np.int32) / 4.0).astype(np.float32)
    g = f.copy()
    g[2, 700, 100] = np.inf
    with pytest.raises(dc.DcdfError) as e:
        dc.Superchunk.build(g, [2, 8], fractional_bits=2)
    assert e.value.code == -2  # DCDF_ERR_NONFINITE
    check(dc, f, [2, 8], fractional_bits=2)
