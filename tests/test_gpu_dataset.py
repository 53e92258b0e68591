"""The Span / Dataset facade (dcdf_amd/dataset.py) against the assertions of the reference's own Python test-suite
(py-dcdf/tests/test_dcdf.py), whose data and recipe are transcribed in tests/golden/pydcdf_fixture.json by
tests/golden/make_pydcdf_fixture.py (the reference's extension module cannot be built here).  Every append runs
dcdf_superchunk_build on the GPU; every read is a batched GPU query."""
import itertools
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "pydcdf_fixture.json")) as f:
    FX = json.load(f)
TEST_ARRAY = np.array(FX["test_array"])


@pytest.fixture(scope="module")
def dcdf():
    import dcdf_amd
    from dcdf_amd import _lib, dataset
    assert _lib.lib().dcdf_device_name(), "no GPU"
    return dataset


def make_data(instants):  # test_dcdf.py:46-49
    data = np.tile(TEST_ARRAY, [instants // 3 + 1, 2, 2])[:instants]
    assert data.shape == (instants, 16, 16)
    return data


def make_one(dcdf, resolver, dtype):  # test_dcdf.py:52-57
    t = dcdf.Coordinate.time("t", 0, np.timedelta64(100, "s"))
    y = dcdf.Coordinate.range("y", -160, 20, 16, dtype)
    x = dcdf.Coordinate.range("x", -200, 25, 16, dtype)
    return dcdf.Dataset.new([t, y, x], [16, 16], resolver)


@pytest.mark.parametrize("dtype", [np.int32, np.int64, np.float32, np.float64])
def test_new(dcdf, dtype):  # test_dcdf.py:65-103
    dataset = make_one(dcdf, dcdf.Resolver(), dtype)
    assert [c.name for c in dataset.coordinates] == ["t", "y", "x"]
    assert dataset.shape == (16, 16) and dataset.prev is None and dataset.cid is None and len(dataset.variables) == 0
    with pytest.raises(ValueError):
        assert len(dataset.t) == 0
    assert dataset.t[10] == np.datetime64("1970-01-01T00:16:40")
    assert dataset.t.dtype == np.datetime64
    expected = np.arange(np.datetime64("1970-01-01T00:33:20"), np.datetime64("1970-01-01T00:50:00"), np.timedelta64("100", "s"))
    assert np.array_equal(dataset.t[20:30], expected)
    assert len(dataset.get_coordinate("y")) == 16 and dataset.y[10] == 40 and dataset.y.dtype == dtype
    assert np.array_equal(dataset.y[10:], np.arange(40, 160, 20))
    assert len(dataset.get_coordinate("x")) == 16 and dataset.x[10] == 50 and dataset.x.dtype == dtype
    assert np.array_equal(dataset.x[:10], np.arange(-200, 50, 25))
    with pytest.raises(AttributeError):
        dataset.doesnotexist


@pytest.fixture(scope="module")
def populated(dcdf):  # test_dcdf.py:110-170
    resolver = dcdf.Resolver()
    dataset = make_one(dcdf, resolver, np.float64)
    assert dataset.cid is None
    test_data = {}
    cid = None
    for v in FX["variables"]:
        dtype = getattr(np, v["dtype"])
        data = make_data(v["instants"]).astype(dtype)
        kw = {"dtype": dtype}
        if v["round"]:
            dataset = dataset.add_variable(v["name"], v["span_size"], v["chunk_size"], v["k2_levels"], True, v["fractional_bits"], **kw)
        else:
            dataset = dataset.add_variable(v["name"], v["span_size"], v["chunk_size"], v["k2_levels"], **kw)
        cuts = [0] + v["splits"] + [v["instants"]]
        for a, b in zip(cuts[:-1], cuts[1:]):
            dataset = dataset.append(v["name"], data[a:b])
        if v["round"]:
            data = ((data * 4 + 0.001).round() / 4).astype(dtype)  # what two fractional bits keep (test_dcdf.py:149,158)
            assert dataset.cid is None and dataset.prev == cid
        test_data[v["name"]] = data
        if v["name"] == FX["commit_after"]:
            assert dataset.prev is None
            cid = dataset.commit()
            dataset = resolver.get_dataset(cid)
            assert dataset.cid == cid
    return dataset, test_data


VARIABLES = [v["name"] for v in FX["variables"]]


def test_populate(populated):  # test_dcdf.py:183-232
    dataset, _ = populated
    for v in FX["variables"]:
        var = dataset.get_variable(v["name"])
        assert (var.name, var.round, var.span_size, var.chunk_size, var.k2_levels) == (v["name"], v["round"], 10, 20, (2, 2))
        assert var.dtype is getattr(np, v["dtype"])
        if v["round"]:
            assert var.fractional_bits == 2
        assert var.shape == (v["instants"], 16, 16)


@pytest.mark.parametrize("var", VARIABLES)
def test_get(populated, var):  # test_dcdf.py:235-247
    dataset, test_data = populated
    data, variable = test_data[var], getattr(dataset, var)
    instants, rows, cols = variable.shape
    for instant in range(0, instants, 13):
        for row in range(0, rows, 4):
            for col in range(0, cols, 3):
                expected, got = data[instant, row, col], variable[instant, row, col].data
                if np.isnan(expected) and np.isnan(got):
                    continue
                assert got == expected


@pytest.mark.parametrize("var", VARIABLES)
def test_cell(populated, var):  # test_dcdf.py:250-261
    dataset, test_data = populated
    data, variable = test_data[var], dataset.get_variable(var)
    instants, rows, cols = variable.shape
    for row in range(0, rows, 4):
        for col in range(0, cols, 3):
            start = row + col
            end = instants - start
            assert np.array_equal(variable[start:end, row, col].data, data[start:end, row, col], equal_nan=True)


@pytest.mark.parametrize("var", VARIABLES)
def test_window(populated, var):  # test_dcdf.py:264-277
    dataset, test_data = populated
    data, variable = test_data[var], dataset.get_variable(var)
    instants, rows, cols = variable.shape
    for top in range(0, rows // 2, 4):
        bottom = top + rows // 2
        for left in range(0, cols // 2, 3):
            right = left + cols // 2
            start = top + bottom
            end = instants - start
            assert np.array_equal(variable[start:end, top:bottom, left:right].data, data[start:end, top:bottom, left:right], equal_nan=True)


def test_all_slice_permutations(populated):  # test_dcdf.py:280-299
    dataset, test_data = populated
    data, variable = test_data["apples"], dataset.apples
    slice_args = [42, slice(23, 80), slice(None, 20)]
    for t, y in itertools.product([42, slice(23, 80)], [9, slice(6, None)]):
        slice_args.append((t, y))
    for t, y, x in itertools.product([42, slice(23, 80)], [9, slice(6, 13)], [6, slice(3, 15)]):
        slice_args.append((t, y, x))
    for arg in slice_args:
        expected, got = data.__getitem__(arg), variable.__getitem__(arg).data
        if isinstance(expected, (int, float, np.number)):
            assert got == expected
        else:
            assert got.shape == expected.shape and np.array_equal(expected, got)


def test_errors(dcdf, populated):  # test_dcdf.py:302-336
    dataset, _ = populated
    with pytest.raises(ValueError):
        dataset.append("apples", np.array(range(10), dtype=np.byte))
    with pytest.raises(ValueError):
        dcdf.Coordinate.range("foo", 0, 1, 10, np.byte)
    for cls in (dcdf.Dataset, dcdf.Coordinate, dcdf.Variable):
        with pytest.raises(RuntimeError):
            cls(None)
    with pytest.raises(ValueError):
        dcdf.Coordinate.range("foo", 0, 1, 10)[1:2:3]
    with pytest.raises(IndexError):
        dataset.apples[1, 2, 3, 4]


def test_span_tree_and_stored_objects(dcdf, populated):
    """What the reference's Span layer guarantees (span.rs:50-109, dataset.rs:880-935): strides multiply by span_size per level,
    every full sub-span holds exactly `stride` instants, the objects in the store are framed as resolver.rs:126-138 frames them
    and named by their SHA-256 (testing.rs:172-183)."""
    import hashlib
    dataset, _ = populated
    res = dataset.bananas._resolver
    for cid, obj in res.objects.items():
        assert obj[:6] == bytes([0xDC, 0xE0, 0, 0, 0, 1]) and cid == bytes([1, 0x12, 0x12, 0x20]) + hashlib.sha256(obj).digest()
    root = res.node(dataset.bananas.cid)                     # 511 instants, chunk_size 20, span_size 10: 26 chunks -> 3 bottom spans
    assert root.stride == 200 and root.shape == [511, 16, 16] and len(root) == 3
    bottoms = [res.node(c) for c in root.spans]
    assert [b.stride for b in bottoms] == [20, 20, 20] and [len(b) for b in bottoms] == [10, 10, 6]
    assert [b.shape[0] for b in bottoms] == [200, 200, 111]
    last = res.node(bottoms[-1].spans[-1])                   # the incomplete tail chunk: 11 instants, re-encoded on every append
    assert last.shape == [11, 16, 16]
    hits = dataset.bananas.search(100, 140, 0, 16, 0, 16, 9, 9)
    want = np.argwhere(make_data(511).astype(np.int32)[100:140] == 9) + np.array([100, 0, 0])
    assert np.array_equal(hits, want)


def test_real_world_data(dcdf):  # test_dcdf.py:340-365: one day of CPC precipitation, 360 x 720 float32, k2_levels [4, 6]
    rw = FX["real_world"]
    testdata = np.load(os.path.join(HERE, "golden", rw["file"]))["precip"].reshape(rw["shape"]).astype(np.float32)
    t = dcdf.Coordinate.time("time", np.datetime64("1979-01-01"), np.timedelta64(1, "D"))
    lat = dcdf.Coordinate.range("latitude", -89.75, 0.5, 360, np.float32)
    lon = dcdf.Coordinate.range("longitude", -179.75, 0.5, 720, np.float32)
    dataset = dcdf.Dataset.new([t, lat, lon], (360, 720), dcdf.Resolver())
    dataset = dataset.add_variable("precip", rw["span_size"], rw["chunk_size"], rw["k2_levels"])
    dataset = dataset.append("precip", testdata)
    variable = dataset.precip
    instants, rows, cols = variable.shape
    assert (instants, rows, cols) == (1, 360, 720)
    for row in range(0, rows, 4):
        for col in range(0, cols, 3):
            expected, got = testdata[0, row, col], variable[0, row, col].data
            assert (np.isnan(expected) and np.isnan(got)) or got == expected
    assert np.array_equal(variable[0, 100:200, 300:500].data, testdata[0, 100:200, 300:500], equal_nan=True)
