"""Host logic of the Span / Dataset facade (dcdf_amd/dataset.py) that needs no GPU: the Dac reader against the oracle's Dac
writer (dac.rs:37-132), Span nodes as span.rs:288-324 frames them, coordinates (py-dcdf __init__.py:150-243)."""
import numpy as np
import pytest

import oracle_lib as O
from dcdf_amd import dataset as D


def test_dac_reader_matches_the_oracle_writer():
    rng = np.random.default_rng(3)
    for vals in ([], [0], [0, 2, -3, -512, 131073, -1073741866], rng.integers(-2 ** 40, 2 ** 40, size=500).tolist(),
                 rng.integers(-100, 100, size=1000).tolist(), [2 ** 62, -(2 ** 62), 0, 1]):
        blob = b"\x07" + O.dac_serialize(vals) + b"\x09"
        got, pos = D._dac_values(blob, 1)
        assert got.tolist() == list(vals) and pos == len(blob) - 1


def test_span_node_roundtrip_and_framing():
    cids = [bytes([1, 0x12, 0x12, 0x20]) + bytes([i]) * 32 for i in range(3)]
    s = D._Span(D.MMEncoding.F32, [45, 16, 24], 20, cids)
    obj = s.serialize()
    assert obj[:8] == bytes([0xDC, 0xE0, 0, 0, 0, 1, 2, 3])           # object header, NODE_MMSTRUCT3, NODE_SPAN
    assert obj[8] == 32 and len(obj) == 8 + 1 + 5 * 4 + 3 * 36        # encoding, shape, stride, count, CIDs (span.rs:293-303)
    p = D._Span.parse(obj)
    assert (p.encoding, p.shape, p.stride, p.spans) == (32, [45, 16, 24], 20, cids)
    r = D.Resolver()
    cid = r.save(obj)
    assert isinstance(r.node(cid), D._Span) and r.ls(cid) == [("0", cids[0]), ("1", cids[1]), ("2", cids[2])]


def test_coordinates_and_private_constructors():
    t = D.Coordinate.time("t", np.datetime64("1979-01-01"), np.timedelta64(1, "D"))
    assert t[2] == np.datetime64("1979-01-03") and t.dtype == np.datetime64
    with pytest.raises(ValueError):
        len(t)
    y = D.Coordinate.range("y", -89.75, 0.5, 360, np.float32)
    assert len(y) == 360 and y[1] == np.float32(-89.25) and y.dtype == np.float32 and y[358:].tolist() == [89.25, 89.75]
    with pytest.raises(ValueError):
        y[1:5:2]
    for cls in (D.Dataset, D.Coordinate, D.Variable):
        with pytest.raises(RuntimeError):
            cls(None)
    ds = D.Dataset.new([t, y, D.Coordinate.range("x", 0, 1, 4, np.int32)], [360, 4], D.Resolver())
    ds2 = ds.add_variable("v", 10, 20, [4, 5], True, 3, np.float64)
    assert ds.variables == [] and ds2.v.round and ds2.v.fractional_bits == 3 and ds2.v.shape == (0, 360, 4) and ds2.v.dtype is np.float64
    with pytest.raises(AttributeError):
        ds2.w
