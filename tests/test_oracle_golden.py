"""Pins the CPU oracle against the reference's own known-answer tests.

Every expected value comes from tests/golden/reference_vectors.json, which transcribes the literals
asserted by the cited reference tests.
"""
import json
import math
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_vectors.json")) as f:
    G = json.load(f)


def _arr(name, dtype=np.int64):
    def conv(x):
        return float("nan") if x == "nan" else x
    return np.array([[[conv(v) for v in row] for row in inst] for inst in G[name]], dtype=dtype)


def _f(x):
    return {"nan": float("nan"), "inf": float("inf"), "-inf": float("-inf")}.get(x, x) if isinstance(x, str) else x


def test_snapshot_build_known_answer():  # snapshot.rs:538-558
    g = G["snapshot_build"]
    d = O.snapshot_dump(_arr("array8")[0], k=2)
    assert d["nodemap"]["length"] == g["nodemap_length"]
    assert d["nodemap"]["words"] == g["nodemap_words"]
    assert d["max"] == g["max"]
    assert d["min"] == g["min"]
    assert d["size"] == d["serialized_len"]  # snapshot.rs:877-879
    assert d["size"] == G["derived_sizes"]["snapshot_array8_bytes"]


def test_snapshot_padding_known_answer():  # snapshot.rs:560-572
    data = np.zeros((9, 9), dtype=np.int64) + 5
    data[:8, :8] = _arr("array8")[0]
    d = O.snapshot_dump(data, k=2)
    assert d["nodemap"]["length"] == G["snapshot_padding"]["nodemap_length"]
    assert d["sidelen"] == 16
    assert O.sl_get(data, data, 0, 8, 8) == G["snapshot_padding"]["get_8_8"]


def test_snapshot_single_node_tree():  # snapshot.rs:586-599
    g = G["snapshot_single_node"]
    data = np.zeros((g["side"], g["side"]), dtype=np.int64) + g["value"]
    d = O.snapshot_dump(data)
    assert len(d["nodemap"]["words"]) == g["nodemap_words_len"]
    assert len(d["max"]) == g["max_level0_len"]
    assert len(d["min"]) == 0
    for r in range(0, 16, 5):
        for c in range(0, 16, 3):
            assert O.sl_get(data, data, 0, r, c) == g["value"]


@pytest.mark.parametrize("key,t_index", [("log_build_0_1", 1), ("log_build_0_2", 2)])
def test_log_build_known_answer(key, t_index):  # log.rs:901-937
    g = G[key]
    a = _arr("array8")
    d = O.log_dump(a[0], a[t_index], k=2)
    assert d["nodemap"]["length"] == g["nodemap_length"]
    assert d["nodemap"]["words"] == g["nodemap_words"]
    assert d["equal"]["length"] == g["equal_length"]
    assert d["equal"]["words"] == g["equal_words"]
    assert d["max"] == g["max"]
    assert d["min"] == g["min"]
    assert d["size"] == d["serialized_len"]  # log.rs:1624-1626
    if key == "log_build_0_1":
        assert d["size"] == G["derived_sizes"]["log_0_1_bytes"]


def test_log_padding_known_answer():  # log.rs:939-955
    data = np.zeros((3, 9, 9), dtype=np.int64) + 5
    data[:, :8, :8] = _arr("array8")
    data[0] = _arr("array9")[0]
    d = O.log_dump(data[0], data[1])
    assert d["nodemap"]["length"] == G["log_padding"]["nodemap_length"]
    assert O.sl_get(data[0], data[1], 1, 8, 8) == G["log_padding"]["get_8_8"]


def test_bitmap_from_bitmap():  # bitmap.rs:261-284
    for g in G["bitmap_from_bitmap"]:
        d = O.bitmap_dump(g["length"], g["bytes"])
        assert d["words"] == g["words"]
        assert d["index"] == g["index"]
        assert d["size"] == 8 + 4 * len(g["index"]) + 4 * len(g["words"])  # bitmap.rs:169-171


def test_bitmap_get_and_push():  # bitmap.rs:318-334
    bits = G["bitmap_get"]["bits"]
    d = O.bitmap_push_dump(bits)
    assert d["length"] == len(bits)
    word = d["words"][0]
    for i, b in enumerate(bits):
        assert ((word >> (31 - i)) & 1) == b


def test_bitmap_rank():  # bitmap.rs:336-359
    g = G["bitmap_rank"]
    bits = []
    for byte in g["bytes"]:
        bits += [(byte >> (7 - j)) & 1 for j in range(8)]
    for i in range(g["length"] + 1):
        r, naive, _ = O.bitmap_rank(g["length"], g["bytes"], i)
        assert r == naive == sum(bits[:i])
    with pytest.raises(O.OracleError):
        O.bitmap_rank(g["length"], g["bytes"], g["panic_at"])


def test_bitmap_rank_megabit():  # bitmap.rs:386-392 (random bits; indexed rank == naive rank)
    rng = np.random.default_rng(1)
    n = 1 << 20
    data = rng.integers(0, 256, n >> 3, dtype=np.uint8).tobytes()
    for i in rng.integers(0, n, 100):
        r, naive, _ = O.bitmap_rank(n, data, int(i))
        assert r == naive


def test_dac_known_answer():  # dac.rs:163-199
    g = G["dac_get"]
    d = O.dac_dump(g["values"])
    assert d["collect"] == g["values"]
    assert d["collect_reread"] == g["values"]
    assert d["size"] == d["serialized_len"]
    cont0 = d["levels"][0]["bitmap"]["words"][0]
    assert ((cont0 >> (31 - 2)) & 1) == g["level0_cont_bit2"]
    d = O.dac_dump(G["dac_this_one"]["values"])
    assert d["collect"] == G["dac_this_one"]["values"]


def test_dac_empty():
    d = O.dac_dump([])
    assert d["levels"] == [] and d["size"] == 1


def test_to_fixed_known_answers():  # fixed.rs:208-243
    for v, bits, rnd, want in G["to_fixed"]:
        assert O.to_fixed(_f(v), bits, rnd, "f64") == want
    v, bits, rnd, want = G["to_fixed_nan"]
    assert O.to_fixed(_f(v), bits, rnd) == want
    assert O.to_fixed(float("nan"), 12, 0, "f32") == 0


def test_to_fixed_panics():  # fixed.rs:258-299
    kinds = {"precision": -3, "nonfinite": -2, "overflow": -4}
    for v, bits, rnd, kind in G["to_fixed_panics"]:
        with pytest.raises(O.OracleError) as e:
            O.to_fixed(_f(v), bits, rnd)
        assert e.value.code == kinds[kind]


def test_from_fixed_known_answers():  # fixed.rs:245-256
    for v, bits, ft, want in G["from_fixed"]:
        assert O.from_fixed(v, bits, ft) == want
    assert math.isnan(O.from_fixed(0, 3, "f32")) and math.isnan(O.from_fixed(0, 3, "f64"))
    assert abs(O.from_fixed(6554 * 2 + 1, 16, "f64") - 0.1) < 1e-5


def test_round_trip_issue5():  # fixed.rs:305-309
    g = G["round_trip_issue5"]
    n = float(np.float32(g["value"]))
    assert O.from_fixed(O.to_fixed(n, g["bits"], 0, "f32"), g["bits"], "f32") == n


def test_to_fixed_negative_quirk():
    # fract() > 0 is false for negatives -> never rounded, never rejected.
    assert O.to_fixed(-0.0625, 3, 0) == int(-0.0625 * 8 * 2) + 1  # trunc(-1.0)+1 == 0
    assert O.to_fixed(-1.0625, 3, 1) == int(-1.0625 * 8 * 2) + 1


def test_suggest_fraction():  # fixed.rs:311-401
    for data, ft, rnd, bits in G["suggest_fraction"]:
        if data == "fixed_array":
            full = np.zeros((100, 8, 8), dtype=np.float32)
            src = _arr("fixed_array", np.float32)
            for i in range(100):
                full[i] = src[i % 3]
            arr = full
        else:
            arr = np.array([_f(x) for x in data])
        assert O.suggest_fraction(arr, ft) == (bool(rnd), bits)


def test_sidelen():  # snapshot.rs:118-119
    assert O.sidelen(8, 8) == 8 and O.sidelen(9, 9) == 16 and O.sidelen(1, 1) == 1 and O.sidelen(256, 200) == 256
    assert O.sidelen(9, 9, 3) == 9 and O.sidelen(10, 3, 3) == 27
    for e in range(0, 20):
        assert O.sidelen(1 << e, 1, 2) == 1 << e
        if e:
            assert O.sidelen((1 << e) + 1, 1, 2) == 1 << (e + 1)
