"""The superchunk oracle against what the reference's own fixtures assert (superchunk.rs:1006-1179, mmstruct.rs:463-479)."""
import json
import os

import numpy as np

import oracle_superchunk as OS

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_vectors.json")) as f:
    G = json.load(f)


def array_n(n, T=100):
    a8 = np.array(G["array8"], dtype=np.int64)
    a = np.stack([a8[i % 3] for i in range(T)])
    idx = np.arange(n) % 8
    return np.ascontiguousarray(a[:, idx][:, :, idx])


def counts(refs):
    return sum(r is not None for r in refs), sum(r is None for r in refs)


def test_no_subchunks():  # superchunk.rs:1006-1020: [3, 0] -> 64 references
    st = OS.superchunk_build(array_n(8), [3, 0], 2, OS.Store())[1]
    assert len(st["references"]) == 64


def test_no_subchunks_coarse():  # superchunk.rs:1040-1064: every 2x2 block uniform -> all elided
    a8 = array_n(8)
    a = np.repeat(np.repeat(a8, 2, axis=1), 2, axis=2)
    st = OS.superchunk_build(a, [3, 1], 2, OS.Store())[1]
    assert len(st["references"]) == 64 and counts(st["references"]) == (0, 64)


def test_external_subchunks_deduplicated():  # superchunk.rs:1068-1094: 16 external references, 4 distinct links
    store = OS.Store()
    obj, st = OS.superchunk_build(array_n(16), [2, 2], 2, store)
    assert len(st["references"]) == 16 and counts(st["references"]) == (16, 0) and len(st["links"]) == 4
    assert st["size_self"] == len(obj) - 1


def test_mixed_subchunks():  # superchunk.rs:1098-1131: 8 external, 8 elided
    st = OS.superchunk_build(array_n(17), [2, 3], 2, OS.Store())[1]
    assert len(st["references"]) == 16 and counts(st["references"]) == (8, 8)


def test_elide_everything():  # superchunk.rs:1135-1164
    a = np.zeros((100, 16, 16), dtype=np.int64) + np.arange(100)[:, None, None]
    st = OS.superchunk_build(a, [2, 2], 2, OS.Store())[1]
    assert counts(st["references"]) == (0, 16) and st["elided"] == 16


def test_nested_superchunks():  # superchunk.rs:1168-1188, mmstruct.rs:463-479
    store = OS.Store()
    obj, st = OS.superchunk_build(array_n(17), [1, 2, 2], 2, store)
    assert len(st["references"]) == 4
    kinds = sorted(o[6] for o in store.values())
    assert 1 in kinds and 2 in kinds  # Links and MMStruct3 objects were stored
