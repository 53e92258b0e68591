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
    assert st["size_self"] == len(obj) - 2  # header + body minus the encoding byte Superchunk::size() leaves out


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


# ---- MMStruct3Build.size, derived term by term from the reference's own formulas (not from the oracle's byte strings) ----
def _bitmap_size(n):  # bitmap.rs:169-171
    return 8 + 4 * (n // 128) + 4 * ((n + 31) // 32)


def _dac_size(values):  # dac.rs:68-74 over the planes of dac.rs:96-132
    size, zz = 1, [((v << 1) ^ (v >> 63)) & (2 ** 64 - 1) for v in values]
    while zz:
        size += _bitmap_size(len(zz)) + len(zz)
        zz = [z >> 8 for z in zz if z >> 8]
    return size


def _superchunk_size(n_refs_elided, n_refs_external, maxs, mins):  # superchunk.rs:654-670, line by line
    return (7              # Resolver::HEADER_SIZE
            + 4 * 3        # shape
            + 4            # sidelen
            + 1            # levels
            + 4            # chunks_sidelen
            + 4            # subsidelen
            + 1            # fractional_bits              (no term for `encoding`, which save_to writes at superchunk.rs:692)
            + 4            # n_references
            + n_refs_elided * 1 + n_refs_external * 5     # Reference::size, superchunk.rs:868-876
            + 36           # external_cid.encoded_len(): CIDv1 + sha2-256
            + 4            # n_local
            + 0            # local chunks: Superchunk::build passes vec![]
            + _dac_size(maxs) + _dac_size(mins))


def test_build_size_external_subchunks_term_by_term():
    """superchunk.rs:259-262: size = data.size() + external.size() + sum of MMStruct3::size of the stored sub-chunks, for
    the [2, 2] fixture of superchunk.rs:1068-1094 (array16: 16 tiles of 4x4, 100 instants)."""
    import oracle_lib as O
    a = array_n(16)
    store = OS.Store()
    obj, st = OS.superchunk_build(a, [2, 2], 2, store)
    tiles = [a[:, r:r + 4, c:c + 4] for r in range(0, 16, 4) for c in range(0, 16, 4)]
    maxs = [int(t[i].max()) for i in range(100) for t in tiles]  # instant-major, superchunk.rs:190-198
    mins = [int(t[i].min()) for i in range(100) for t in tiles]
    self_size = _superchunk_size(0, 16, maxs, mins)
    links_size = 7 + 4 + 36 * 4                                                # links.rs:90-92, 4 distinct CIDs
    subs = sum(len(O.chunk_build(t)) + 1 for t in tiles)                      # Chunk::size() == bytes written (chunk.rs:572-574) + 1 (mmstruct.rs:187-197)
    assert st["size_self"] == self_size == len(obj) - 2
    assert st["size"] == self_size + links_size + subs
    # by hand: 37 fixed bytes + 16 * 5 (references) + 36 (CID) + 4 (n_local) = 157; each Dac holds 1600 one-byte values:
    # 1 + (8 + 4 * (1600 // 128) + 4 * 50) + 1600 = 1857; 157 + 2 * 1857 = 3871.  Links: 7 + 4 + 4 * 36 = 155.
    assert (self_size, links_size) == (3871, 155)


def test_build_size_nested_term_by_term():
    """[1, 2, 2] over array17 (superchunk.rs:1168-1188): the four children are Superchunk nodes whose MMStruct3::size is
    Superchunk::size() + 1 (mmstruct.rs:187-197) -- each again without the encoding byte."""
    a = array_n(17)
    store = OS.Store()
    obj, st = OS.superchunk_build(a, [1, 2, 2], 2, store)
    assert st["size_self"] == len(obj) - 2
    import math
    import oracle_lib as O
    child_sizes = []
    for r in (0, 16):
        for c in (0, 16):
            sub = a[:, r:r + 16, c:c + 16]
            if all(int(sub[i].min()) == int(sub[i].max()) for i in range(sub.shape[0])):
                continue  # uniform in every instant: elided, nothing stored (superchunk.rs:144-148) -- the 1x1 corner
            side = max(sub.shape[1:])
            needed = math.ceil(math.log(side) / math.log(2)) if side > 1 else 0
            if needed <= 2:  # superchunk.rs:150-165: fits the next level's tree -> a plain sub-chunk
                child_sizes.append(len(O.chunk_build(sub)) + 1)
            else:
                cobj, cst = OS.superchunk_build(sub, [2, 2], 2, OS.Store())
                assert cst["size_self"] == len(cobj) - 2
                child_sizes.append(cst["size_self"] + 1)
    n_links = len(st["links"])
    assert len(child_sizes) == sum(r is not None for r in st["references"])
    assert st["size"] == st["size_self"] + (7 + 4 + 36 * n_links) + sum(child_sizes)
