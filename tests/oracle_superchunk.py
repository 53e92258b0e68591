"""CPU restatement of `Superchunk::build` and the stored-object framing around it -- TEST INFRASTRUCTURE ONLY (the checker of
dcdf_amd/superchunk.py; never imported by the product).  Follows, line by line:

  superchunk.rs:88-270   Superchunk::build (tiling, elision, nested superchunks, references, de-dup through external_references)
  superchunk.rs:652-720  Superchunk::size / save_to            superchunk.rs:826-880  Reference::write_to
  mmbuffer.rs:366-499    MMBuffer3::min_max, min_max_float (its NaN behaviour included), compute_fractional_bits :596-613
  resolver.rs:17-18,126-138  object header (0xDCE0, version 1, node type)   mmstruct.rs:205-226  node tags
  links.rs:65-76         Links::save_to                            testing.rs:172-183   CIDv1(0x12, sha2-256)

Chunk::build bytes and Dac::from bytes come from the C++ oracle (oracle_lib).  Pinned by the structural assertions the
reference's own fixtures make (superchunk.rs:1006-1179): reference counts, elision counts, the four de-duplicated links."""
import hashlib
import math
import struct

import numpy as np

import oracle_lib as O

NODE_LINKS, NODE_MMSTRUCT3, NODE_SUBCHUNK, NODE_SUPERCHUNK = 1, 2, 4, 5
HEADER_SIZE = 7
ENC = {np.dtype("int32"): 4, np.dtype("int64"): 8, np.dtype("float32"): 32, np.dtype("float64"): 64}


def header(node_type):  # resolver.rs:130-133
    return struct.pack(">HIB", 0xDCDF + 1, 1, node_type)


def cid_of(obj):  # testing.rs:172-183: Cid::new_v1(SHA2_256 as codec, sha2-256 multihash)
    return bytes([0x01, 0x12, 0x12, 0x20]) + hashlib.sha256(obj).digest()


class Store(dict):
    """MemoryMapper (testing.rs:91-198): cid -> object bytes."""

    def save(self, obj):
        cid = cid_of(obj)
        self[cid] = obj
        return cid


def _ftype(a):
    return "f32" if a.dtype == np.float32 else "f64"


def min_max(a, fractional_bits, round_):
    """MMBuffer3::min_max (mmbuffer.rs:366-395): per instant (min, max) as stored i64."""
    out = []
    for sub in a:
        if a.dtype.kind == "i":
            out.append((int(sub.min()), int(sub.max())))
            continue
        it = iter(sub.ravel().tolist())  # mmbuffer.rs:466-499 (min_max_float), NaN quirks and all
        first = next(it)
        mn = mx = first
        while mn != mn:
            try:
                v = next(it)
            except StopIteration:
                break
            mn = mx = v
        for n in it:
            if n != n:
                mn = n
            elif n < mn:
                mn = n
            elif n > mx:
                mx = n
        tf = lambda v: O.to_fixed(v, fractional_bits, round_, _ftype(a))
        out.append((tf(mn), tf(mx)))
    return out


def compute_fractional_bits(a, fractional_bits, round_):  # mmbuffer.rs:596-613
    if a.dtype.kind == "i":
        return 0
    rnd, bits = O.suggest_fraction(np.ascontiguousarray(a), _ftype(a))
    if round_:
        return min(bits, fractional_bits)
    if rnd:
        raise O.OracleError(-3)  # panic!("loss of precision")
    return bits


def superchunk_build(a, levels, k, store, fractional_bits=0, round_=False):
    """Returns (stored object bytes of the MMStruct3::Superchunk node, stats dict).  Sub-objects are saved into `store`."""
    instants, rows, cols = a.shape
    sidelen = float(max(rows, cols))
    total_levels = int(math.ceil(math.log(sidelen) / math.log(k))) if sidelen > 1 else 0  # superchunk.rs:98-101
    if sum(levels) != total_levels:
        raise ValueError("Need %d tree levels to encode array, but %d levels passed in." % (total_levels, sum(levels)))
    sidelen = k ** total_levels
    sublevels = levels[1:]
    at_bottom = len(sublevels) == 1
    lv = levels[0]
    subsidelen = k ** lv
    chunks_sidelen = sidelen // subsidelen
    elided, mm, builds = [], [], []
    for row in range(subsidelen):
        top = row * chunks_sidelen
        bottom = min(top + chunks_sidelen, rows)
        for col in range(subsidelen):
            left = col * chunks_sidelen
            right = min(left + chunks_sidelen, cols)
            if top >= rows or left >= cols:
                elided.append(True)
                mm.append([(0, 0)] * instants)
                continue
            sub = a[:, top:bottom, left:right]
            smm = min_max(sub, fractional_bits, round_)
            mm.append(smm)
            if all(x == y for x, y in smm):
                elided.append(True)
                continue
            shape = sub.shape
            build_subchunk = at_bottom
            if not at_bottom:
                sl = float(max(shape[1:]))
                needed = int(math.ceil(math.log(sl) / math.log(k))) if sl > 1 else 0
                build_subchunk = needed <= sublevels[0]
            fb = compute_fractional_bits(sub, fractional_bits, round_)  # superchunk.rs:167
            if build_subchunk:
                data, ns, nl, _ = O.chunk_build(sub, k=k, fractional_bits=fb, round_=round_, want_snapshots=True)
                obj = header(NODE_MMSTRUCT3) + bytes([NODE_SUBCHUNK]) + data  # resolver.rs:126-138, mmstruct.rs:215-218
                builds.append({"obj": obj, "size": len(data) + 1, "snapshots": ns, "logs": nl})
            else:
                obj, st = superchunk_build(sub, sublevels, k, store, fb, round_)
                builds.append({"obj": obj, "size": st["size_self"] + 1, "snapshots": st["snapshots"], "logs": st["logs"]})
            elided.append(False)
    n_sub = subsidelen * subsidelen
    mins, maxs = [], []
    for i in range(instants):  # instant-major (superchunk.rs:190-198)
        for s in range(n_sub):
            mins.append(mm[s][i][0])
            maxs.append(mm[s][i][1])
    external, ext_index, references, sizes = [], {}, [], []
    n_elided = n_snap = n_logs = 0
    bi = iter(builds)
    for i in range(n_sub):
        if elided[i]:
            n_elided += 1
            references.append(None)
            continue
        b = next(bi)
        if all(maxs[n] == mins[n] for n in range(i, n_sub * instants, n_sub)):
            n_elided += 1
            references.append(None)
            continue
        sizes.append(b["size"])
        cid = store.save(b["obj"])
        if cid not in ext_index:  # superchunk.rs:222-232
            ext_index[cid] = len(external)
            external.append(cid)
        references.append(ext_index[cid])
        n_snap += b["snapshots"]
        n_logs += b["logs"]
    links_obj = header(NODE_LINKS) + struct.pack(">I", len(external)) + b"".join(external)  # links.rs:65-76
    size_external = HEADER_SIZE + 4 + sum(len(c) for c in external)
    external_cid = store.save(links_obj)
    body = struct.pack(">IIIIBIIBB", instants, rows, cols, sidelen, lv, chunks_sidelen, subsidelen, fractional_bits, ENC[a.dtype])
    body += struct.pack(">I", len(references))
    for r in references:  # superchunk.rs:843-861
        body += b"\x00" if r is None else b"\x02" + struct.pack(">I", r)
    body += external_cid + struct.pack(">I", 0)
    body += O.dac_serialize(maxs) + O.dac_serialize(mins)
    obj = header(NODE_MMSTRUCT3) + bytes([NODE_SUPERCHUNK]) + body
    # Superchunk::size(), superchunk.rs:652-670: the formula lists every field save_to writes except the one `encoding` byte
    # of superchunk.rs:692, so it is one less than header + body (tests/test_oracle_superchunk.py derives it term by term)
    size_self = HEADER_SIZE + len(body) - 1
    stats = {"size_self": size_self, "size": size_self + size_external + sum(sizes), "elided": n_elided, "local": 0,
             "external": len(external), "snapshots": n_snap, "logs": n_logs, "references": references, "links": external}
    return obj, stats
