"""Span / Dataset facade over the MI355X chunk engine, shaped like py-dcdf (py-dcdf/dcdf/__init__.py:37-350): the callers
either side of the hot path, with an in-memory store instead of IPFS.

What is mirrored, and from where:
  Dataset.new / add_variable / append / commit / get_variable / attribute access    py-dcdf __init__.py:52-148, dataset.rs:116-384
  Variable.append: chunk_size slices -> Superchunk::build, span tree growth           dataset.rs:834-935 (tail re-encode :162-213, :937-957)
  Span: append / update / find_span, queries routed over the time axis               span.rs:50-275
  Superchunk node: tile routing, elided tiles answered from the max Dac               superchunk.rs:313-633, :672-768
  Variable.get / cell / window / __getitem__ with py-dcdf's slicing rules            py-dcdf __init__.py:272-350
  Coordinate.time / range                                                            py-dcdf __init__.py:150-243

What runs where: the encode of every appended slice is dcdf_superchunk_build (tile min/max, fractional bits, every
Chunk::build, Dacs, CIDs on the GPU); every query ends in ONE batched GPU call over the sub-chunks it touches
(dcdf_query_get_batch / fill_cell_batch / fill_window_batch_typed).  The tree walking in between is host control flow, as in
the reference.  Span, Superchunk, Links and sub-chunk objects are stored byte-for-byte as the reference's Resolver would store
them (same framing, same CIDs); the Dataset / Variable / Coordinate records are kept as plain Python state -- their on-disk
node encodings (dataset.rs:386-640) belong to the storage layer that is out of scope here, so `commit()` names a snapshot of
that state by the SHA-256 of its canonical description, not by the reference's Dataset-node CID.

No IPFS, no LRU cache: `Resolver` owns a dict."""
import hashlib
import struct

import numpy as np

from . import chunk as _chunk
from .chunk import Chunk, Cube
from .superchunk import Superchunk

NODE_LINKS, NODE_MMSTRUCT3, NODE_SPAN, NODE_SUBCHUNK, NODE_SUPERCHUNK = 1, 2, 3, 4, 5
_FACTORY_KEY = object()


class MMEncoding:  # py-dcdf __init__.py:8-29
    Time, I32, I64, F32, F64 = 0, 4, 8, 32, 64
    from_dtype = {np.datetime64: Time, np.int32: I32, np.int64: I64, np.float32: F32, np.float64: F64}
    to_dtype = {Time: np.datetime64, I32: np.int32, I64: np.int64, F32: np.float32, F64: np.float64}


def _header(node_type):  # resolver.rs:130-133
    return struct.pack(">HIB", 0xDCE0, 1, node_type)


def _cid(obj):  # testing.rs:172-183 (MemoryMapper): CIDv1, codec 0x12, sha2-256
    return bytes([0x01, 0x12, 0x12, 0x20]) + hashlib.sha256(obj).digest()


def _from_fixed(n, bits, dtype):  # fixed.rs:81-86: 0 is NaN, else (n - 1) / 2^(bits + 1)
    if dtype in (np.float32, np.float64):
        return dtype(np.nan) if n == 0 else dtype((n - 1) / float(1 << (bits + 1)))
    return dtype(n)


def _dac_values(buf, pos):
    """Dac::read_from + get of every index (dac.rs:48-93): returns (int64 values, position after the Dac)."""
    nlev = buf[pos]
    pos += 1
    planes = []
    for _ in range(nlev):
        (nbits, k) = struct.unpack_from(">II", buf, pos)
        pos += 8 + 4 * (nbits // 32 // k)
        nw = (nbits + 31) // 32
        bits = np.unpackbits(np.frombuffer(buf, dtype=np.uint8, count=4 * nw, offset=pos))[:nbits].astype(bool)  # MSB first (bitmap.rs:176-183)
        pos += 4 * nw
        byt = np.frombuffer(buf, dtype=np.uint8, count=nbits, offset=pos).astype(np.uint64)
        pos += nbits
        planes.append((bits, byt))
    if not planes:
        return np.zeros(0, dtype=np.int64), pos
    n = len(planes[0][1])
    zz = np.zeros(n, dtype=np.uint64)
    idx = np.arange(n)            # which value each entry of the current plane belongs to
    for j, (bits, byt) in enumerate(planes):
        zz[idx] |= byt << np.uint64(8 * j)
        idx = idx[bits]           # values that continue: plane j + 1 holds them in order (rank = position among the set bits)
    val = (zz >> np.uint64(1)).astype(np.int64) ^ -(zz & np.uint64(1)).astype(np.int64)  # zig-zag decode, dac.rs:139-142
    return val, pos


class _Span:  # span.rs:22-86 (an immutable value: append / update return new spans)
    def __init__(self, encoding, shape, stride, spans):
        self.encoding, self.shape, self.stride, self.spans = encoding, list(shape), stride, list(spans)

    def serialize(self):  # mmstruct.rs:215-218 + span.rs:288-303
        body = struct.pack(">BIIIII", self.encoding, self.shape[0], self.shape[1], self.shape[2], self.stride, len(self.spans))
        return _header(NODE_MMSTRUCT3) + bytes([NODE_SPAN]) + body + b"".join(self.spans)

    @staticmethod
    def parse(obj):
        enc, t, r, c, stride, n = struct.unpack_from(">BIIIII", obj, 8)
        return _Span(enc, [t, r, c], stride, [obj[29 + 36 * i:29 + 36 * i + 36] for i in range(n)])

    def __len__(self):
        return len(self.spans)


class _SuperNode:
    """A stored Superchunk node (superchunk.rs:672-768), parsed for routing."""

    def __init__(self, obj):
        (t, r, c, self.sidelen, self.levels, self.chunks_sidelen, self.subsidelen, self.fractional_bits, self.encoding,
         nref) = struct.unpack_from(">IIIIBIIBBI", obj, 8)
        self.shape = [t, r, c]
        pos = 8 + struct.calcsize(">IIIIBIIBBI")
        self.references = []
        for _ in range(nref):  # superchunk.rs:843-861
            tag = obj[pos]
            pos += 1
            if tag == 0:
                self.references.append(None)
            else:
                self.references.append((tag, struct.unpack_from(">I", obj, pos)[0]))
                pos += 4
        self.external_cid = obj[pos:pos + 36]
        pos += 36
        (n_local,) = struct.unpack_from(">I", obj, pos)
        pos += 4
        if n_local:
            raise ValueError("local sub-chunks are never written by Superchunk::build (superchunk.rs:245)")
        self.max, pos = _dac_values(obj, pos)
        self.min, pos = _dac_values(obj, pos)


class Resolver:
    """py-dcdf's Resolver (__init__.py:37-50) over an in-memory store: cid -> stored object bytes."""

    def __init__(self, cache_bytes=None):
        self.objects = {}
        self._datasets = {}
        self._open = {}    # cid -> opened dcdf_amd.Chunk (device-resident)
        self._nodes = {}   # cid -> parsed _Span / _SuperNode / [link cids]

    def save(self, obj):
        cid = _cid(obj)
        self.objects[cid] = obj
        return cid

    def get_dataset(self, cid):
        return self._datasets[cid]

    def ls(self, cid):
        node = self.node(cid)
        if isinstance(node, _Span):
            return [(str(i), c) for i, c in enumerate(node.spans)]
        if isinstance(node, _SuperNode):
            return [("subchunks", node.external_cid)]
        return []

    def node(self, cid):
        """The parsed object behind a CID: _Span, _SuperNode, list of link CIDs, or an opened Chunk."""
        n = self._nodes.get(cid)
        if n is None:
            obj = self.objects[cid]
            kind = obj[6]
            if kind == NODE_LINKS:  # links.rs:65-76
                (cnt,) = struct.unpack_from(">I", obj, 7)
                n = [obj[11 + 36 * i:11 + 36 * i + 36] for i in range(cnt)]
            elif kind == NODE_MMSTRUCT3 and obj[7] == NODE_SPAN:
                n = _Span.parse(obj)
            elif kind == NODE_MMSTRUCT3 and obj[7] == NODE_SUPERCHUNK:
                n = _SuperNode(obj)
            elif kind == NODE_MMSTRUCT3 and obj[7] == NODE_SUBCHUNK:
                n = Chunk(obj[8:])  # opened on the GPU (dcdf_chunk_open) and kept
            else:
                raise ValueError("unexpected node type %d" % kind)
            self._nodes[cid] = n
        return n


class Coordinate:  # py-dcdf __init__.py:150-243
    def __init__(self, inner=None, private=None):
        if private is not _FACTORY_KEY:
            raise RuntimeError("Coordinate objects come from Coordinate.time(...) or Coordinate.range(...)")
        self.name, self.kind, self.start, self.step, self.steps, self._dtype = inner

    @classmethod
    def time(cls, name, start, step):
        if isinstance(start, np.datetime64):
            start = int(start.astype("datetime64[s]").astype(np.int64))
        if isinstance(step, np.timedelta64):
            step = int(step / np.timedelta64(1, "s"))
        return cls((name, "time", int(start), int(step), None, np.datetime64), _FACTORY_KEY)

    @classmethod
    def range(cls, name, start, step, steps, dtype=np.float64):
        if dtype not in (np.int32, np.int64, np.float32, np.float64):
            raise ValueError("a range Coordinate holds int32, int64, float32 or float64 values, not %r" % (dtype,))
        return cls((name, "range", start, step, int(steps), dtype), _FACTORY_KEY)

    @property
    def dtype(self):
        return self._dtype

    def get(self, index):
        if self.kind == "time":
            return np.datetime64(self.start + self.step * int(index), "s")
        return self._dtype(self.start + self.step * index)

    def slice(self, start, end):
        if self.kind == "time":
            return np.array([self.get(i) for i in range(start, end)], dtype="datetime64[s]")
        return (self.start + self.step * np.arange(start, end)).astype(self._dtype)

    def __getitem__(self, key):
        """One coordinate value for an integer, an array of them for `a:b` (open ends allowed; strides are not)."""
        if not isinstance(key, slice):
            return self.get(key)
        if key.step not in (None, 1):
            raise ValueError("a Coordinate can only be sliced with stride 1, got %r" % (key.step,))
        first = key.start if key.start is not None else 0
        last = key.stop if key.stop is not None else len(self)  # (an unbounded time axis needs an explicit stop)
        return self.slice(first, last)

    def __len__(self):
        if self.kind == "time":
            raise ValueError("a time coordinate is unbounded")  # (py-dcdf: len() of a time axis raises ValueError, test_dcdf.py:81-82)
        return self.steps


class Variable:  # py-dcdf __init__.py:246-336 over dataset.rs:642-986
    def __init__(self, inner=None, private=None):
        if private is not _FACTORY_KEY:
            raise RuntimeError("Variable objects come from Dataset.add_variable(...)")
        (self.name, self._round, self.span_size, self.chunk_size, self._k2_levels, self.encoding, self.cid, self._resolver) = inner

    # ---- what py-dcdf exposes -------------------------------------------------------------------------------------------
    @property
    def k2_levels(self):
        return tuple(self._k2_levels)

    @property
    def round(self):
        return self._round is not None

    @property
    def fractional_bits(self):
        return 0 if self._round is None else self._round

    @property
    def dtype(self):
        return MMEncoding.to_dtype[self.encoding]

    @property
    def shape(self):
        return tuple(self._root().shape)

    def get(self, instant, row, col):
        self._check(instant, instant + 1, row, row + 1, col, col + 1)
        return self.cell(instant, instant + 1, row, col)[0]

    def cell(self, start, stop, row, col):
        self._check(start, stop, row, row + 1, col, col + 1)
        out = np.zeros(stop - start, dtype=self.dtype)
        jobs = []
        for cid, a0, a1, o0 in self._time_pieces(start, stop):
            self._route_cell(cid, a0, a1, row, col, out, o0, jobs)
        if jobs:  # ONE launch for every sub-chunk series (dcdf_query_fill_cell_batch)
            series = _chunk.fill_cell_batch([j[0] for j in jobs], [j[1] for j in jobs])
            for (ch, _, o0, fb), s in zip(jobs, series):
                out[o0:o0 + len(s)] = self._typed(s, fb)
        return out

    def window(self, start, stop, top, bottom, left, right):
        self._check(start, stop, top, bottom, left, right)
        out = np.zeros((stop - start, bottom - top, right - left), dtype=self.dtype)
        jobs = []
        for cid, a0, a1, o0 in self._time_pieces(start, stop):
            self._route_window(cid, a0, a1, top, bottom, left, right, out[o0:o0 + (a1 - a0)], jobs)
        if jobs:  # ONE launch, typed result (dcdf_query_fill_window_batch_typed: from_fixed on the GPU with each chunk's own bits)
            flat, off = _chunk.fill_window_batch([j[0] for j in jobs], [j[1] for j in jobs], dtype=self.dtype)
            for (ch, cu, dst), o in zip(jobs, off):
                dst[...] = flat[int(o):int(o) + dst.size].reshape(dst.shape)
        return out

    def search(self, start, stop, top, bottom, left, right, lower, upper):
        """(instant, row, col) of the cells with lower <= stored value <= upper (mmarray.rs:206; span.rs:231-270): not in
        py-dcdf, which never exposed search.  Integers only (stored values)."""
        self._check(start, stop, top, bottom, left, right)
        hits = []
        for cid, a0, a1, o0 in self._time_pieces(start, stop):
            self._route_search(cid, a0, a1, top, bottom, left, right, lower, upper, start + o0 - a0, 0, 0, hits)
        return np.array(sorted(hits), dtype=np.int64).reshape(-1, 3)

    def __getitem__(self, key):
        """NumPy-style basic indexing with up to three integers / unit-stride slices; the answer is lazy (`.data` decodes).
        Every axis is first reduced to a half-open range plus a "drop this axis" mark; the pattern of marks then picks the
        cheapest query: a point, a cell's time series, or a window whose marked axes are squeezed out."""
        axes = key if isinstance(key, tuple) else (key,)
        if len(axes) > 3:
            raise IndexError("a Variable has 3 axes (instant, row, col); %d indices were given" % len(axes))
        lo, hi, drop = [], [], []
        for axis, extent in enumerate(self.shape):
            ix = axes[axis] if axis < len(axes) else slice(None)
            if isinstance(ix, (int, np.integer)):
                at = int(ix) + (extent if ix < 0 else 0)
                lo.append(at)
                hi.append(at + 1)
                drop.append(True)
            elif isinstance(ix, slice):
                a, b, step = ix.indices(extent)
                if step != 1:
                    raise IndexError("axis %d: only unit-stride slices are supported" % axis)
                lo.append(a)
                hi.append(max(a, b))
                drop.append(False)
            else:
                raise IndexError("axis %d: cannot index with %r" % (axis, type(ix).__name__))

        def decode():
            if drop == [True, True, True]:
                return self.get(lo[0], lo[1], lo[2])
            if drop == [False, True, True]:
                return self.cell(lo[0], hi[0], lo[1], lo[2])
            block = self.window(lo[0], hi[0], lo[1], hi[1], lo[2], hi[2])
            return block.reshape([n for n, d in zip(block.shape, drop) if not d])

        return _Lazy(decode)

    # ---- routing ----------------------------------------------------------------------------------------------------------
    def _root(self):
        return self._resolver.node(self.cid)

    def _check(self, start, stop, top, bottom, left, right):  # mmarray.rs:218-229 panics; here an IndexError
        t, r, c = self.shape
        if not (0 <= start <= stop <= t and 0 <= top <= bottom <= r and 0 <= left <= right <= c):
            raise IndexError("window (%d:%d, %d:%d, %d:%d) outside the variable's shape %s" % (start, stop, top, bottom, left, right, (t, r, c)))

    def _time_pieces(self, start, stop):
        """The bottom-level nodes (superchunks of chunk_size instants) a time range meets: (cid, local start, local end, offset
        into the result) -- Span::fill_window's loop (span.rs:190-216) applied level by level."""
        out = []

        def walk(cid, a0, a1, o0):
            node = self._resolver.node(cid)
            if not isinstance(node, _Span):
                out.append((cid, a0, a1, o0))
                return
            span, inst = a0 // node.stride, a0 % node.stride  # find_span, span.rs:272-274
            done = 0
            while done < a1 - a0:
                ln = min(node.stride - inst, (a1 - a0) - done)
                walk(node.spans[span], inst, inst + ln, o0 + done)
                inst, span, done = 0, span + 1, done + ln

        if stop > start:
            walk(self.cid, start, stop, 0)
        return out

    def _typed(self, stored, fbits):
        if self.dtype in (np.float32, np.float64):
            s = np.asarray(stored, dtype=np.int64)
            out = ((s - 1) / float(1 << (fbits + 1))).astype(self.dtype)
            out[s == 0] = np.nan
            return out
        return np.asarray(stored).astype(self.dtype)

    def _sub(self, node, index):
        """The object behind External(index) of a superchunk: (cid, parsed node)."""
        cid = self._resolver.node(node.external_cid)[index]
        return cid, self._resolver.node(cid)

    def _route_cell(self, cid, a0, a1, row, col, out, o0, jobs):  # superchunk.rs:356-400
        node = self._resolver.node(cid)
        if isinstance(node, Chunk):
            jobs.append((node, (a0, a1, row, col), o0, node.fractional_bits))
            return
        cs = node.chunks_sidelen
        ch = (row // cs) * node.subsidelen + col // cs
        ref = node.references[ch]
        if ref is None:  # elided: the tile's one value per instant is its max (superchunk.rs:370-373, SuperCellIter)
            stride = node.subsidelen * node.subsidelen
            out[o0:o0 + a1 - a0] = self._typed(node.max[ch + np.arange(a0, a1) * stride], node.fractional_bits)
            return
        self._route_cell(self._sub(node, ref[1])[0], a0, a1, row % cs, col % cs, out, o0, jobs)

    def _tiles(self, node, top, bottom, left, right):
        """Superchunk::subchunks_for (superchunk.rs:589-633): (tile index, local rect, slice of the window)."""
        cs, ss = node.chunks_sidelen, node.subsidelen
        for ti in range(top // cs, (bottom - 1) // cs + 1):
            for tj in range(left // cs, (right - 1) // cs + 1):
                r0, r1 = max(top, ti * cs), min(bottom, ti * cs + cs)
                c0, c1 = max(left, tj * cs), min(right, tj * cs + cs)
                yield ti * ss + tj, (r0 - ti * cs, r1 - ti * cs, c0 - tj * cs, c1 - tj * cs), (r0 - top, r1 - top, c0 - left, c1 - left)

    def _route_window(self, cid, a0, a1, top, bottom, left, right, out, jobs):  # superchunk.rs:404-470
        node = self._resolver.node(cid)
        if isinstance(node, Chunk):
            jobs.append((node, Cube(a0, a1, top, bottom, left, right), out))
            return
        stride = node.subsidelen * node.subsidelen
        for ch, (r0, r1, c0, c1), (s0, s1, t0, t1) in self._tiles(node, top, bottom, left, right):
            dst = out[:, s0:s1, t0:t1]
            ref = node.references[ch]
            if ref is None:
                dst[...] = self._typed(node.max[ch + np.arange(a0, a1) * stride], node.fractional_bits)[:, None, None]
            else:
                self._route_window(self._sub(node, ref[1])[0], a0, a1, r0, r1, c0, c1, dst, jobs)

    def _route_search(self, cid, a0, a1, top, bottom, left, right, lower, upper, dt, dr, dc_, hits):  # superchunk.rs:516-585
        node = self._resolver.node(cid)
        if isinstance(node, Chunk):
            for t, r, c in node.iter_search(Cube(a0, a1, top, bottom, left, right), lower, upper).tolist():
                hits.append((t + dt, r + dr, c + dc_))
            return
        stride, cs, ss = node.subsidelen * node.subsidelen, node.chunks_sidelen, node.subsidelen
        for ch, (r0, r1, c0, c1), _ in self._tiles(node, top, bottom, left, right):
            ref = node.references[ch]
            org_r, org_c = dr + (ch // ss) * cs, dc_ + (ch % ss) * cs
            if ref is None:
                for t in range(a0, a1):
                    if lower <= node.max[ch + t * stride] <= upper:
                        hits.extend((t + dt, org_r + r, org_c + c) for r in range(r0, r1) for c in range(c0, c1))
            else:
                self._route_search(self._sub(node, ref[1])[0], a0, a1, r0, r1, c0, c1, lower, upper, dt, org_r, org_c, hits)

    # ---- growth: Variable::append and friends (dataset.rs:834-986) -----------------------------------------------------
    def _with_cid(self, cid):
        return Variable((self.name, self._round, self.span_size, self.chunk_size, self._k2_levels, self.encoding, cid, self._resolver), _FACTORY_KEY)

    def _node_shape(self, cid):
        return self._resolver.node(cid).shape if not isinstance(self._resolver.node(cid), Chunk) else self._resolver.node(cid).shape()

    def _tail_spans(self):  # dataset.rs:959-974: the spans from the root down to the last bottom-level span
        out, span = [], self._root()
        while span.stride > self.chunk_size:
            out.append(span)
            span = self._resolver.node(span.spans[-1])
        out.append(span)
        return out

    def _tail_data(self):  # dataset.rs:937-957: the last chunk if it is not full
        tail = self._tail_spans()[-1]
        if len(tail) == 0:
            return None
        cid = tail.spans[-1]
        return cid if self._node_shape(cid)[0] < self.chunk_size else None

    def _span_append(self, span, child_cid):  # span.rs:50-92
        cshape = self._node_shape(child_cid)
        if span.spans and self._node_shape(span.spans[-1])[0] != span.stride:
            raise ValueError("Can't append to span when last subspan is not full")
        if cshape[1] != span.shape[1] or cshape[2] != span.shape[2]:
            raise ValueError("Shape of subspan (%d, %d) doesn't match shape of span (%d, %d)" % (cshape[1], cshape[2], span.shape[1], span.shape[2]))
        if cshape[0] > span.stride:
            raise ValueError("Attempt to add subspan with length (%d) greater than stride (%d)" % (cshape[0], span.stride))
        return _Span(span.encoding, [span.shape[0] + cshape[0], cshape[1], cshape[2]], span.stride, span.spans + [child_cid])

    def _span_update(self, span, child_cid):  # span.rs:96-109: replace the last subspan
        spans = span.spans[:-1]
        return self._span_append(_Span(span.encoding, [len(spans) * span.stride, span.shape[1], span.shape[2]], span.stride, spans), child_cid)

    def _save_span(self, span):
        cid = self._resolver.save(span.serialize())
        self._resolver._nodes[cid] = span
        return cid

    def _save_spans(self, spans):  # dataset.rs:976-986
        spans = list(spans)
        span = spans.pop()
        while spans:
            span = self._span_update(spans.pop(), self._save_span(span))
        return self._with_cid(self._save_span(span))

    def _create_open_span(self, shape):  # dataset.rs:880-935
        span = _Span(self.encoding, [0, shape[0], shape[1]], self.chunk_size, [])
        spans = self._tail_spans()
        left_hand = spans.pop()
        while True:
            if spans:
                parent = spans.pop()
                if len(parent) == self.span_size:  # full as well: a new parent for the new span, then one level up
                    new_parent = _Span(self.encoding, [0, shape[0], shape[1]], self.span_size * span.stride, [])
                    left_hand = parent
                    span = self._span_append(new_parent, self._save_span(span))
                else:
                    span = self._span_append(parent, self._save_span(span))
                    break
            else:  # no room anywhere: a new root, the old root moves one level down
                new_root = _Span(self.encoding, [0, shape[0], shape[1]], self.span_size * span.stride, [])
                new_root = self._span_append(new_root, self._save_span(left_hand))
                span = self._span_append(new_root, self._save_span(span))
                break
        while spans:
            span = self._span_update(spans.pop(), self._save_span(span))
        return self._with_cid(self._save_span(span))

    def _fractional_bits(self, data):  # MMBuffer3::compute_fractional_bits, mmbuffer.rs:596-613
        if data.dtype.kind != "f":
            return 0
        kind, bits = _chunk.suggest_fraction(data)  # on the GPU
        if self._round is not None:
            return min(bits, self._round)
        if kind == "round":
            raise ValueError("loss of precision: pass round=True, fractional_bits=N to add_variable (fixed.rs:58-61)")
        return bits

    def _append(self, data, update):  # dataset.rs:834-878
        variable = self
        spans = variable._tail_spans()
        instants, rows, cols = data.shape
        for start in range(0, instants, variable.chunk_size):
            end = min(start + variable.chunk_size, instants)
            buf = np.ascontiguousarray(data[start:end])
            build = Superchunk.build(buf, list(variable._k2_levels), k=2, fractional_bits=variable._fractional_bits(buf),
                                     round=variable._round is not None)  # dcdf_superchunk_build: the GPU encode
            variable._resolver.objects.update(build.objects)
            span = spans.pop()
            if span.shape[0] == variable.span_size * span.stride:  # the tail span is full: save, open a new one
                spans.append(span)
                variable = variable._save_spans(spans)
                variable = variable._create_open_span([rows, cols])
                spans = variable._tail_spans()
                span = spans.pop()
                assert len(span) == 0
            if update:
                update = False
                span = variable._span_update(span, build.cid)
            else:
                span = variable._span_append(span, build.cid)
            spans.append(span)
        return variable._save_spans(spans)


class Dataset:  # py-dcdf __init__.py:52-148 over dataset.rs:60-384
    def __init__(self, inner=None, private=None):
        if private is not _FACTORY_KEY:
            raise RuntimeError("Dataset objects come from Dataset.new(...) or Resolver.get_dataset(...)")
        (self._coordinates, self._shape, self._variables, self.prev, self.cid, self._resolver) = inner

    @classmethod
    def new(cls, coordinates, shape, resolver):
        t, y, x = coordinates
        return cls(([t, y, x], tuple(int(s) for s in shape), [], None, None, resolver), _FACTORY_KEY)

    @property
    def coordinates(self):
        return list(self._coordinates)

    @property
    def variables(self):
        return list(self._variables)

    @property
    def shape(self):
        return self._shape

    def _next(self, variables):
        prev = self.cid if self.cid is not None else self.prev  # dataset.rs:144-148
        return Dataset((self._coordinates, self._shape, variables, prev, None, self._resolver), _FACTORY_KEY)

    def add_variable(self, name, span_size, chunk_size, k2_levels, round=False, fractional_bits=0, dtype=np.float32):
        encoding = MMEncoding.from_dtype[np.dtype(dtype).type]
        span = _Span(encoding, [0, self._shape[0], self._shape[1]], chunk_size, [])  # an empty span (dataset.rs:128-129)
        cid = self._resolver.save(span.serialize())
        var = Variable((name, int(fractional_bits) if round else None, int(span_size), int(chunk_size), [int(x) for x in k2_levels],
                        encoding, cid, self._resolver), _FACTORY_KEY)
        return self._next(self._variables + [var])

    def append(self, name, data):
        data = np.asarray(data)
        if data.dtype not in (np.int32, np.int64, np.float32, np.float64):
            raise ValueError("append() takes int32, int64, float32 or float64 arrays, not %s" % data.dtype)
        var = self.get_variable(name)
        if var is None:
            raise KeyError(name)
        if data.ndim != 3 or data.dtype != var.dtype:
            raise ValueError("expected a 3-D %s array" % np.dtype(var.dtype).name)
        # the last, incomplete chunk is decoded, the new instants go behind it and the whole is encoded again
        # (dataset.rs:171-189): its first chunk_size slice REPLACES the old tail (`update`)
        tail = var._tail_data()
        if tail is not None:
            t, r, c = var._node_shape(tail)
            old = np.zeros((t, r, c), dtype=var.dtype)
            jobs = []
            var._route_window(tail, 0, t, 0, r, 0, c, old, jobs)
            if jobs:
                flat, off = _chunk.fill_window_batch([j[0] for j in jobs], [j[1] for j in jobs], dtype=var.dtype)
                for (ch, cu, dst), o in zip(jobs, off):
                    dst[...] = flat[int(o):int(o) + dst.size].reshape(dst.shape)
            new = var._append(np.concatenate([old, data]), True)
        else:
            new = var._append(data, False)
        return self._next([new if v.name == name else v for v in self._variables])

    def commit(self):
        desc = repr((self._shape, [(c.name, c.kind, c.start, c.step, c.steps, np.dtype(c._dtype).name if c.kind != "time" else "time") for c in self._coordinates],
                     [(v.name, v._round, v.span_size, v.chunk_size, v._k2_levels, v.encoding, v.cid.hex()) for v in self._variables],
                     self.prev.hex() if self.prev else None)).encode()
        cid = _cid(desc)
        self._resolver._datasets[cid] = Dataset((self._coordinates, self._shape, self._variables, self.prev, cid, self._resolver), _FACTORY_KEY)
        self.cid = cid
        return cid

    def get_coordinate(self, name):
        return next((c for c in self._coordinates if c.name == name), None)

    def get_variable(self, name):
        return next((v for v in self._variables if v.name == name), None)

    def __getattr__(self, name):
        if not name.startswith("_"):
            for c in self.__dict__.get("_coordinates", []):
                if c.name == name:
                    return c
            for v in self.__dict__.get("_variables", []):
                if v.name == name:
                    return v
        raise AttributeError(name)


class _Lazy:
    """What indexing a Variable returns: nothing is decoded until `.data` (or an index into it) is asked for, then once."""

    def __init__(self, decode):
        self._decode = decode
        self._value = None
        self._done = False

    @property
    def data(self):
        if not self._done:
            self._value, self._done, self._decode = self._decode(), True, None
        return self._value

    def __getitem__(self, key):
        return self.data[key]
