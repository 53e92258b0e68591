"""Host-side mirror of the reference's `Chunk` interface (dcdf/src/chunk.rs) over the MI355X library.

Same names, argument meaning and error behaviour as the reference for this path:
  Chunk.build(buffer, shape, k)      chunk.rs:42     -> MMStruct3Build (mmstruct.rs:24-34)
  Chunk.get / fill_cell / fill_window / iter_search   chunk.rs:127,135,152,213
  Chunk.shape / size / write_to / read_from           chunk.rs:119,272,235,247
  geom.Rect / geom.Cube (auto-reordered bounds)       geom.rs
Panics of the reference surface as DcdfError (negative code)."""
import ctypes as C

import numpy as np

from . import _lib as L

_ENC = {np.dtype("int32"): L.DCDF_I32, np.dtype("int64"): L.DCDF_I64, np.dtype("float32"): L.DCDF_F32,
        np.dtype("float64"): L.DCDF_F64}
_DT = {L.DCDF_I32: np.int32, L.DCDF_I64: np.int64, L.DCDF_F32: np.float32, L.DCDF_F64: np.float64}


class Rect:  # geom.rs:4-41
    def __init__(self, top, bottom, left, right):
        self.top, self.bottom = (top, bottom) if top <= bottom else (bottom, top)
        self.left, self.right = (left, right) if left <= right else (right, left)

    def rows(self):
        return self.bottom - self.top

    def cols(self):
        return self.right - self.left


class Cube:  # geom.rs:71-120
    def __init__(self, start, end, top, bottom, left, right):
        self.start, self.end = (start, end) if start <= end else (end, start)
        self.top, self.bottom = (top, bottom) if top <= bottom else (bottom, top)
        self.left, self.right = (left, right) if left <= right else (right, left)

    def instants(self):
        return self.end - self.start

    def rows(self):
        return self.bottom - self.top

    def cols(self):
        return self.right - self.left

    def rect(self):
        return Rect(self.top, self.bottom, self.left, self.right)

    def _c(self):
        return L.Cube(self.start, self.end, self.top, self.bottom, self.left, self.right)


class MMStruct3Build:  # mmstruct.rs:24-34
    def __init__(self, data, size, snapshots, logs, minmax=None):
        self.data = data
        self.size = size
        self.elided = 0
        self.local = 0
        self.external = 0
        self.snapshots = snapshots
        self.logs = logs
        self.minmax = minmax  # [instants, 2] stored (min, max) per instant


def _desc(a, fractional_bits, round_):
    if a.ndim != 3 or a.dtype not in _ENC:
        raise ValueError("expected a 3-D int32/int64/float32/float64 array")
    d = L.TileDesc()
    d.base = a.ctypes.data
    d.dtype = _ENC[a.dtype]
    d.stride_t, d.stride_r, d.stride_c = [s // a.itemsize for s in a.strides]
    d.instants, d.rows, d.cols = a.shape
    d.fractional_bits = int(fractional_bits)
    d.round = 1 if round_ else 0
    return d


def build_batch(arrays, k=2, fractional_bits=0, round=False):
    """Chunk::build for many independent tiles in one GPU launch.  Returns a list of MMStruct3Build
    (or DcdfError instances for tiles the reference would have panicked on)."""
    arrays = [np.asarray(a) for a in arrays]
    n = len(arrays)
    descs = (L.TileDesc * n)()
    fb = fractional_bits if isinstance(fractional_bits, (list, tuple)) else [fractional_bits] * n
    for i, a in enumerate(arrays):
        descs[i] = _desc(a, fb[i], round)
    out = C.POINTER(L.Encoded)()
    L.check(L.lib().dcdf_chunk_build_batch(descs, C.c_size_t(n), int(k), L.MEM_HOST, C.byref(out)), "chunk_build")
    res = []
    try:
        for i in range(n):
            e = out[i]
            if e.status != 0:
                res.append(L.DcdfError(e.status, "Chunk::build"))
                continue
            data = C.string_at(e.bytes, e.len)
            mm = np.ctypeslib.as_array(e.minmax, shape=(arrays[i].shape[0], 2)).copy()
            res.append(MMStruct3Build(Chunk(data, lazy=True), e.len, e.snapshots, e.logs, mm))
    finally:
        L.lib().dcdf_free_encoded(out, C.c_size_t(n))
    return res


class Chunk:
    """An encoded chunk (the serialized `Chunk::write_to` image) opened on the GPU."""

    def __init__(self, data, lazy=False):
        """`lazy`: upload + parse on the GPU at the first query instead of now (what `build_batch` hands out: a freshly
        built chunk is usually written out, not queried)."""
        self._bytes = bytes(data)
        self._handle = None
        if not lazy:
            self._open()

    @classmethod
    def open_batch_device(cls, ptrs, lens, fetch=None):
        """dcdf_chunk_open_batch(DCDF_MEM_DEVICE): ptrs / lens = ctypes arrays of device pointers and byte counts of encoded
        chunks already in HBM.  `fetch(j)` (optional) returns chunk j's bytes for write_to() -- they are only copied to the
        host when asked for."""
        n = len(ptrs)
        hs = (C.c_void_p * n)()
        st = (C.c_int32 * n)()
        L.check(L.lib().dcdf_chunk_open_batch(ptrs, lens, C.c_size_t(n), L.MEM_DEVICE, hs, st), "chunk_open_batch")
        bad = next((j for j in range(n) if st[j] != 0), None)
        if bad is not None:  # every handle the call did hand out goes back (the chunks behind `bad` too: they share the slab)
            for j in range(n):
                if hs[j]:
                    L.lib().dcdf_chunk_close(C.c_void_p(hs[j]))
            raise L.DcdfError(st[bad], "chunk_open_batch: chunk %d" % bad)
        out = []
        for j in range(n):
            c = cls.__new__(cls)
            c._bytes = None
            c._fetch = (lambda j=j: fetch(j)) if fetch else None
            c._handle = C.c_void_p(hs[j])
            c._shape = None  # (dcdf_chunk_info on first use: thousands of handles are opened at once)
            out.append(c)
        return out

    def _open(self):
        h = C.c_void_p()
        L.check(L.lib().dcdf_chunk_open(self._bytes, C.c_size_t(len(self._bytes)), C.byref(h)), "Chunk::read_from")
        self._handle = h
        self._read_info()

    def _read_info(self):
        h = self._handle
        shp = (C.c_uint32 * 3)()
        enc, fb, nb = C.c_int32(), C.c_uint32(), C.c_uint32()
        L.check(L.lib().dcdf_chunk_info(h, shp, C.byref(enc), C.byref(fb), C.byref(nb)))
        self._shape = (int(shp[0]), int(shp[1]), int(shp[2]))
        self._encoding = int(enc.value)
        self._fractional_bits = int(fb.value)
        self._n_blocks = int(nb.value)

    @property
    def _h(self):
        if self._handle is None:
            self._open()
        return self._handle

    def _info(self):
        if self._handle is None:
            self._open()
        elif self._shape is None:
            self._read_info()

    @property
    def encoding(self):
        self._info()
        return self._encoding

    @property
    def fractional_bits(self):
        self._info()
        return self._fractional_bits

    @property
    def n_blocks(self):
        self._info()
        return self._n_blocks

    # -- construction --------------------------------------------------------------------------
    @staticmethod
    def build(buffer, shape=None, k=2, fractional_bits=0, round=False):
        """chunk.rs:42.  `buffer`: ndarray[instants, rows, cols] (any strides)."""
        a = np.asarray(buffer)
        if shape is not None and tuple(shape) != a.shape:
            raise ValueError("shape does not match buffer")
        r = build_batch([a], k=k, fractional_bits=fractional_bits, round=round)[0]
        if isinstance(r, Exception):
            raise r
        return r

    @classmethod
    def read_from(cls, data):  # chunk.rs:247
        return cls(data)

    def write_to(self):  # chunk.rs:235
        if self._bytes is None:  # opened from device memory: the bytes come to the host only now
            if getattr(self, "_fetch", None) is None:
                raise ValueError("this chunk was opened from device memory without a way to fetch its bytes")
            self._bytes = self._fetch()
        return self._bytes

    def size(self):  # chunk.rs:272
        return len(self.write_to())

    def shape(self):  # chunk.rs:119
        self._info()
        return list(self._shape)

    def close(self):
        if getattr(self, "_handle", None):
            L.lib().dcdf_chunk_close(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- queries -------------------------------------------------------------------------------
    def get(self, instant, row, col):  # chunk.rs:127 (stored i64; use cell()/window() for typed values)
        v = C.c_int64()
        L.check(L.lib().dcdf_chunk_get(self._h, C.c_uint32(instant), C.c_uint32(row), C.c_uint32(col), C.byref(v)),
                "Chunk::get")
        return v.value

    def fill_cell(self, start, end, row, col):  # chunk.rs:135
        n = abs(end - start)
        out = np.zeros(n, dtype=np.int64)
        L.check(L.lib().dcdf_chunk_fill_cell(self._h, C.c_uint32(start), C.c_uint32(end), C.c_uint32(row),
                                             C.c_uint32(col), C.c_void_p(out.ctypes.data)), "Chunk::fill_cell")
        return out

    def fill_window(self, bounds, dtype=None, out=None):  # chunk.rs:152
        dtype = np.dtype(dtype or _DT[self.encoding])
        if out is None:
            out = np.zeros((bounds.instants(), bounds.rows(), bounds.cols()), dtype=dtype)
        st = [s // out.itemsize for s in out.strides]
        c = bounds._c()
        L.check(L.lib().dcdf_chunk_fill_window(self._h, C.byref(c), C.c_void_p(out.ctypes.data), _ENC[out.dtype],
                                               C.c_int64(st[0]), C.c_int64(st[1]), C.c_int64(st[2])), "Chunk::fill_window")
        return out

    def iter_search(self, bounds, lower, upper):  # chunk.rs:213 -> (instant,row,col) triples, sorted
        c = bounds._c()
        n = C.c_size_t()
        cap = 1 << 16
        while True:
            out = np.zeros((cap, 3), dtype=np.uint32)
            rc = L.lib().dcdf_chunk_search(self._h, C.byref(c), C.c_int64(lower), C.c_int64(upper),
                                           C.c_void_p(out.ctypes.data), C.c_size_t(cap), C.byref(n))
            if rc == -11:  # DCDF_ERR_CAPACITY
                cap = n.value
                continue
            L.check(rc, "Chunk::iter_search")
            return out[:n.value]


def _handles(chunks):
    return (C.c_void_p * len(chunks))(*[c._h for c in chunks])


def get_batch(chunks, points, out_device_ptr=None):
    """Chunk::get for many (chunk, (instant, row, col)) pairs in one launch (dcdf_query_get_batch; what Superchunk::get,
    superchunk.rs:313-352, routes).  Returns int64[n] stored values, or writes them to device memory at out_device_ptr."""
    pts = np.ascontiguousarray(np.asarray(points, dtype=np.uint32).reshape(-1, 3))
    n = len(pts)
    out = None if out_device_ptr else np.zeros(n, dtype=np.int64)
    L.check(L.lib().dcdf_query_get_batch(_handles(chunks), C.c_void_p(pts.ctypes.data), C.c_size_t(n),
                                         C.c_void_p(out_device_ptr or out.ctypes.data), L.MEM_DEVICE if out_device_ptr else L.MEM_HOST,
                                         None), "get_batch")
    return out


def fill_cell_batch(chunks, cells):
    """Chunk::fill_cell for many (chunk, (start, end, row, col)) in one launch (superchunk.rs:356-400).  Returns the list of
    int64 series."""
    cl = np.ascontiguousarray(np.asarray(cells, dtype=np.uint32).reshape(-1, 4))
    n = len(cl)
    ln = np.abs(cl[:, 1].astype(np.int64) - cl[:, 0].astype(np.int64)).astype(np.uint64)
    off = np.zeros(n, dtype=np.uint64)
    if n > 1:
        off[1:] = np.cumsum(ln)[:-1]
    out = np.zeros(max(1, int(ln.sum())), dtype=np.int64)
    L.check(L.lib().dcdf_query_fill_cell_batch(_handles(chunks), C.c_void_p(cl.ctypes.data), C.c_size_t(n), C.c_void_p(out.ctypes.data),
                                               C.c_void_p(off.ctypes.data), L.MEM_HOST, None), "fill_cell_batch")
    return [out[int(off[i]):int(off[i]) + int(ln[i])] for i in range(n)]


def fill_window_batch(chunks, cubes, dtype=np.int64, out_device_ptr=None, out_offset=None):
    """fill_window of many (chunk, Cube) pairs in one launch with a typed result (dcdf_query_fill_window_batch_typed).  Host
    form: returns (flat ndarray of dtype, offsets uint64[n]) with window q at flat[offsets[q]:]; device form
    (out_device_ptr): the kernel writes window q at element out_offset[q] of the caller's device array, returns kernel ms."""
    dtype = np.dtype(dtype)
    cub = (L.Cube * len(cubes))(*[c._c() for c in cubes])
    vol = np.array([c.instants() * c.rows() * c.cols() for c in cubes], dtype=np.uint64)
    ms = C.c_float()
    if out_device_ptr is None:
        off = np.zeros(len(cubes), dtype=np.uint64)
        if len(cubes) > 1:
            off[1:] = np.cumsum(vol)[:-1]
        out = np.zeros(max(1, int(vol.sum())), dtype=dtype)
        L.check(L.lib().dcdf_query_fill_window_batch_typed(_handles(chunks), cub, C.c_size_t(len(cubes)), C.c_void_p(out.ctypes.data),
                                                           _ENC[dtype], L.MEM_HOST, C.c_void_p(off.ctypes.data), C.byref(ms)),
                "fill_window_batch")
        return out, off
    off = np.ascontiguousarray(np.asarray(out_offset, dtype=np.uint64))
    L.check(L.lib().dcdf_query_fill_window_batch_typed(_handles(chunks), cub, C.c_size_t(len(cubes)), C.c_void_p(out_device_ptr),
                                                       _ENC[dtype], L.MEM_DEVICE, C.c_void_p(off.ctypes.data), C.byref(ms)),
            "fill_window_batch")
    return ms.value


# py-dcdf flavoured helpers (SURVEY 8b "Python shape")
def build_chunk(array, k=2, fractional_bits=0, round=False):
    return Chunk.build(array, k=k, fractional_bits=fractional_bits, round=round).data.write_to()


def window(data, t0, t1, r0, r1, c0, c1, dtype=None):
    return Chunk(data).fill_window(Cube(t0, t1, r0, r1, c0, c1), dtype=dtype)


def search(data, t0, t1, r0, r1, c0, c1, lower, upper):
    return Chunk(data).iter_search(Cube(t0, t1, r0, r1, c0, c1), lower, upper)


def suggest_fraction(data):
    """fixed.rs:96-159 (`suggest_fraction`, used per buffer by mmbuffer.rs:596-613): for a float32/float64 array
    [instants, rows, cols] returns ("precise", bits) -- the fewest fractional bits that store every value exactly --
    or ("round", bits) when even the widest shift leaves a fraction.  Computed on the GPU."""
    a = np.asarray(data)
    if a.ndim != 3 or a.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
        raise ValueError("expected a 3-D float32/float64 array")
    d = _desc(a, 0, False)
    rnd, bits = C.c_int32(), C.c_int32()
    L.check(L.lib().dcdf_suggest_fraction(C.byref(d), L.MEM_HOST, C.byref(rnd), C.byref(bits)), "suggest_fraction")
    return ("round" if rnd.value else "precise", int(bits.value))
