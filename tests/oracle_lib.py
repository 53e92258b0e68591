"""ctypes loader for the CPU oracle (oracle/_build/libk2r_oracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "_build", "libk2r_oracle.so")

ENC = {np.dtype("int32"): 4, np.dtype("int64"): 8, np.dtype("float32"): 32, np.dtype("float64"): 64}
DT = {4: np.int32, 8: np.int64, 32: np.float32, 64: np.float64}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
        _lib = C.CDLL(_SO)
        _lib.orc_from_fixed_f32.restype = C.c_float
        _lib.orc_from_fixed_f64.restype = C.c_double
        _lib.orc_sidelen.restype = C.c_uint64
        _lib.orc_free.argtypes = [C.c_void_p]
        _lib.orc_chunk_close.argtypes = [C.c_void_p]
    return _lib


_native = None


def native_lib():
    """The oracle rebuilt -O3 -march=native on THIS host (timing only: bench.py's cpu_baseline); None if that fails."""
    global _native
    if _native is None:
        try:
            import tempfile
            out = tempfile.mkdtemp(prefix="k2r_oracle_native_")  # never in-tree: a -march=native .so must not travel to another host
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native", "OUT=" + out],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            _native = C.CDLL(os.path.join(out, "libk2r_oracle_native.so"))
        except Exception:
            _native = False
    return _native or None


class OracleError(Exception):
    def __init__(self, code):
        super().__init__("oracle error %d" % code)
        self.code = code


def _check(rc):
    if rc != 0:
        raise OracleError(rc)


def _take_i64(ptr, n):
    arr = np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].copy()
    lib().orc_free(ptr)
    return [int(x) for x in arr]


class _Cursor:
    def __init__(self, vals):
        self.v, self.i = vals, 0

    def one(self):
        x = self.v[self.i]
        self.i += 1
        return x

    def vec(self):
        n = self.one()
        out = self.v[self.i:self.i + n]
        self.i += n
        return out

    def bitmap(self):
        length = self.one()
        words = self.vec()
        index = self.vec()
        return {"length": length, "words": words, "index": index}


def _elem_strides(a):
    return [s // a.itemsize for s in a.strides]


def chunk_build(a, k=2, fractional_bits=0, round_=False, want_snapshots=False):
    """Chunk::build + write_to on a 3-D numpy array (any strides).  Returns bytes (and stats)."""
    a = np.asarray(a)
    assert a.ndim == 3
    enc = ENC[a.dtype]
    st = _elem_strides(a)
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    ns, nl = C.c_uint32(), C.c_uint32()
    si = (C.c_uint32 * max(1, a.shape[0]))()
    rc = lib().orc_chunk_build(C.c_void_p(a.ctypes.data), enc, C.c_int64(st[0]), C.c_int64(st[1]), C.c_int64(st[2]),
                               C.c_uint32(a.shape[0]), C.c_uint32(a.shape[1]), C.c_uint32(a.shape[2]), k,
                               int(fractional_bits), int(bool(round_)), C.byref(out), C.byref(n), C.byref(ns),
                               C.byref(nl), si)
    _check(rc)
    data = C.string_at(out, n.value)
    lib().orc_free(out)
    if want_snapshots:
        return data, ns.value, nl.value, list(si[:ns.value])
    return data


def chunk_build_forced(a, k=2, block_len=4):
    a = np.ascontiguousarray(a, dtype=np.int64)
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    _check(lib().orc_chunk_build_forced(C.c_void_p(a.ctypes.data), C.c_uint32(a.shape[0]), C.c_uint32(a.shape[1]),
                                        C.c_uint32(a.shape[2]), k, C.c_uint32(block_len), C.byref(out), C.byref(n)))
    data = C.string_at(out, n.value)
    lib().orc_free(out)
    return data


class Chunk:
    """Opened (deserialized) chunk on the oracle."""

    def __init__(self, data):
        self._buf = bytes(data)
        self._h = C.c_void_p()
        _check(lib().orc_chunk_open(self._buf, C.c_size_t(len(self._buf)), C.byref(self._h)))
        info = (C.c_uint64 * 7)()
        _check(lib().orc_chunk_info(self._h, info))
        self.shape = (int(info[0]), int(info[1]), int(info[2]))
        self.encoding = int(info[3])
        self.fractional_bits = int(info[4])
        self.n_blocks = int(info[5])
        self.size = int(info[6])

    def close(self):
        if self._h:
            lib().orc_chunk_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def block_lengths(self):
        out = (C.c_uint32 * max(1, self.n_blocks))()
        _check(lib().orc_chunk_block_lengths(self._h, out))
        return list(out[:self.n_blocks])

    def serialize(self):
        out = C.POINTER(C.c_uint8)()
        n = C.c_size_t()
        _check(lib().orc_chunk_serialize(self._h, C.byref(out), C.byref(n)))
        data = C.string_at(out, n.value)
        lib().orc_free(out)
        return data

    def get(self, instant, row, col):
        v = C.c_int64()
        _check(lib().orc_chunk_get(self._h, C.c_uint32(instant), C.c_uint32(row), C.c_uint32(col), C.byref(v)))
        return v.value

    def fill_cell(self, start, end, row, col):
        out = np.zeros(max(0, end - start), dtype=np.int64)
        _check(lib().orc_chunk_fill_cell(self._h, C.c_uint32(start), C.c_uint32(end), C.c_uint32(row), C.c_uint32(col),
                                         C.c_void_p(out.ctypes.data)))
        return out

    def fill_window(self, start, end, top, bottom, left, right, dtype=None):
        dtype = np.dtype(dtype or DT[self.encoding])
        out = np.zeros((abs(end - start), abs(bottom - top), abs(right - left)), dtype=dtype)
        st = _elem_strides(out)
        _check(lib().orc_chunk_fill_window(self._h, C.c_uint32(start), C.c_uint32(end), C.c_uint32(top),
                                           C.c_uint32(bottom), C.c_uint32(left), C.c_uint32(right),
                                           C.c_void_p(out.ctypes.data), ENC[dtype], C.c_int64(st[0]), C.c_int64(st[1]),
                                           C.c_int64(st[2])))
        return out

    def search(self, start, end, top, bottom, left, right, lower, upper):
        out = C.POINTER(C.c_uint32)()
        n = C.c_size_t()
        _check(lib().orc_chunk_search(self._h, C.c_uint32(start), C.c_uint32(end), C.c_uint32(top), C.c_uint32(bottom),
                                      C.c_uint32(left), C.c_uint32(right), C.c_int64(lower), C.c_int64(upper),
                                      C.byref(out), C.byref(n)))
        arr = np.ctypeslib.as_array(out, shape=(max(1, n.value) * 3,))[:n.value * 3].copy().reshape(-1, 3)
        lib().orc_free(out)
        return arr


def snapshot_dump(a2, k=2):
    a2 = np.ascontiguousarray(a2, dtype=np.int64)
    out = C.POINTER(C.c_int64)()
    n = C.c_size_t()
    _check(lib().orc_snapshot_dump(C.c_void_p(a2.ctypes.data), C.c_uint32(a2.shape[0]), C.c_uint32(a2.shape[1]), k,
                                   C.byref(out), C.byref(n)))
    c = _Cursor(_take_i64(out, n.value))
    return {"nodemap": c.bitmap(), "max": c.vec(), "min": c.vec(), "size": c.one(), "serialized_len": c.one(),
            "sidelen": c.one()}


def log_dump(s2, t2, k=2):
    s2 = np.ascontiguousarray(s2, dtype=np.int64)
    t2 = np.ascontiguousarray(t2, dtype=np.int64)
    out = C.POINTER(C.c_int64)()
    n = C.c_size_t()
    _check(lib().orc_log_dump(C.c_void_p(s2.ctypes.data), C.c_void_p(t2.ctypes.data), C.c_uint32(s2.shape[0]),
                              C.c_uint32(s2.shape[1]), k, C.byref(out), C.byref(n)))
    c = _Cursor(_take_i64(out, n.value))
    return {"nodemap": c.bitmap(), "equal": c.bitmap(), "max": c.vec(), "min": c.vec(), "size": c.one(),
            "serialized_len": c.one(), "sidelen": c.one()}


def sl_get(s2, t2, which, row, col, k=2):
    s2 = np.ascontiguousarray(s2, dtype=np.int64)
    t2 = np.ascontiguousarray(t2, dtype=np.int64)
    v = C.c_int64()
    _check(lib().orc_sl_get(C.c_void_p(s2.ctypes.data), C.c_void_p(t2.ctypes.data), C.c_uint32(s2.shape[0]),
                            C.c_uint32(s2.shape[1]), k, which, C.c_uint32(row), C.c_uint32(col), C.byref(v)))
    return v.value


def sl_exhaustive(s2, t2, which, k=2, lo=4, hi=9):
    s2 = np.ascontiguousarray(s2, dtype=np.int64)
    t2 = np.ascontiguousarray(t2, dtype=np.int64)
    r = [C.c_uint64() for _ in range(4)]
    _check(lib().orc_sl_exhaustive(C.c_void_p(s2.ctypes.data), C.c_void_p(t2.ctypes.data), C.c_uint32(s2.shape[0]),
                                   C.c_uint32(s2.shape[1]), k, which, C.c_int64(lo), C.c_int64(hi), *[C.byref(x) for x in r]))
    return {"windows": r[0].value, "bad_window_cells": r[1].value, "bad_searches": r[2].value, "bad_gets": r[3].value}


def bitmap_dump(length, data):
    b = bytes(data)
    out = C.POINTER(C.c_int64)()
    n = C.c_size_t()
    _check(lib().orc_bitmap_dump(C.c_uint64(length), b, C.c_size_t(len(b)), C.byref(out), C.byref(n)))
    c = _Cursor(_take_i64(out, n.value))
    bm = c.bitmap()
    bm["size"] = c.one()
    return bm


def bitmap_push_dump(bits):
    b = bytes(int(bool(x)) for x in bits)
    out = C.POINTER(C.c_int64)()
    n = C.c_size_t()
    _check(lib().orc_bitmap_push_dump(b, C.c_size_t(len(b)), C.byref(out), C.byref(n)))
    c = _Cursor(_take_i64(out, n.value))
    bm = c.bitmap()
    bm["size"] = c.one()
    return bm


def bitmap_rank(length, data, i):
    b = bytes(data)
    r, nv, bit = C.c_uint64(), C.c_uint64(), C.c_int()
    _check(lib().orc_bitmap_rank(C.c_uint64(length), b, C.c_size_t(len(b)), C.c_uint64(i), C.byref(r), C.byref(nv),
                                 C.byref(bit)))
    return r.value, nv.value, bit.value


def dac_dump(values):
    v = np.ascontiguousarray(values, dtype=np.int64)
    out = C.POINTER(C.c_int64)()
    n = C.c_size_t()
    _check(lib().orc_dac_dump(C.c_void_p(v.ctypes.data), C.c_size_t(len(v)), C.byref(out), C.byref(n)))
    c = _Cursor(_take_i64(out, n.value))
    levels = []
    for _ in range(c.one()):
        bm = c.bitmap()
        levels.append({"bitmap": bm, "bytes": c.vec()})
    return {"levels": levels, "collect": c.vec(), "size": c.one(), "serialized_len": c.one(), "collect_reread": c.vec()}


def dac_serialize(values):
    """Dac::from(values) serialized (dac.rs:37-44)."""
    v = np.ascontiguousarray(values, dtype=np.int64)
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    _check(lib().orc_dac_serialize(C.c_void_p(v.ctypes.data), C.c_size_t(len(v)), C.byref(out), C.byref(n)))
    data = C.string_at(out, n.value)
    lib().orc_free(out)
    return data


def to_fixed(v, bits, round_, ftype="f64"):
    out = C.c_int64()
    if ftype == "f32":
        rc = lib().orc_to_fixed_f32(C.c_float(v), bits, int(round_), C.byref(out))
    else:
        rc = lib().orc_to_fixed_f64(C.c_double(v), bits, int(round_), C.byref(out))
    _check(rc)
    return out.value


def from_fixed(v, bits, ftype="f64"):
    if ftype == "f32":
        return lib().orc_from_fixed_f32(C.c_int64(v), bits)
    return lib().orc_from_fixed_f64(C.c_int64(v), bits)


def suggest_fraction(data, ftype="f64"):
    a = np.ascontiguousarray(data, dtype=np.float32 if ftype == "f32" else np.float64).ravel()
    rnd, bits = C.c_int(), C.c_int()
    fn = lib().orc_suggest_fraction_f32 if ftype == "f32" else lib().orc_suggest_fraction_f64
    _check(fn(C.c_void_p(a.ctypes.data), C.c_size_t(a.size), C.byref(rnd), C.byref(bits)))
    return bool(rnd.value), bits.value


def sidelen(rows, cols, k=2):
    return int(lib().orc_sidelen(C.c_uint32(rows), C.c_uint32(cols), k))


def bench_build(a4, k=2, native=False):
    """a4: [n_chunks, instants, rows, cols] contiguous.  Returns (seconds, total_bytes, fnv).  native: use the
    -march=native build of this host when it can be made (same code, same results)."""
    a4 = np.ascontiguousarray(a4)
    sec, tb, h = C.c_double(), C.c_uint64(), C.c_uint64()
    L = (native_lib() if native else None) or lib()
    _check(L.orc_bench_build(C.c_void_p(a4.ctypes.data), ENC[a4.dtype], C.c_uint32(a4.shape[0]),
                                 C.c_uint32(a4.shape[1]), C.c_uint32(a4.shape[2]), C.c_uint32(a4.shape[3]), k,
                                 C.byref(sec), C.byref(tb), C.byref(h)))
    return sec.value, tb.value, h.value


def bench_queries(chunks, cubes, lower=None, upper=None, threads=1, native=False):
    """Timed batch of chunk-level queries on the oracle (orc_bench_queries): chunks = [oracle Chunk] per query, cubes = uint32[n, 6];
    lower/upper given -> iter_search, else fill_window.  Returns (seconds, decoded cells or hits)."""
    n = len(chunks)
    hs = (C.c_void_p * n)(*[c._h for c in chunks])
    cub = np.ascontiguousarray(np.asarray(cubes, dtype=np.uint32).reshape(n, 6))
    search = lower is not None
    lo = np.ascontiguousarray(np.asarray(lower if search else np.zeros(n), dtype=np.int64))
    hi = np.ascontiguousarray(np.asarray(upper if search else np.zeros(n), dtype=np.int64))
    sec, work = C.c_double(), C.c_uint64()
    Lb = (native_lib() if native else None) or lib()
    _check(Lb.orc_bench_queries(hs, C.c_void_p(cub.ctypes.data), C.c_void_p(lo.ctypes.data), C.c_void_p(hi.ctypes.data), C.c_size_t(n),
                                int(search), int(threads), C.byref(sec), C.byref(work)))
    return sec.value, work.value
