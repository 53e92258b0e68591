"""Device-resident encode session (dcdf_encoder_* of include/dcdf_k2r.h): tiles already in HBM,
encoded bytes stay in HBM.  Used by bench.py and the full-size GPU tests."""
import ctypes as C

import numpy as np

from . import _lib as L


class Encoder:
    def __init__(self, descs, k=2, out_cap_per_tile=0):
        """descs: list of (device_ptr, dtype_code, (st, sr, sc), (instants, rows, cols)[, fractional_bits, round])"""
        n = len(descs)
        self.n = n
        self._descs = (L.TileDesc * n)()
        for i, desc in enumerate(descs):
            ptr, dt, st, shp = desc[:4]
            d = self._descs[i]
            d.fractional_bits = desc[4] if len(desc) > 4 else 0
            d.round = 1 if (len(desc) > 5 and desc[5]) else 0
            d.base = ptr
            d.dtype = dt
            d.stride_t, d.stride_r, d.stride_c = st
            d.instants, d.rows, d.cols = shp
        self._h = C.c_void_p()
        L.check(L.lib().dcdf_encoder_create(self._descs, C.c_size_t(n), int(k), C.c_size_t(out_cap_per_tile),
                                            C.byref(self._h)), "encoder_create")

    def run(self):
        ms = C.c_float()
        L.check(L.lib().dcdf_encoder_run(self._h, C.byref(ms)), "encoder_run")
        return ms.value

    def result(self, i):
        st, ln, ns, nl = C.c_int32(), C.c_uint64(), C.c_uint32(), C.c_uint32()
        L.check(L.lib().dcdf_encoder_result(self._h, C.c_size_t(i), C.byref(st), C.byref(ln), C.byref(ns), C.byref(nl), None))
        return st.value, ln.value, ns.value, nl.value

    def device_bytes(self, i):
        """(device pointer, length) of tile i's Chunk::write_to bytes where the encoder left them in HBM."""
        st, ln = C.c_int32(), C.c_uint64()
        p = C.POINTER(C.c_uint8)()
        L.check(L.lib().dcdf_encoder_result(self._h, C.c_size_t(i), C.byref(st), C.byref(ln), None, None, C.byref(p)))
        if st.value != 0:
            raise L.DcdfError(st.value, "tile %d" % i)
        return C.cast(p, C.c_void_p).value, ln.value

    def open_chunks(self, indices=None):
        """dcdf_chunk_open_batch over this session's device buffers: the encoded chunks become queryable without ever
        leaving HBM (parsed on the device, all side-16 tables in one launch).  Returns [dcdf_amd.Chunk]."""
        from .chunk import Chunk
        idx = list(range(self.n)) if indices is None else list(indices)
        ptrs = (C.c_void_p * len(idx))()
        lens = (C.c_uint64 * len(idx))()
        for j, i in enumerate(idx):
            ptrs[j], lens[j] = self.device_bytes(i)
        return Chunk.open_batch_device(ptrs, lens, fetch=lambda j: self.fetch(idx[j]))

    def fetch(self, i):
        st, ln, _, _ = self.result(i)
        if st != 0:
            raise L.DcdfError(st, "tile %d" % i)
        buf = np.zeros(ln, dtype=np.uint8)
        L.check(L.lib().dcdf_encoder_fetch(self._h, C.c_size_t(i), C.c_void_p(buf.ctypes.data), C.c_size_t(ln)))
        return buf.tobytes()

    def gather(self):
        """Host-side gather of this GPU's results (dcdf_encoder_gather): returns (buf uint8, offsets uint64[n],
        lens uint64[n], minmax int64[sum instants, 2]); tile i's Chunk::write_to bytes are buf[offsets[i]:offsets[i]+lens[i]]."""
        nb, nm = C.c_uint64(), C.c_uint64()
        L.check(L.lib().dcdf_encoder_gather_size(self._h, C.byref(nb), C.byref(nm)), "gather_size")
        buf = np.empty(max(1, nb.value), dtype=np.uint8)
        offs = np.zeros(self.n, dtype=np.uint64)
        lens = np.zeros(self.n, dtype=np.uint64)
        mm = np.zeros((max(1, nm.value) // 2, 2), dtype=np.int64)
        L.check(L.lib().dcdf_encoder_gather(self._h, C.c_void_p(buf.ctypes.data), C.c_size_t(buf.size),
                                            C.c_void_p(offs.ctypes.data), C.c_void_p(lens.ctypes.data),
                                            C.c_void_p(mm.ctypes.data)), "gather")
        return buf, offs, lens, mm

    def object_sha256(self):
        """SHA-256 of the stored object of every tile (the reference's 8-byte object header + chunk bytes), hashed on
        the device; returns (digests[n, 32] uint8, kernel_ms)."""
        out = np.zeros((self.n, 32), dtype=np.uint8)
        ms = C.c_float()
        L.check(L.lib().dcdf_encoder_object_sha256(self._h, C.c_void_p(out.ctypes.data), C.byref(ms)), "object_sha256")
        return out, ms.value

    def object_cids(self):
        """Binary CIDs of the stored objects as the reference's MemoryMapper names them (testing.rs:172-183):
        CIDv1, codec 0x12, multihash sha2-256 -> 36 bytes each; None for failed tiles."""
        dig, _ = self.object_sha256()
        return [None if self.result(i)[0] != 0 else bytes([0x01, 0x12, 0x12, 0x20]) + dig[i].tobytes() for i in range(self.n)]

    def total_bytes(self):
        return int(L.lib().dcdf_encoder_total_bytes(self._h))

    def close(self):
        if getattr(self, "_h", None):
            L.lib().dcdf_encoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def synth_fill(dev_ptr, dtype_code, seed, t0, t1, r0, r1, c0, c1):
    from . import synth
    tab = np.ascontiguousarray(synth._COS.astype(np.int32))
    L.check(L.lib().dcdf_synth_fill(C.c_void_p(dev_ptr), C.c_int32(dtype_code), C.c_uint64(seed), C.c_int64(t0),
                                    C.c_int64(t1), C.c_int64(r0), C.c_int64(r1), C.c_int64(c0), C.c_int64(c1),
                                    C.c_void_p(tab.ctypes.data)), "synth_fill")


class DeviceBuffer:
    """A plain device allocation through the C ABI (dcdf_device_alloc): lets device-resident sessions run without torch."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        L.check(L.lib().dcdf_device_alloc(C.c_size_t(self.nbytes), C.byref(p)), "device_alloc")
        self.ptr = p.value

    def read(self, offset, nbytes, dtype=np.uint8):
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        L.check(L.lib().dcdf_device_copy(C.c_void_p(out.ctypes.data), C.c_void_p(self.ptr + offset), C.c_size_t(nbytes), 0))
        return out

    def write(self, offset, array):
        a = np.ascontiguousarray(array)
        L.check(L.lib().dcdf_device_copy(C.c_void_p(self.ptr + offset), C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 1))

    def free(self):
        if getattr(self, "ptr", None):
            L.lib().dcdf_device_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
