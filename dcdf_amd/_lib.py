"""ctypes binding of libdcdf_k2r.so (the HIP/gfx950 product library).  There is no fallback: if the
library is missing or no GPU is present every operation raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DCDF_K2R_LIB", os.path.join(_HERE, "libdcdf_k2r.so"))  # override: diagnostic builds

DCDF_I32, DCDF_I64, DCDF_F32, DCDF_F64 = 4, 8, 32, 64
MEM_HOST, MEM_DEVICE = 0, 1


class TileDesc(C.Structure):
    _fields_ = [("base", C.c_void_p), ("dtype", C.c_int32), ("_pad0", C.c_int32), ("stride_t", C.c_int64),
                ("stride_r", C.c_int64), ("stride_c", C.c_int64), ("instants", C.c_uint32), ("rows", C.c_uint32),
                ("cols", C.c_uint32), ("fractional_bits", C.c_uint8), ("round", C.c_uint8), ("_pad1", C.c_uint8 * 2)]


class Encoded(C.Structure):
    _fields_ = [("bytes", C.POINTER(C.c_uint8)), ("len", C.c_size_t), ("snapshots", C.c_uint32), ("logs", C.c_uint32),
                ("status", C.c_int32), ("_pad", C.c_int32), ("minmax", C.POINTER(C.c_int64))]


class StoredObject(C.Structure):
    _fields_ = [("cid", C.c_uint8 * 36), ("bytes", C.POINTER(C.c_uint8)), ("len", C.c_size_t)]


class SuperchunkBuild(C.Structure):
    _fields_ = [("objects", C.POINTER(StoredObject)), ("n_objects", C.c_size_t), ("size", C.c_uint64), ("elided", C.c_uint32),
                ("local", C.c_uint32), ("external", C.c_uint32), ("snapshots", C.c_uint32), ("logs", C.c_uint32)]


class Cube(C.Structure):
    _fields_ = [("start", C.c_uint32), ("end", C.c_uint32), ("top", C.c_uint32), ("bottom", C.c_uint32),
                ("left", C.c_uint32), ("right", C.c_uint32)]


class DcdfError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        msg = what
        try:
            msg = "%s: %s" % (what, lib().dcdf_strerror(code).decode())
        except Exception:
            pass
        super().__init__("dcdf_k2r error %d %s" % (code, msg))


_lib = None

# every symbol include/dcdf_k2r.h declares
SYMBOLS = [
    "dcdf_chunk_build_batch", "dcdf_chunk_build", "dcdf_free_encoded", "dcdf_encoder_create", "dcdf_encoder_run",
    "dcdf_encoder_result", "dcdf_encoder_fetch", "dcdf_encoder_gather_size", "dcdf_encoder_gather", "dcdf_encoder_total_bytes", "dcdf_encoder_destroy", "dcdf_superchunk_build", "dcdf_free_superchunk", "dcdf_chunk_open",
    "dcdf_chunk_close", "dcdf_chunk_info", "dcdf_chunk_get", "dcdf_chunk_fill_cell", "dcdf_chunk_fill_window",
    "dcdf_chunk_search", "dcdf_query_fill_window_batch", "dcdf_query_search_batch", "dcdf_query_fill_window_batch_typed", "dcdf_query_search_batch_mem", "dcdf_query_get_batch", "dcdf_query_fill_cell_batch", "dcdf_chunk_open_batch", "dcdf_chunk_instant_layout", "dcdf_raster_create", "dcdf_raster_destroy", "dcdf_raster_fill_window_batch", "dcdf_raster_search_batch", "dcdf_suggest_fraction", "dcdf_encoder_object_sha256",
    "dcdf_synth_fill", "dcdf_calib_read", "dcdf_device_alloc", "dcdf_device_free", "dcdf_device_copy", "dcdf_strerror", "dcdf_device_name", "dcdf_abi_version", "dcdf_last_hip_error", "dcdf_device_pool_trim",
]


def lib():
    """Loads the HIP library; raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libdcdf_k2r.so not built (run __graft_entry__.build() / make -C dcdf_amd/csrc); "
                               "the MI355X path has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.dcdf_strerror.restype = C.c_char_p
        L.dcdf_device_name.restype = C.c_char_p
        L.dcdf_encoder_total_bytes.restype = C.c_uint64
        L.dcdf_encoder_total_bytes.argtypes = [C.c_void_p]
        L.dcdf_free_encoded.argtypes = [C.c_void_p, C.c_size_t]
        L.dcdf_free_encoded.restype = None
        L.dcdf_encoder_destroy.argtypes = [C.c_void_p]
        L.dcdf_encoder_destroy.restype = None
        L.dcdf_chunk_close.argtypes = [C.c_void_p]
        L.dcdf_chunk_close.restype = None
        L.dcdf_free_superchunk.argtypes = [C.c_void_p]
        L.dcdf_free_superchunk.restype = None
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise DcdfError(rc, what)
