"""ctypes loader for the host-side simulator of the kernel bodies (tests/_build/libk2r_sim.so).
TEST INFRASTRUCTURE ONLY: the product path never loads this."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_libs = {}
ENC = {np.dtype("int32"): 4, np.dtype("int64"): 8, np.dtype("float32"): 32, np.dtype("float64"): 64}


def lib():
    name = "libk2r_sim_asan.so" if os.environ.get("K2R_SIM_ASAN") else "libk2r_sim.so"
    if name not in _libs:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
        _libs[name] = C.CDLL(os.path.join(ROOT, "tests", "_build", name))
    return _libs[name]


def encode(a, fractional_bits=0, round_=False, cap=None, force_novec=False, want_minmax=False):
    a = np.asarray(a)
    assert a.ndim == 3
    st = [s // a.itemsize for s in a.strides]
    if cap is None:
        cap = a.size * 12 + 65536
    out = np.zeros(cap, dtype=np.uint8)
    mm = np.zeros((a.shape[0], 2), dtype=np.int64)
    status, ns, nl, ln = C.c_int32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
    rc = lib().sim_encode(C.c_void_p(a.ctypes.data), ENC[a.dtype], C.c_int64(st[0]), C.c_int64(st[1]),
                          C.c_int64(st[2]), C.c_uint32(a.shape[0]), C.c_uint32(a.shape[1]), C.c_uint32(a.shape[2]),
                          int(fractional_bits), int(bool(round_)), C.c_void_p(out.ctypes.data), C.c_uint64(cap),
                          C.c_void_p(mm.ctypes.data), int(force_novec), C.byref(status), C.byref(ns), C.byref(nl),
                          C.byref(ln))
    if rc != 0:
        return rc, None, 0, 0, None
    data = bytes(out[:ln.value]) if status.value == 0 else None
    if want_minmax:
        return status.value, data, ns.value, nl.value, mm
    return status.value, data, ns.value, nl.value
