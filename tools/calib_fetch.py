"""Calibrates rocprofv3 FETCH_SIZE on the encoder's own read pattern (run under rocprofv3 --pmc FETCH_SIZE)."""
import ctypes as C
import sys
sys.path.insert(0, ".")
import torch
from dcdf_amd import _lib as L
n, T = 1024, 32
x = torch.randint(-1000, 1000, (n, T, 256, 256), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
L.check(L.lib().dcdf_calib_read(C.c_void_p(x.data_ptr()), C.c_uint32(n), C.c_uint32(T)))
print("known_bytes", n * T * 256 * 256 * 4)
