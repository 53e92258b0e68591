"""CPU-side checks of the product library: it loads, exports every symbol include/dcdf_k2r.h declares, and
refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(ROOT, "dcdf_amd", "libdcdf_k2r.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-j", "8", "-C", os.path.join(ROOT, "dcdf_amd", "csrc")])
    from dcdf_amd import _lib
    return _lib.lib()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "dcdf_k2r.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(dcdf_[a-z_0-9]+)\s*\(", hdr))
    from dcdf_amd import _lib
    assert declared == set(_lib.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.dcdf_abi_version() == 3
    assert lib.dcdf_strerror(-8).decode().startswith("unsupported")


def test_no_oracle_or_simulator_in_product(lib):
    so = os.path.join(ROOT, "dcdf_amd", "libdcdf_k2r.so")
    syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
    assert "orc_" not in syms and "sim_encode" not in syms
    ldd = subprocess.run(["ldd", so], capture_output=True, text=True).stdout
    assert "k2r_oracle" not in ldd and "k2r_sim" not in ldd
    assert "amdhip64" in ldd


def test_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import dcdf_amd
    a = np.zeros((2, 8, 8), dtype=np.int32)
    with pytest.raises(dcdf_amd.DcdfError) as e:
        dcdf_amd.Chunk.build(a)
    assert e.value.code == -9  # DCDF_ERR_NO_DEVICE
    with pytest.raises(dcdf_amd.DcdfError):
        dcdf_amd.Chunk(b"\x08\x00\x00\x00\x00\x00")
