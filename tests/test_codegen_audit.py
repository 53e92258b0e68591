"""The built encoder kernels keep their state in registers / LDS: no scratch-resident execution context and no LDS access
through FLAT instructions (tools/check_codegen.py; DESIGN.md section 8 explains the fault this guards against)."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_encoder_objects_pass_the_codegen_audit():
    if not glob.glob(os.path.join(ROOT, "dcdf_amd", "csrc", "_build", "enc_L*.o")):
        pytest.skip("encoder objects not built here (the library was shipped prebuilt)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_codegen.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
