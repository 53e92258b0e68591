"""Generates tests/golden/cpc_precip_day.npz from the data file the reference's own tests hold
(/root/reference/py-dcdf/tests/testdata.txt, used by py-dcdf/tests/test_dcdf.py:357-365: one 360 x 720 float32 day of
CPC precipitation; 166 555 NaN, 54 976 zeros, max 244.9, up to 29 fractional bits).  Data only: 259 200 float32 values.
Run in the build container (the reference tree does not exist on the GPU box)."""
import os

import numpy as np

src = "/root/reference/py-dcdf/tests/testdata.txt"
here = os.path.dirname(os.path.abspath(__file__))
with open(src) as f:
    data = np.array([np.float32(x) for x in f], dtype=np.float32).reshape(360, 720)
np.savez_compressed(os.path.join(here, "cpc_precip_day.npz"), precip=data)
print(data.shape, int(np.isnan(data).sum()), float(np.nanmax(data)))
