import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dcdf_amd as dc
import oracle_lib as O
from dcdf_amd import synth
a = synth.cells(0xDCDF0002, 0, 32, 0, 256, 0, 256, np.int32)
f = (a / 8.0).astype(np.float32)
f[5, 100, 7] = np.nan
g = f[:6].copy()
g[2, 9, 9] = np.float32(5.0 + 1 / 64.0)
g[2, 9, 9] = np.float32(3e8)
for rnd in (True,):
    for rep in range(2):
        r = dc.build_batch([g], fractional_bits=3, round=rnd)[0]
        ref = O.chunk_build(g, fractional_bits=3, round_=rnd)
        data = r.data.write_to()
        nd = sum(1 for i in range(min(len(data), len(ref))) if data[i] != ref[i])
        first = next((i for i in range(min(len(data), len(ref))) if data[i] != ref[i]), -1)
        print("round", rnd, "rep", rep, "len", len(data), len(ref), "differing bytes", nd, "first", first, data[:12].hex(), ref[:12].hex(), r.snapshots, r.logs)
        if first >= 0:
            print(" around first:", data[max(0,first-4):first+12].hex(), ref[max(0,first-4):first+12].hex())
