# The README's example, with its claims as assertions (needs an MI355X: python examples/quickstart.py)

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # the repo root: dcdf_amd is used in-tree
import numpy as np, dcdf_amd as dc
from dcdf_amd.raster import EncodedRaster
a = np.random.default_rng(0).integers(0, 1000, size=(32, 256, 256)).astype(np.int32)
built = dc.Chunk.build(a)
data = built.data.write_to()
chunk = dc.Chunk.read_from(data)
assert chunk.get(3, 10, 20) == a[3, 10, 20]
assert (chunk.fill_window(dc.Cube(0, 8, 0, 64, 0, 64)) == a[:8, :64, :64]).all()
hits = chunk.iter_search(dc.Cube(0, 32, 0, 256, 0, 256), 100, 110)
assert len(hits) == int(((a >= 100) & (a <= 110)).sum())
grid = EncodedRaster.chunk_grid(a.shape, tile=128, chunk_size=16)
chunks = [b.data for b in dc.build_batch([np.ascontiguousarray(a[t0:t1, r0:r1, c0:c1]) for t0, t1, r0, r1, c0, c1 in grid])]
R = EncodedRaster(a.shape, chunks, tile=128, chunk_size=16)
flat, offsets, _ = R.fill_windows_flat([(10, 20, 100, 140, 120, 200)], dtype=np.int32)
assert (flat.reshape(10, 40, 80) == a[10:20, 100:140, 120:200]).all()
triples, offs, counts, _ = R.search_flat([(0, 32, 0, 256, 0, 256)], [100], [110])
assert int(counts[0]) == len(hits)
print("readme example ok")
