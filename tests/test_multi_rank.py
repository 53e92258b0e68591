"""N>1 path on CPU: world_size-2 gloo run of the sharding + host-side gather (no GPU here, so each rank
"encodes" its shard with the oracle; on the GPU box bench.py does the same sharding with the HIP encoder)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, hashlib
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch.distributed as dist
import oracle_lib as O
from dcdf_amd import synth, shard
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
shapes = [(6, 32, 32)] * 5 + [(3, 32, 32)] * 3 + [(6, 16, 16)] * 4
cells = [t * r * c for t, r, c in shapes]
owner = shard.partition(cells, dist.get_world_size())
mine = shard.my_chunks(owner, dist.get_rank())
local = {}
for i in mine:
    t, r, c = shapes[i]
    local[i] = O.chunk_build(synth.cells(100 + i, 0, t, 0, r, 0, c, np.int32))
full = shard.gather_encoded(local, len(shapes), dist)
if dist.get_rank() == 0:
    h = hashlib.sha256(b"".join(full)).hexdigest()
    loads = [sum(cells[i] for i in shard.my_chunks(owner, k)) for k in range(dist.get_world_size())]
    print("RESULT", h, max(loads) - min(loads), len(full))
dist.barrier()
dist.destroy_process_group()
'''


def test_partition_is_balanced_and_deterministic():
    from dcdf_amd import shard
    cells = [32] * 11 * 256 + [13] * 256  # 11 full + 1 short time segment per tile
    for w in (1, 2, 4, 8):
        owner = shard.partition(cells, w)
        loads = [sum(c for c, o in zip(cells, owner) if o == r) for r in range(w)]
        assert max(loads) - min(loads) <= 32
        assert owner == shard.partition(cells, w)
        assert sorted(sum((shard.my_chunks(owner, r) for r in range(w)), [])) == list(range(len(cells)))


def test_two_rank_gloo_gather_matches_single_process():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hashlib
    import oracle_lib as O
    from dcdf_amd import synth
    shapes = [(6, 32, 32)] * 5 + [(3, 32, 32)] * 3 + [(6, 16, 16)] * 4
    want = hashlib.sha256(b"".join(O.chunk_build(synth.cells(100 + i, 0, t, 0, r, 0, c, np.int32))
                                   for i, (t, r, c) in enumerate(shapes))).hexdigest()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER, ROOT], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    line = [l for l in outs[0][0].splitlines() if l.startswith("RESULT")][0].split()
    assert line[1] == want and int(line[3]) == len(shapes)
