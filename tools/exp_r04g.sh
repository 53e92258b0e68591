#!/bin/bash
# round 4: rehearsal of the multi-rank bench path on one GPU (gloo backend, both ranks on GPU 0) after the host-side changes of the round
O=gpurun_out/r04g; mkdir -p $O
DCDF_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --decode-queries 0 --also= > $O/bench2.json 2> $O/bench2.err || { tail -20 $O/bench2.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04g/bench2.json').read().strip().splitlines()[-1])
print('n_gpus', d['n_gpus'], 'value %.4e'%d['value'], 'ms', d['ms_per_step'], 'sha ok', d['gather'].get('sha_matches_golden'), d['config']['parallelism'])
PY
