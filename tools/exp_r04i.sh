#!/bin/bash
# round 4, experiment I: what does a Snapshot instant cost next to a Log instant?  (config1 chunks of 1, 2, 3, 5 instants)
O=gpurun_out/r04i; mkdir -p $O
for t in 1 2 3 5 9; do
python bench.py --no-gather --cpu-sample 0 --host-sample 0 --verify 0 --steps 5 --warmup 2 --workload config1 --chunks 4096 --instants $t ${EXTRA_ARGS:-} > $O/t$t.json 2> $O/t$t.err || { tail -5 $O/t$t.err; exit 1; }
done
python - <<'PY'
import json,glob
for t in (1,2,3,5,9):
    d=json.loads(open('gpurun_out/r04i/t%d.json'%t).read().strip().splitlines()[-1])
    print('instants', t, 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'snapshots', d['config'].get('snapshots_rank0'), 'per chunk-instant and CU us %.2f'%(d['roofline']['kernel_ms']*1000*256/(4096*t)))
PY
