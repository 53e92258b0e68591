#!/bin/bash
# 384 chunks on 256 CUs (the per-GPU share of configs[3] at N = 8): speculative halves vs whole chunks, long runs (clocks settled)
O=gpurun_out/r03g; mkdir -p $O
B="python bench.py --cpu-sample 0 --host-sample 0 --no-gather --verify 0 --days 192 --extent 2048 --steps 100 --warmup 30"
for rep in 1 2; do
$B > $O/split_$rep.json 2>> $O/err.log
K2R_SPLIT=0 $B > $O/nosplit_$rep.json 2>> $O/err.log
done
python bench.py --cpu-sample 0 --host-sample 0 --no-gather --verify 0 --days 32 --steps 100 --warmup 30 > $O/one_round.json 2>> $O/err.log
K2R_SPLIT=all python bench.py --cpu-sample 0 --host-sample 0 --no-gather --verify 0 > $O/splitall.json 2>> $O/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03g/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'cells/s %.3e'%d['value'], 'failed', d['config']['failed_tiles_rank0'])
PY
