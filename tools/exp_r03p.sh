#!/bin/bash
# 384 chunks on 256 CUs: whole chunks / halves of the last 128 / halves of all, and the per-tile distribution
O=gpurun_out/r03p; mkdir -p $O
B="python bench.py --cpu-sample 0 --host-sample 0 --no-gather --verify 0 --days 192 --extent 2048 --steps 100 --warmup 30"
for rep in 1 2; do
$B > $O/split_$rep.json 2>> $O/err.log
K2R_SPLIT=0 $B > $O/nosplit_$rep.json 2>> $O/err.log
K2R_SPLIT=all $B > $O/splitall_$rep.json 2>> $O/err.log
done
K2R_PROFILE_PRINT=1 python bench.py --cpu-sample 0 --host-sample 0 --no-gather --verify 0 --days 192 --extent 2048 --steps 3 --warmup 1 > $O/prof.json 2> $O/prof.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03p/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'ms/step %.3f'%d['ms_per_step'], 'cells/s %.3e'%d['value'], 'failed', d['config']['failed_tiles_rank0'])
PY
tail -40 $O/prof.err
