#!/bin/bash
# speculative parts with a shared snapshot copy: the per-GPU shares of configs[3] at N = 8 (384 chunks), N = 4 (768), N = 2 (1536)
# and the whole raster, against whole chunks
O=gpurun_out/r03r; mkdir -p $O
B="python bench.py --cpu-sample 0 --host-sample 0 --no-gather --verify 0 --steps 60 --warmup 20"
run() { # name, env, args
  env $2 $B $3 > $O/$1.json 2>> $O/err.log
}
for rep in 1 2; do
run n8_parts4_$rep K2R_X=0 "--days 192 --extent 2048"
run n8_parts2_$rep K2R_PARTS=2 "--days 192 --extent 2048"
run n8_parts8_$rep K2R_PARTS=8 "--days 192 --extent 2048"
run n8_whole_$rep K2R_SPLIT=0 "--days 192 --extent 2048"
run n4_parts4_$rep K2R_X=0 "--days 384 --extent 2048"
run n4_whole_$rep K2R_SPLIT=0 "--days 384 --extent 2048"
run n2_tail_$rep K2R_SPLIT=tail "--days 365 --extent 2048 --steps 20 --warmup 5"
run n2_whole_$rep K2R_SPLIT=0 "--days 365 --extent 2048 --steps 20 --warmup 5"
done
for rep in 1 2; do
run n1_tail_$rep K2R_SPLIT=tail "--steps 10 --warmup 3"
run n1_whole_$rep K2R_SPLIT=0 "--steps 10 --warmup 3"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03r/*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, 'unreadable'); continue
    print(f.split('/')[-1], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'ms/step %.3f'%d['ms_per_step'], 'cells/s %.3e'%d['value'], 'chunks', d['config']['chunks_on_rank0'], 'failed', d['config']['failed_tiles_rank0'])
PY
tail -5 $O/err.log
