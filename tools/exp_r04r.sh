#!/bin/bash
# round 4: the per-GPU shares of configs[3] at N = 8 (384 chunks) and N = 4 (768) on one GPU, speculative parts against whole chunks
# (the splice copy is out of the timed region now: k_stitch<false> checks and accounts, materialize() appends on first use)
O=gpurun_out/r04r; mkdir -p $O
python -m pytest tests/test_gpu_encode.py -m gpu -x -q -k "speculative or parts" > $O/pytest.log 2>&1 || { tail -20 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
B="python bench.py --cpu-sample 0 --host-sample 0 --no-gather --verify 4 --decode-queries 0 --also= --steps 60 --warmup 20"
run() { env $2 $B $3 > $O/$1.json 2>> $O/err.log; }
for rep in 1 2; do
run n8_parts4_$rep K2R_X=0 "--days 192 --extent 2048"
run n8_parts8_$rep K2R_PARTS=8 "--days 192 --extent 2048"
run n8_whole_$rep K2R_SPLIT=0 "--days 192 --extent 2048"
run n4_parts4_$rep K2R_X=0 "--days 384 --extent 2048"
run n4_whole_$rep K2R_SPLIT=0 "--days 384 --extent 2048"
done
run n1_whole_1 K2R_SPLIT=0 "--steps 10 --warmup 3"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04r/*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, 'unreadable'); continue
    print(f.split('/')[-1], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'ms/step %.3f'%d['ms_per_step'], 'chunks', d['config']['chunks_on_rank0'], 'failed', d['config']['failed_tiles_rank0'], 'verified', d['config']['bytes_verified_vs_oracle'])
PY
tail -3 $O/err.log
