#!/bin/bash
# the bench lines DESIGN.md 7.0 quotes
O=gpurun_out/r03n; mkdir -p $O
python bench.py > $O/config2_i32.json 2> $O/err.log
for dt in i64 f32 f64; do python bench.py --dtype $dt --cpu-sample 0 --host-sample 0 > $O/config2_$dt.json 2>> $O/err.log; done
python bench.py --workload config1 --cpu-sample 0 --host-sample 0 > $O/config1_i32.json 2>> $O/err.log
python bench.py --dataset noise --days 64 --cpu-sample 0 --host-sample 0 > $O/config2_noise64.json 2>> $O/err.log
DCDF_BENCH_BACKEND=gloo python bench.py --gpus 2 --cpu-sample 0 --host-sample 0 > $O/config2_2ranks_gloo.json 2>> $O/err.log
# four ranks on the one card: 768 chunks per rank, i.e. the speculative-parts path of a short queue, gathered over gloo
DCDF_BENCH_BACKEND=gloo python bench.py --gpus 4 --cpu-sample 0 --host-sample 0 --steps 3 --warmup 1 > $O/config2_4ranks_gloo.json 2>> $O/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03n/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'cells/s %.3e'%d['value'], 'frac %.4f'%d['roofline']['frac'], 'failed', d['config']['failed_tiles_rank0'], (d.get('gather') or {}).get('sha256_of_concatenation_in_chunk_order','')[:12], (d['roofline'].get('measured_copy') or {}).get('GB/s'))
    except Exception as e: print(f,'ERR',e)
PY
