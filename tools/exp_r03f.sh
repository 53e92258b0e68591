#!/bin/bash
# GPU suite + default bench + forced-split bench (SHA of the gather must equal) + noise dataset
O=gpurun_out/r03f; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1800 $O/bench.json
K2R_SPLIT=all python bench.py --cpu-sample 0 --host-sample 0 > $O/bench_splitall.json 2>> $O/bench.err
K2R_MAX_WGS=256 python bench.py --cpu-sample 0 --host-sample 0 --no-gather --days 192 --extent 2048 > $O/bench_384.json 2>> $O/bench.err
K2R_SPLIT=0 python bench.py --cpu-sample 0 --host-sample 0 --no-gather --days 192 --extent 2048 > $O/bench_384_nosplit.json 2>> $O/bench.err
python bench.py --dataset noise --days 64 --cpu-sample 0 --host-sample 0 > $O/bench_noise.json 2>> $O/bench.err
python - <<'PY'
import json
for f in ['bench','bench_splitall','bench_384','bench_384_nosplit','bench_noise']:
    try:
        d=json.loads(open('gpurun_out/r03f/%s.json'%f).read().strip().splitlines()[-1])
        print(f, 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'cells/s %.3e'%d['value'], 'frac %.4f'%d['roofline']['frac'], 'failed', d['config']['failed_tiles_rank0'], 'snaps', d['config']['snapshots_rank0'], (d.get('gather') or {}).get('sha256_of_concatenation_in_chunk_order','')[:16])
    except Exception as e: print(f,'ERR',e)
PY
