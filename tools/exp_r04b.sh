#!/bin/bash
# round 4, experiment B: lean phase 1 -- parity (encode tests + the full-raster parity test) and three timing runs
set -e
O=gpurun_out/r04b; mkdir -p $O
python -m pytest tests/test_gpu_encode.py tests/test_gpu_configs.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
B="python bench.py --no-gather --cpu-sample 0 --host-sample 0 --verify 4 --steps 5 --warmup 2"
for i in 1 2 3; do $B > $O/lean_$i.json 2> $O/lean_$i.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04b/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'cells/s %.3e'%d['value'], 'frac %.4f'%d['roofline']['frac'], 'failed', d['config']['failed_tiles_rank0'])
PY
