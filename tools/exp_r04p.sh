#!/bin/bash
# the general path's figures: --dataset wide (three-byte logs) and --dataset noise (every instant a snapshot), + the new test
set -e
O=gpurun_out/r04p; mkdir -p $O
python -m pytest tests/test_gpu_encode.py tests/test_gpu_query.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
B="python bench.py --no-gather --cpu-sample 0 --host-sample 0 --verify 4 --decode-queries 0 --also= --steps 3 --warmup 1"
$B --dataset wide > $O/wide.json 2> $O/wide.err || { tail -5 $O/wide.err; exit 1; }
$B --dataset noise --days 64 > $O/noise64.json 2> $O/noise64.err || { tail -5 $O/noise64.err; exit 1; }
python - <<'PY'
import json
for n in ('wide','noise64'):
    d=json.loads(open('gpurun_out/r04p/%s.json'%n).read().strip().splitlines()[-1])
    print(n, 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'cells/s %.3e'%d['value'], 'frac %.4f'%d['roofline']['frac'], 'failed', d['config']['failed_tiles_rank0'], 'snapshots', d['config']['snapshots_rank0'], 'verified', d['config']['bytes_verified_vs_oracle'])
PY
