#!/bin/bash
# tools/gpu_check.sh TAG [pytest args]: the GPU parity suite, then the headline bench (kernel time only), into gpurun_out/TAG/
set -e
T=${1:-chk}; shift || true
O=gpurun_out/$T; mkdir -p $O
python -m pytest tests -m gpu -x -q "$@" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for i in 1 2 3; do
python bench.py --no-gather --cpu-sample 0 --verify 4 --steps 5 --warmup 2 > $O/bench_$i.json 2> $O/bench_$i.err || { tail -20 $O/bench_$i.err; exit 1; }
done
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'cells/s %.3e'%d['value'], 'frac %.4f'%d['roofline']['frac'], 'failed', d['config']['failed_tiles_rank0'])
PY
