#!/bin/bash
# round 4, experiment W: the list-mode Dac planes from dense remainder arrays (level-order stores) in the integer kernels: the whole GPU
# suite, then noise / wide / model
O=gpurun_out/r04w; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
rc=$?; tail -5 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
B="python bench.py --no-gather --cpu-sample 0 --host-sample 0 --decode-queries 0 --also= --steps 3 --warmup 1"
timeout -k 10 300 $B --dataset noise --days 64 > $O/noise.json 2> $O/noise.err || exit 1
timeout -k 10 300 $B --dataset wide > $O/wide.json 2> $O/wide.err || exit 1
timeout -k 10 300 $B --steps 5 --warmup 2 > $O/model.json 2> $O/model.err || exit 1
timeout -k 10 300 $B --steps 5 --warmup 2 --dtype i64 > $O/model64.json 2> $O/model64.err || exit 1
python - <<'PY'
import json
for f in ("noise","wide","model","model64"):
    d=json.loads(open('gpurun_out/r04w/%s.json'%f).read().strip().splitlines()[-1])
    print(f, 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'cells/s %.3e'%d['value'], 'frac %.4f'%d['roofline']['frac'], 'failed', d['config']['failed_tiles_rank0'], 'verified', d['config']['bytes_verified_vs_oracle'])
PY
