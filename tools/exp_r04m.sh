#!/bin/bash
# float paths after moving the exact conversion out of line: parity (encode tests) + timing of f32 / f64 / i32
set -e
O=gpurun_out/r04m; mkdir -p $O
python -m pytest tests/test_gpu_encode.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for dt in f32 f64 i32; do
python bench.py --no-gather --cpu-sample 0 --host-sample 0 --verify 4 --decode-queries 0 --also= --steps 5 --warmup 2 --dtype $dt > $O/$dt.json 2> $O/$dt.err || { tail -5 $O/$dt.err; exit 1; }
done
python - <<'PY'
import json
for dt in ('f32','f64','i32'):
    d=json.loads(open('gpurun_out/r04m/%s.json'%dt).read().strip().splitlines()[-1])
    print(dt, 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'frac %.4f'%d['roofline']['frac'], 'failed', d['config']['failed_tiles_rank0'], 'verified', d['config']['bytes_verified_vs_oracle'])
PY
