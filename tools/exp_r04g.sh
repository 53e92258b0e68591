#!/bin/bash
# quick check: encode parity tests, then timing of the shipped library and of VARIANTS
set -e
O=gpurun_out/r04g; mkdir -p $O
python -m pytest tests/test_gpu_encode.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
B="python bench.py --no-gather --cpu-sample 0 --host-sample 0 --verify 8 --steps 5 --warmup 2"
for v in ship ${VARIANTS:-}; do
  if [ "$v" = ship ]; then unset DCDF_K2R_LIB; else export DCDF_K2R_LIB=$PWD/dcdf_amd/libdcdf_k2r_$v.so; fi
  for i in 1 2; do $B > $O/time_${v}_$i.json 2> $O/time_${v}_$i.err; done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04g/time_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'frac %.4f'%d['roofline']['frac'], 'failed', d['config']['failed_tiles_rank0'], 'verified', d['config']['bytes_verified_vs_oracle'])
PY
