#!/bin/bash
# A/B of compiler scheduling strategies for the sidelen-256 int32 encoder instantiation
O=gpurun_out/r03m; mkdir -p $O
for rep in 1 2; do
for v in base nofence; do
  L=$PWD/dcdf_amd/libdcdf_k2r_$v.so; [ $v = base ] && L=$PWD/dcdf_amd/libdcdf_k2r.so
  DCDF_K2R_LIB=$L python bench.py --no-gather --cpu-sample 0 --host-sample 0 --verify 4 --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'kernel_ms %.3f' % d['roofline']['kernel_ms'], 'failed', d['config']['failed_tiles_rank0'], 'verified', d['config']['bytes_verified_vs_oracle'])"
done
done
