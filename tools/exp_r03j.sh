#!/bin/bash
O=gpurun_out/r03j; mkdir -p $O
for v in base stag2 stag5; do
  L=$PWD/dcdf_amd/libdcdf_k2r_$v.so; [ $v = base ] && L=$PWD/dcdf_amd/libdcdf_k2r.so
  for d in 32 192 365; do
    ext=4096; [ $d = 192 ] && ext=2048
    DCDF_K2R_LIB=$L python bench.py --no-gather --cpu-sample 0 --host-sample 0 --verify 0 --steps 30 --warmup 10 --days $d --extent $ext 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'days $d', d['config']['chunks_total'], 'kernel_ms %.3f' % d['roofline']['kernel_ms'])"
  done
done
