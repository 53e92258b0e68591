#!/bin/bash
# compiler-flag variants of the headline instantiation, A/B on one box
O=gpurun_out/r03t; mkdir -p $O
B="python bench.py --no-gather --cpu-sample 0 --host-sample 0 --verify 4 --steps 8 --warmup 3"
for rep in 1 2 3; do
for v in base $VARIANTS; do
  if [ $v = base ]; then $B > $O/${v}_$rep.json 2>> $O/err.log; else DCDF_K2R_LIB=dcdf_amd/libdcdf_k2r_$v.so $B > $O/${v}_$rep.json 2>> $O/err.log; fi
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03t/*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: print(f,'unreadable'); continue
    print(f.split('/')[-1], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'verified', d['config']['bytes_verified_vs_oracle'], 'failed', d['config']['failed_tiles_rank0'])
PY
