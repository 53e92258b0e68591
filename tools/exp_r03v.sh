#!/bin/bash
# A/B of the working tree's library against dcdf_amd/libdcdf_k2r_base.so (the previous build), alternating, one box
O=gpurun_out/r03v; mkdir -p $O; rm -f $O/*.json
B="python bench.py --no-gather --cpu-sample 0 --host-sample 0 --verify 8 --steps 8 --warmup 3"
for rep in 1 2 3; do
$B > $O/new_$rep.json 2>> $O/err.log
DCDF_K2R_LIB=dcdf_amd/libdcdf_k2r_base.so $B > $O/base_$rep.json 2>> $O/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03v/*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: print(f,'unreadable'); continue
    print(f.split('/')[-1], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'verified', d['config']['bytes_verified_vs_oracle'], 'failed', d['config']['failed_tiles_rank0'])
PY
