#!/bin/bash
# round 4, experiment C: per-wave cycle breakdown of the current build (diagnostic library) + one timing run of the shipped one
set -e
O=gpurun_out/r04c; mkdir -p $O
B="python bench.py --no-gather --cpu-sample 0 --host-sample 0 --verify 0 --steps 5 --warmup 2"
$B > $O/ship.json 2> $O/ship.err
DCDF_K2R_LIB=$PWD/dcdf_amd/libdcdf_k2r_prof.so K2R_PROFILE_PRINT=1 $B --steps 1 --warmup 1 > $O/prof.json 2> $O/prof.err
for v in ${VARIANTS:-}; do DCDF_K2R_LIB=$PWD/dcdf_amd/libdcdf_k2r_$v.so $B > $O/var_$v.json 2> $O/var_$v.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04c/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'cells/s %.3e'%d['value'], 'frac %.4f'%d['roofline']['frac'])
PY
grep "k2r-pw" $O/prof.err | tail -18
