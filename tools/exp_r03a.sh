#!/bin/bash
# round 3, experiment A: (1) does kernel time follow VALU count or total instruction count?  (2) what is co-residency worth?
set -e
O=gpurun_out/r03a; mkdir -p $O
B="python bench.py --no-gather --cpu-sample 0 --verify 0 --steps 5 --warmup 2"
$B > $O/base_config2.json 2> $O/base_config2.err
for rep in 1 2 3; do
  for t in pad0 pad1 pad2; do
    DCDF_K2R_LIB=$PWD/dcdf_amd/libdcdf_k2r_$t.so $B > $O/${t}_$rep.json 2>> $O/pad.err
    echo "$t $rep done"
  done
done
$B --workload config1 --side 256 --chunks 1024 > $O/c1_side256.json 2>> $O/c1.err
$B --workload config1 --side 128 --chunks 4096 > $O/c1_side128.json 2>> $O/c1.err
$B --workload config1 --side 64 --chunks 16384 > $O/c1_side64.json 2>> $O/c1.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03a/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'cells/s %.3e'%d['value'], 'frac %.4f'%d['roofline']['frac'])
    except Exception as e:
        print(f, 'ERR', e)
PY
