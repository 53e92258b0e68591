#!/bin/bash
# round 4, experiment Z4: A/B on one box -- the dense remainder arrays compiled in (shipped) and out (-DK2R_NO_DENSE_LISTS): int64, int32, noise
O=gpurun_out/r04z; mkdir -p $O
B="python bench.py --no-gather --cpu-sample 0 --host-sample 0 --verify 0 --decode-queries 0 --also= --steps 5 --warmup 2"
for i in 1 2; do
  for cfg in "i64:--dtype i64" "i32:" "noise:--dataset noise --days 64"; do
    tag=${cfg%%:*}; args=${cfg#*:}
    timeout -k 10 300 $B $args > $O/ab_dense_${tag}_$i.json 2> $O/ab.err || exit 1
    DCDF_K2R_LIB=$PWD/dcdf_amd/libdcdf_k2r_nodense.so timeout -k 10 300 $B $args > $O/ab_nodense_${tag}_$i.json 2> $O/ab.err || exit 1
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04z/ab_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], 'kernel_ms %.3f'%d['roofline']['kernel_ms'])
PY
