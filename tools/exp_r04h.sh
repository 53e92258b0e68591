#!/bin/bash
# round 4, experiment H: what each phase of the stash-log path costs on the critical path (phase-skipping timing builds, K2R_DIAG_SKIP)
O=gpurun_out/r04h; mkdir -p $O
B="python bench.py --no-gather --cpu-sample 0 --host-sample 0 --verify 0 --steps 5 --warmup 2"
for v in ship ${VARIANTS:-}; do
  if [ "$v" = ship ]; then unset DCDF_K2R_LIB; else export DCDF_K2R_LIB=$PWD/dcdf_amd/libdcdf_k2r_$v.so; fi
  $B > $O/time_${v}.json 2> $O/time_${v}.err || { tail -5 $O/time_${v}.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04h/time_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], 'kernel_ms %.3f'%d['roofline']['kernel_ms'])
PY
