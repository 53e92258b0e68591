#!/bin/bash
# round 4, experiment A: baseline of the round (three bench runs of the shipped build) and the per-WAVE cycle breakdown of the
# diagnostic build (k2r_exec.h pw_*: work vs. barrier wait per wave and phase) on the headline workload.
set -e
O=gpurun_out/r04a; mkdir -p $O
B="python bench.py --no-gather --cpu-sample 0 --host-sample 0 --verify 0 --steps 5 --warmup 2"
for i in 1 2 3; do $B > $O/base_$i.json 2> $O/base_$i.err; done
DCDF_K2R_LIB=$PWD/dcdf_amd/libdcdf_k2r_prof.so K2R_PROFILE_PRINT=1 $B --steps 1 --warmup 1 > $O/prof.json 2> $O/prof.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04a/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'cells/s %.3e'%d['value'], 'frac %.4f'%d['roofline']['frac'])
PY
grep "k2r-p" $O/prof.err | tail -70
