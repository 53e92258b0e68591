#!/bin/bash
# round 4, experiment Z: A/B on one box -- tree top by DPP reductions per wave (shipped) against the five LDS phases (-DK2R_NO_TOPDPP)
O=gpurun_out/r04z; mkdir -p $O
B="python bench.py --no-gather --cpu-sample 0 --host-sample 0 --verify 0 --decode-queries 0 --also= --steps 5 --warmup 2"
for i in 1 2 3; do
  timeout -k 10 300 $B > $O/dpp_$i.json 2> $O/dpp_$i.err || exit 1
  DCDF_K2R_LIB=$PWD/dcdf_amd/libdcdf_k2r_notopdpp.so timeout -k 10 300 $B > $O/lds_$i.json 2> $O/lds_$i.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04z/*_[123].json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], 'kernel_ms %.3f'%d['roofline']['kernel_ms'])
PY
