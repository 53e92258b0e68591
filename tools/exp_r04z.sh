#!/bin/bash
# round 4, experiment Z2: the byte form of the compact copy, with and without its wide-node branch (timing only for the latter)
O=gpurun_out/r04z; mkdir -p $O
B="python bench.py --no-gather --cpu-sample 0 --host-sample 0 --verify 0 --decode-queries 0 --also= --steps 5 --warmup 2"
for i in 1 2; do
  timeout -k 10 300 $B > $O/cc_$i.json 2> $O/cc_$i.err || exit 1
  DCDF_K2R_LIB=$PWD/dcdf_amd/libdcdf_k2r_ccnowide.so timeout -k 10 300 $B > $O/ccnw_$i.json 2> $O/ccnw_$i.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04z/cc*_[12].json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'failed', d['config']['failed_tiles_rank0'])
PY
