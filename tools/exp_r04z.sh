#!/bin/bash
# round 4, experiment Z5: the float32 kernel with the out-of-line exact conversion compiled out of it (timing only; exact inputs never take it)
O=gpurun_out/r04z; mkdir -p $O
B="python bench.py --no-gather --cpu-sample 0 --host-sample 0 --verify 0 --decode-queries 0 --also= --steps 5 --warmup 2 --dtype f32"
for i in 1 2; do
  timeout -k 10 300 $B > $O/f32_$i.json 2> $O/f32.err || exit 1
  DCDF_K2R_LIB=$PWD/dcdf_amd/libdcdf_k2r_fnoslow.so timeout -k 10 300 $B > $O/f32ns_$i.json 2> $O/f32.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04z/f32*_[12].json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'failed', d['config']['failed_tiles_rank0'])
PY
