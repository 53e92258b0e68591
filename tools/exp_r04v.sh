#!/bin/bash
# round 4, experiment V: per-phase / per-wave cycles of the general (re-reading) emission path.
# V1 (profiles/r04_encoder_experiments.json): --dataset wide and --dataset noise.  V2: a model Snapshot instant (--days 1).  V3/V4: noise
# again with the dense remainder arrays (the steps of dac_finish stamped separately), then the shipped build's timings.
O=gpurun_out/r04v; mkdir -p $O
B="python bench.py --no-gather --cpu-sample 0 --host-sample 0 --decode-queries 0 --also= --steps 3 --warmup 1"
DCDF_K2R_LIB=$PWD/dcdf_amd/libdcdf_k2r_prof.so K2R_PROFILE_PRINT=1 timeout -k 10 300 $B --verify 0 --steps 1 --dataset noise --days 64 > $O/noise3.json 2> $O/noise3.err || exit 1
grep "k2r-pw" $O/noise3.err | tail -16
timeout -k 10 300 $B --dataset noise --days 64 > $O/noise.json 2> $O/noise.err || exit 1
timeout -k 10 300 $B --dataset wide > $O/wide.json 2> $O/wide.err || exit 1
timeout -k 10 300 $B --steps 5 --warmup 2 > $O/model.json 2> $O/model.err || exit 1
python - <<'PY'
import json
for f in ("noise","wide","model"):
    d=json.loads(open('gpurun_out/r04v/%s.json'%f).read().strip().splitlines()[-1])
    print(f, 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'frac %.4f'%d['roofline']['frac'], 'failed', d['config']['failed_tiles_rank0'], 'verified', d['config']['bytes_verified_vs_oracle'])
PY
