#!/bin/bash
# round 4, experiment V: per-phase / per-wave cycles of the general (re-reading) emission path.
# V1 (profiles/r04_encoder_experiments.json): --dataset wide and --dataset noise.  V2: a model Snapshot instant (--days 1).
O=gpurun_out/r04v; mkdir -p $O
B="python bench.py --no-gather --cpu-sample 0 --host-sample 0 --verify 0 --decode-queries 0 --also= --steps 1 --warmup 1"
DCDF_K2R_LIB=$PWD/dcdf_amd/libdcdf_k2r_prof.so K2R_PROFILE_PRINT=1 timeout -k 10 300 $B --days 1 > $O/snap.json 2> $O/snap.err || exit 1
grep "k2r-pw" $O/snap.err | tail -20
