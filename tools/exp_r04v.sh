#!/bin/bash
# round 4, experiment V: per-phase / per-wave cycles of the general (re-reading) emission path: --dataset wide and --dataset noise
O=gpurun_out/r04v; mkdir -p $O
B="python bench.py --no-gather --cpu-sample 0 --host-sample 0 --verify 0 --decode-queries 0 --also= --steps 1 --warmup 1"
DCDF_K2R_LIB=$PWD/dcdf_amd/libdcdf_k2r_prof.so K2R_PROFILE_PRINT=1 timeout -k 10 300 $B --dataset wide > $O/wide.json 2> $O/wide.err || exit 1
DCDF_K2R_LIB=$PWD/dcdf_amd/libdcdf_k2r_prof.so K2R_PROFILE_PRINT=1 timeout -k 10 300 $B --dataset noise --days 64 > $O/noise.json 2> $O/noise.err || exit 1
echo WIDE; grep "k2r-p" $O/wide.err | tail -40
echo NOISE; grep "k2r-p" $O/noise.err | tail -40
