#!/bin/bash
# end-of-round evidence: the whole GPU suite, the profiles of the final kernel (rocprofv3 stats + PMC passes), the per-wave breakdown
set -e
O=gpurun_out/r04n; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
bash tools/collect_profiles.sh r04 > $O/collect.log 2>&1 || { tail -20 $O/collect.log; exit 1; }
tail -1 $O/collect.log
DCDF_K2R_LIB=$PWD/dcdf_amd/libdcdf_k2r_prof.so K2R_PROFILE_PRINT=1 python bench.py --no-gather --cpu-sample 0 --host-sample 0 --verify 0 --decode-queries 0 --also= --steps 1 --warmup 1 > $O/prof.json 2> $O/prof.err
grep "k2r-pw" $O/prof.err | tail -18
