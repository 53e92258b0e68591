#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel stats of the default bench command, then separate PMC passes
# (FETCH_SIZE, WRITE_SIZE, SQ mix) as MI355X_MICROARCH.md prescribes, and condenses them into profiles-ready files.
# usage: tools/collect_profiles.sh <tag>     outputs under gpurun_out/prof_<tag>/
set -o pipefail
TAG=${1:-rXX}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT/stats $OUT/fetch $OUT/write $OUT/sq $OUT/sq2
B="python3 bench.py --cpu-sample 0 --host-sample 0 --verify 1 --no-gather --decode-queries 0 --also= ${BENCH_ARGS:-}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B > $OUT/write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq -- $B > $OUT/sq.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/sq2 -- $B > $OUT/sq2.log 2>&1 || exit 1
python3 tools/pmc_summary.py $OUT $TAG
