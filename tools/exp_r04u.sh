#!/bin/bash
# round 4, experiment U: full GPU suite + ingest + default bench after the pooled streams / banded superchunk assembly
mkdir -p gpurun_out/r04u
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04u/pytest.log 2>&1
rc=$?; tail -5 gpurun_out/r04u/pytest.log; [ $rc -eq 0 ] || exit $rc
K2R_SC_TIMING=1 timeout -k 10 600 python tools/bench_ingest.py > gpurun_out/r04u/ingest.json 2> gpurun_out/r04u/ingest.err || exit 1
grep "whole call" gpurun_out/r04u/ingest.err | tail -3
cat gpurun_out/r04u/ingest.json
timeout -k 10 900 python bench.py > gpurun_out/r04u/bench.json 2> gpurun_out/r04u/bench.err || exit 1
cat gpurun_out/r04u/bench.json
