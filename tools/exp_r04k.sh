#!/bin/bash
# round 4: the caller path -- superchunk / dataset parity tests, the step breakdown of one slice (K2R_SC_TIMING), the ingest bench
set -e
O=gpurun_out/r04k; mkdir -p $O
python -m pytest tests/test_gpu_superchunk.py tests/test_gpu_dataset.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
K2R_SC_TIMING=1 python tools/bench_superchunk.py > $O/superchunk.json 2> $O/superchunk_steps.txt || { tail -20 $O/superchunk_steps.txt; exit 1; }
cat $O/superchunk.json; grep k2r-sc $O/superchunk_steps.txt | tail -14
python tools/bench_ingest.py > $O/ingest.json 2> $O/ingest.err || { tail -20 $O/ingest.err; exit 1; }
cat $O/ingest.json
