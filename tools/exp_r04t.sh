#!/bin/bash
# round 4, experiment T: the banded superchunk assembly (upload | sessions | download + hash on three threads)
mkdir -p gpurun_out/r04t
timeout -k 10 600 python -m pytest tests/test_gpu_superchunk.py tests/test_gpu_dataset.py -m gpu -x -q > gpurun_out/r04t/pytest.log 2>&1
rc=$?; tail -5 gpurun_out/r04t/pytest.log; [ $rc -eq 0 ] || exit $rc
K2R_SC_TIMING=1 timeout -k 10 600 python tools/bench_ingest.py > gpurun_out/r04t/ingest.json 2> gpurun_out/r04t/ingest.err || exit 1
grep "whole call" gpurun_out/r04t/ingest.err | tail -3
grep trace gpurun_out/r04t/ingest.err | tail -40 | head -38
python -c "import json; d=json.load(open('gpurun_out/r04t/ingest.json')); print(d['seconds_per_full_slice'], d['input_GB_per_s'], d['ingest_over_h2d_rate'], d['device_resident_slice']['seconds'])"
