#!/bin/bash
# round 4, experiment T: the banded superchunk assembly (upload | sessions | download + hash on three threads)
mkdir -p gpurun_out/r04t
run() {
  tag=$1; shift
  env "$@" K2R_SC_TIMING=1 timeout -k 10 600 python tools/bench_ingest.py --days 96 > gpurun_out/r04t/ingest_$tag.json 2> gpurun_out/r04t/ingest_$tag.err || exit 1
  echo "$tag: $@"; grep "whole call" gpurun_out/r04t/ingest_$tag.err | tail -2
  python -c "import json; d=json.load(open('gpurun_out/r04t/ingest_$tag.json')); print(d['seconds_per_full_slice'], d['ingest_over_h2d_rate'], d['device_resident_slice']['seconds'])"
}
run dflt K2R_X=1
run b256 K2R_SC_BAND_MB=256
run b1024 K2R_SC_BAND_MB=1024
run b0 K2R_SC_BAND_MB=0
tail -36 gpurun_out/r04t/ingest_dflt.err
