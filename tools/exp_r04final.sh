#!/bin/bash
# end of round 4: the whole GPU suite, profiles of the final kernels (encoder + query: rocprofv3 stats and PMC passes), the per-wave
# breakdown, the ingest bench and the driver's default bench command
O=gpurun_out/r04final; mkdir -p $O
bash tools/exp_r04n.sh > $O/r04n.log 2>&1 || { tail -30 $O/r04n.log; exit 1; }
tail -22 $O/r04n.log
cp gpurun_out/prof_r04/traffic_latest.json profiles/traffic_latest.json
bash tools/collect_query_profiles.sh r04 > $O/qcollect.log 2>&1 || { tail -20 $O/qcollect.log; exit 1; }
cp gpurun_out/qprof_r04/query_traffic_latest.json profiles/query_traffic_latest.json
timeout -k 10 600 python tools/bench_ingest.py > $O/ingest.json 2> $O/ingest.err || exit 1
w0=$(date +%s.%N)
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
w1=$(date +%s.%N)
python3 -c "print('default bench wall s', $w1 - $w0)"
cp profiles/traffic_latest.json profiles/query_traffic_latest.json $O/
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04final/bench_default.json').read().strip().splitlines()[-1])
print('value %.4e ms %.3f frac %.4f traffic %s'%(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['traffic']))
print('decode', d['decode']['cells_per_s'], d['decode']['queries_per_s'], d['decode']['roofline']['frac'], d['decode']['roofline']['traffic'])
print('also', {k:(v['kernel_ms'], v['frac']) for k,v in d['also'].items()})
print('host e2e', d['end_to_end_host_buffers']['input_GB_per_s'], 'sha ok', d['gather']['sha_matches_golden'])
i=json.load(open('gpurun_out/r04final/ingest.json'))
print('ingest GB/s', i['input_GB_per_s'], 'over h2d', i['ingest_over_h2d_rate'], 'device slice s', i['device_resident_slice']['seconds'])
PY
