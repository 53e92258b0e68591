#!/bin/bash
# round 4: decode PMC passes -> profiles/query_traffic_latest.json, then the driver's default bench command (wall time recorded)
set -e
O=gpurun_out/r04j; mkdir -p $O
bash tools/collect_query_profiles.sh r04 > $O/qprof.log 2>&1 || { tail -20 $O/qprof.log; tail -20 gpurun_out/qprof_r04/*.log; exit 1; }
tail -1 $O/qprof.log
cp gpurun_out/qprof_r04/query_traffic_latest.json profiles/query_traffic_latest.json
s=$(date +%s.%N)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
e=$(date +%s.%N)
python3 -c "print(\"bench.py default wall: %.1f s\" % ($e - $s))"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04j/bench_default.json').read().strip().splitlines()[-1])
print('value %.3e frac %.4f kernel_ms %.3f'%(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms']))
print('gather', {k:d['gather'][k] for k in ('sha_matches_golden','device_pack_and_d2h_s')}, 'verified', d['config']['bytes_verified_vs_oracle'])
dec=d['decode']
print('decode', {k: dec[k] for k in ('queries_per_s','cells_per_s','kernel_ms','wall_s','answers_checked_vs_model')}, 'roofline', {k: dec['roofline'][k] for k in ('frac','achieved','traffic')})
print('decode cpu', dec['cpu_baseline']['fill_window'], dec['cpu_baseline']['search_window'])
print('also', d['also'])
print('e2e', d['end_to_end_host_buffers'])
PY
