#!/bin/bash
# does the memory system cost the encoder anything?  1024 chunks: distinct / 1024 copies of chunk 0 / 1024 aliases of chunk 0
O=gpurun_out/r03q; mkdir -p $O
B="python bench.py --workload config1 --cpu-sample 0 --host-sample 0 --no-gather --verify 0 --steps 20 --warmup 5"
for rep in 1 2; do
$B > $O/distinct_$rep.json 2>> $O/err.log
DCDF_BENCH_SAME=copy $B > $O/copy_$rep.json 2>> $O/err.log
DCDF_BENCH_SAME=alias $B > $O/alias_$rep.json 2>> $O/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03q/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'cells/s %.3e'%d['value'], 'failed', d['config']['failed_tiles_rank0'])
PY
