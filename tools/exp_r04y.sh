#!/bin/bash
# round 4, experiment Y: device-side structural validation at dcdf_chunk_open_batch: query tests + the cost of opening 3072 chunks
O=gpurun_out/r04y; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_query.py tests/test_gpu_configs.py -m gpu -x -q -k "not config2_full and not config1_all" > $O/pytest.log 2>&1
rc=$?; tail -5 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
K2R_OPEN_TIMING=1 timeout -k 10 600 python tools/bench_query.py --queries 20000 > $O/query.json 2> $O/query.err || exit 1
grep "k2r-open" $O/query.err | tail -12
python -c "import json; d=json.loads(open('gpurun_out/r04y/query.json').read().strip().splitlines()[-1]); print({k: d[k] for k in d if 'open' in k or k in ('queries_per_s','cells_per_s')})"
