#!/bin/bash
set -e
O=gpurun_out/r04s; mkdir -p $O
python -m pytest tests/test_gpu_query.py tests/test_gpu_generic.py tests/test_gpu_configs.py -m gpu -x -q -k "not config2_full and not config1_all" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
python tools/bench_search_k3.py > $O/k3.json 2> $O/k3.err || { tail -20 $O/k3.err; exit 1; }
cat $O/k3.json
