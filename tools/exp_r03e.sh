#!/bin/bash
# GPU suite + the default bench line + the noise dataset line
O=gpurun_out/r03e; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 2500 $O/bench.json
python bench.py --dataset noise --days 64 --cpu-sample 0 --host-sample 0 > $O/bench_noise.json 2>> $O/bench.err; head -c 700 $O/bench_noise.json
