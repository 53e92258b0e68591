set -e
O=gpurun_out/r03c; mkdir -p $O
for d in 1 2 3 5 9 32 33; do
python bench.py --no-gather --cpu-sample 0 --verify 0 --steps 5 --warmup 2 --days $d > $O/days_$d.json 2>> $O/err.log
done
python - <<'PY'
import json,glob
for d in [1,2,3,5,9,32,33]:
    x=json.loads(open('gpurun_out/r03c/days_%d.json'%d).read().strip().splitlines()[-1])
    print(d, 'kernel_ms %.4f'%x['roofline']['kernel_ms'])
PY
