#!/bin/bash
# round 4, experiment F: timing and FETCH_SIZE / WRITE_SIZE of library variants (VARIANTS="ship nt ..."; "ship" = the shipped library)
set -o pipefail
O=gpurun_out/r04f; mkdir -p $O
export TMPDIR=/tmp
B="python3 bench.py --no-gather --cpu-sample 0 --host-sample 0 --verify 0 --steps 5 --warmup 2"
for v in ${VARIANTS:-ship}; do
  if [ "$v" = ship ]; then unset DCDF_K2R_LIB; else export DCDF_K2R_LIB=$PWD/dcdf_amd/libdcdf_k2r_$v.so; fi
  for i in 1 2; do $B > $O/time_${v}_$i.json 2> $O/time_${v}_$i.err || exit 1; done
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$v -- $B > $O/fetch_$v.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$v -- $B > $O/write_$v.log 2>&1 || exit 1
done
python3 - <<'PY'
import json,glob,csv,collections
for f in sorted(glob.glob('gpurun_out/r04f/time_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'frac %.4f'%d['roofline']['frac'])
for d in sorted(glob.glob('gpurun_out/r04f/fetch_*/')+glob.glob('gpurun_out/r04f/write_*/')):
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_encode' in r['Kernel_Name']: agg[r['Dispatch_Id']][r['Counter_Name']]+=float(r['Counter_Value'])
    if agg:
        ids=sorted(agg,key=int); k=list(agg[ids[0]])[0]
        v=sum(agg[i][k] for i in ids)/len(ids)
        print(d.split('/')[-2], k, 'KiB %.0f'%v, '-> GB %.2f'%(v*1024*(2 if k=='FETCH_SIZE' else 1)/1e9), '(FETCH doubled per the guide)' if k=='FETCH_SIZE' else '')
PY
