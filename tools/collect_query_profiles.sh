#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel stats of the decode bench (tools/bench_query.py, 200 k queries as bench.py's
# "decode" sample), then separate FETCH_SIZE / WRITE_SIZE passes (MI355X_MICROARCH.md: separate --pmc passes, FETCH_SIZE doubled),
# condensed into profiles-ready files.   usage: tools/collect_query_profiles.sh <tag>     outputs under gpurun_out/qprof_<tag>/
set -o pipefail
TAG=${1:-rXX}
Q=${QUERIES:-200000}
OUT=gpurun_out/qprof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT/stats $OUT/fetch $OUT/write
B="python3 tools/bench_query.py --queries $Q --batch 100000 --cpu-sample 0 --no-check --no-raster-level"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B > $OUT/write.log 2>&1 || exit 1
python3 - "$OUT" "$TAG" "$Q" <<'PY'
import collections, csv, glob, json, os, sys
out, tag, q = sys.argv[1], sys.argv[2], int(sys.argv[3])
sys.path.insert(0, "tools")
import bench_query
QK = ("k_window_wave", "k_search", "k_raster", "k_scan", "k_get", "k_cell")
def counters(sub, name):
    agg = collections.defaultdict(float)
    for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and any(k in r["Kernel_Name"] for k in QK):
                agg[r["Kernel_Name"].split("(")[0][:60]] += float(r["Counter_Value"])
    return agg
fe, wr = counters("fetch", "FETCH_SIZE"), counters("write", "WRITE_SIZE")
per = {k: {"fetch_bytes_x2": fe.get(k, 0) * 1024 * 2, "write_bytes": wr.get(k, 0) * 1024} for k in sorted(set(fe) | set(wr))}
fetch, write = sum(v["fetch_bytes_x2"] for v in per.values()), sum(v["write_bytes"] for v in per.values())
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    open(out + "/%s_query_kernel_stats.csv" % tag, "w").write(open(f).read())
line = None
for l in open(out + "/stats.log"):
    if l.startswith("{"):
        line = json.loads(l)
rec = {"tag": tag, "queries": q, "source_sha": bench_query.query_source_sha(), "hbm_bytes_all_query_kernels": fetch + write,
       "fetch_bytes_x2": fetch, "write_bytes": write, "per_kernel": per,
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/bench_query.py, summed over the launches of the "
               "query kernels (chunk-level entry points, the ones the roofline's kernel time belongs to); FETCH_SIZE x 1024 x 2 "
               "(gfx950 reports half of a wide read stream; the walks' narrow reads are not calibrated: indicative)"}
if line:
    rec["bench_line_under_profiler"] = {k: line[k] for k in ("value", "queries_per_s", "fill_window", "search_window", "roofline")}
json.dump(rec, open(out + "/%s_query_pmc_summary.json" % tag, "w"), indent=1)
json.dump({k: rec[k] for k in ("tag", "queries", "source_sha", "hbm_bytes_all_query_kernels", "fetch_bytes_x2", "write_bytes", "per_kernel")},
          open(out + "/query_traffic_latest.json", "w"), indent=1)
print(json.dumps({k: rec[k] for k in ("hbm_bytes_all_query_kernels", "fetch_bytes_x2", "write_bytes")}))
PY
