"""Condenses the rocprofv3 outputs of tools/collect_profiles.sh into small files meant for profiles/."""
import collections
import csv
import glob
import json
import sys

out, tag = sys.argv[1], sys.argv[2]


def counters(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_encode" in r["Kernel_Name"]:
                agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if not agg:
        return {}
    ids = sorted(agg, key=int)
    n = len(ids)
    return {k: sum(agg[i][k] for i in ids) / n for k in agg[ids[0]]}, n


stats = None
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    stats = open(f).read()
    open(out + "/%s_kernel_stats.csv" % tag, "w").write(stats)
res = {"tag": tag}
for sub in ("fetch", "write", "sq", "sq2"):
    c = counters(sub)
    if c:
        res[sub] = {k: v for k, v in c[0].items()}
        res[sub + "_dispatches_averaged"] = c[1]
line = None
for l in open(out + "/stats.log"):
    if l.startswith("{"):
        line = json.loads(l)
if line:
    res["bench_line_under_profiler"] = line
if "fetch" in res and "write" in res:
    # MI355X_MICROARCH.md, HBM section: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half
    # of the bytes of a wide (16 B/lane) coalesced read stream -> doubled; WRITE_SIZE is exact for wide stores
    # (our byte-granular stores are not calibrated: treat the write side as indicative).
    fetch = res["fetch"]["FETCH_SIZE"] * 1024 * 2
    write = res["write"]["WRITE_SIZE"] * 1024
    res["hbm_bytes_per_launch"] = fetch + write
    res["hbm_fetch_bytes_per_launch_corrected_x2"] = fetch
    res["hbm_write_bytes_per_launch"] = write
    if line:
        res["algorithmic_bytes_per_launch"] = line["roofline"]["algorithmic_bytes"]
json.dump(res, open(out + "/%s_pmc_summary.json" % tag, "w"), indent=1)
if line and "hbm_bytes_per_launch" in res:
    # what bench.py quotes as roofline.traffic -- only for this workload and these kernel sources
    json.dump({"tag": tag, "workload_key": line["config"].get("workload_key"), "source_sha": line["roofline"].get("source_sha"),
               "hbm_bytes_per_launch": res["hbm_bytes_per_launch"], "fetch_bytes_x2": res["hbm_fetch_bytes_per_launch_corrected_x2"],
               "write_bytes": res["hbm_write_bytes_per_launch"], "algorithmic_bytes_per_launch": res.get("algorithmic_bytes_per_launch")},
              open(out + "/traffic_latest.json", "w"), indent=1)
print(json.dumps({k: res[k] for k in res if k.startswith("hbm") or k.startswith("alg")}))
