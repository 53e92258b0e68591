import csv, collections, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        if "k_encode" in r["Kernel_Name"]:
            agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if not agg:
        continue
    last = sorted(agg, key=int)[-1]
    print(f.split("/")[-3] if "/" in f else f, {k: int(v) for k, v in agg[last].items()})
