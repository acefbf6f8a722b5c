#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv) per kernel name."""
import csv, glob, sys, collections
out = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0][:60]
        out[name][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (row["Dispatch_Id"],)
        if key not in seen:
            seen.add(key); cnt[name] += 1
for name, cs in sorted(out.items(), key=lambda kv: -sum(kv[1].values())):
    print(f"{name:62s} n={cnt[name]:4d} " + " ".join(f"{k}={v / max(cnt[name],1):.4g}" for k, v in sorted(cs.items())))
