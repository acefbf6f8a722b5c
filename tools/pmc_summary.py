#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv) per kernel (template arguments kept):
mean counter value per dispatch.  usage: pmc_summary.py <dir> [<dir> ...]"""
import csv, glob, re, sys, collections

out = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for d in sys.argv[1:]:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            name = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])
            name = re.sub(r"^void ", "", name)
            name = re.sub(r"\(mvp_\w+( const\*)?(, [^)]*)?\)$", "", name)[:64]
            out[name][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[name][row["Counter_Name"]].add((path, row["Dispatch_Id"]))
rows = []
for name, cs in out.items():
    n = max(len(v) for v in disp[name].values())
    rows.append((name, n, {k: v / max(len(disp[name][k]), 1) for k, v in cs.items()}))
for name, n, cs in sorted(rows, key=lambda r: -r[1] * sum(r[2].values())):
    print(f"{name:66s} n={n:4d} " + " ".join(f"{k}={v:.5g}" for k, v in sorted(cs.items())))
