#!/usr/bin/env python3
"""Per-step view of a rocprofv3 --kernel-trace CSV of bench.py: kernels of the last full step in launch order
(backbone blocks elided), per-kernel totals per step.  usage: step_trace.py <kernel_trace.csv> [--all]"""
import csv, re, sys
from collections import OrderedDict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "patch_gather" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
step = rows[a:b]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n)[:78]


tot = OrderedDict()
for r in step:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    e = tot.setdefault(short(r["Kernel_Name"]), [0, 0.0])
    e[0] += 1; e[1] += d
wall = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
ksum = sum(v[1] for v in tot.values())
print(f"step wall {wall:.1f} us, kernel sum {ksum:.1f} us, {len(step)} launches")
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:8.1f} us {100 * t / ksum:5.1f}%  x{c:<3d} avg {t / c:7.2f}  {n}")
if "--all" in sys.argv:
    print("---- launch order (tail after the last backbone GEMM)")
    last = max(i for i, r in enumerate(step) if "layernorm" in r["Kernel_Name"])
    for k, r in enumerate(step[last:], last):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(f"{k:4d} {d:7.2f} {short(r['Kernel_Name'])}")
