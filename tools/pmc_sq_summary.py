#!/usr/bin/env python3
"""Per-kernel averages of the SQ counters of one rocprofv3 --pmc pass (counter_collection.csv) as a table.
usage: pmc_sq_summary.py <dir> [min launches]"""
import collections
import csv
import glob
import os
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for k in agg.values() for c in k})
print("| kernel | launches | " + " | ".join(counters) + " |")
print("|---|---|" + "---|" * len(counters))
rows = []
for k, v in agg.items():
    n = max(len(x) for x in v.values())
    rows.append((-sum(v.get("SQ_WAVE_CYCLES", [0])), k, n, [sum(v[c]) / len(v[c]) if c in v else float("nan") for c in counters]))
for _, k, n, vals in sorted(rows)[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print("| `%s` | %d | " % (k, n) + " | ".join("%.4g" % x for x in vals) + " |")
