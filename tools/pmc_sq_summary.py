#!/usr/bin/env python3
"""Per-kernel averages of the SQ counters of one rocprofv3 --pmc pass (counter_collection.csv) as a table.
usage: pmc_sq_summary.py <dir> [rows] [--by-grid]     (--by-grid: one row per (kernel, grid size) - separates the layers a
kernel name is launched for; VGPR / LDS of the dispatch are printed once per row)"""
import collections
import csv
import glob
import os
import sys

BY_GRID = "--by-grid" in sys.argv
if BY_GRID:
    sys.argv.remove("--by-grid")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        if BY_GRID:
            name += " grid %s vgpr %s+%s lds %s" % (r.get("Grid_Size", "?"), r.get("VGPR_Count", "?"), r.get("Accum_VGPR_Count", "?"),
                                                  r.get("LDS_Block_Size", "?"))
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
