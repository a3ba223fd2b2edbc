#!/usr/bin/env python3
"""Achieved HBM bandwidth per kernel: bytes per launch from the PMC passes (profiles/pmc_traffic.json, FETCH_SIZE x2 +
WRITE_SIZE) divided by the average launch duration of the rocprofv3 --kernel-trace --stats run of the same command
(profiles/r01_rocprofv3_bench_split/kernel_stats.csv).  Prints a markdown table."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernels"]
stats = {}
for row in csv.DictReader(open(os.path.join(ROOT, "profiles", "r01_rocprofv3_bench_split", "kernel_stats.csv"))):
    name = re.sub(r"^void ", "", row["Name"])
    name = re.sub(r"\(.*$", "", name)
    stats[name] = (int(row["Calls"]), float(row["AverageNs"]), float(row["Percentage"]))
rows = []
for name, (calls, avg_ns, pct) in stats.items():
    ent = pmc.get(name)
    if not ent or pct < 0.05:
        continue
    gbs = ent["hbm_bytes_per_launch"] / avg_ns          # bytes / ns = GB/s
    rows.append((pct, name, avg_ns / 1e3, ent["hbm_bytes_per_launch"] / 1e6, gbs))
rows.sort(reverse=True)
print("| Kernel | share of GPU time | avg launch (us) | HBM MB / launch | achieved HBM GB/s | of 8 TB/s |")
print("|---|---|---|---|---|---|")
for pct, name, us, mb, gbs in rows:
    print("| `%s` | %.1f %% | %.1f | %.1f | %.0f | %.2f |" % (name, pct, us, mb, gbs, gbs / 8000.0))
