#!/usr/bin/env python3
"""Dump the per-kernel summary (`top_kernels` view) of a rocprofv3 rocpd database as kernel_stats.csv.

rocprofv3 7.x writes `<name>_results.db` unless `--output-format csv` is given; the view holds the same columns as
the CSV `--stats` summary (durations in ns here, as in the CSV).  Usage: rocpd_stats.py results.db out.csv
"""
import csv
import sqlite3
import sys


def main(db, out):
    con = sqlite3.connect(db)
    cur = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for name, calls, total, avg, pct in cur:
            w.writerow([name, calls, round(total * 1e3), round(avg * 1e3, 1), round(pct, 4)])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
