#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, separate passes) into per-kernel HBM traffic.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE reports
exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) -> doubled; WRITE_SIZE is exact for
16-byte-per-lane streaming stores.
usage: pmc_summary.py <dir with *_counter_collection.csv of the FETCH pass> <dir of the WRITE pass> <out.json>
"""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
                agg[name].append(float(r["Counter_Value"]))
    return agg


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f = fetch.get(k, [])
    w = write.get(k, [])
    if not f or not w:
        continue
    fb = 2.0 * 1024.0 * sum(f) / len(f)
    wb = 1024.0 * sum(w) / len(w)
    out[k] = {"launches": len(f), "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
              "hbm_bytes_per_launch": fb + wb}
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pytorch-kaldi-resnet_amd"))
import build as _build  # noqa: E402

json.dump({"csrc_fingerprint": _build.csrc_fingerprint(),
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 bench.py --steps 2 "
                     "--warmup 1 --no-cpu-baseline --no-roofline`; FETCH_SIZE doubled per the gfx950 note",
           "kernels": out}, open(sys.argv[3], "w"), indent=1)
print("wrote", sys.argv[3], len(out), "kernels")
