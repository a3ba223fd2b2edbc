#!/usr/bin/env python3
"""Fixtures for the Dataset classes, produced by the REFERENCE's own scripts/datasets.py (build container only).

The reference classes use `np.int`, gone from numpy >= 1.24: the generator sets `numpy.int = int`
in-process before importing them - the reference files are not touched and do not travel; only the arrays below do.
An unbalanced 16-line scp over the 6 matrices of tests/golden/io/feats.ark (speaker 0: 1 utterance, speaker 1: 3,
speaker 2: 12) exercises the class-balancing rule (scripts/datasets.py:23-31: cap = min(500, (12+1)//2) = 6 ->
repetitions 6 / 2 / 1) and the speaker-uniform sampler (scripts/datasets.py:74-146).  Samples are drawn with
np.random.seed(...) set right before each __getitem__, so the implementation under test must consume the global numpy
RNG in the same order (utterance index, then crop start) to reproduce them.

Writes tests/golden/datasets.npz + tests/golden/io/unbalanced.scp / unbalanced.utt2spkid.
Run from the repo root:  PYTHONDONTWRITEBYTECODE=1 python tools/make_dataset_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
np.int = int                                   # the shim named in SURVEY.md section 8(c)
sys.path.insert(0, "/root/reference/scripts")
os.chdir(ROOT)

import datasets as refds  # noqa: E402  (the reference's)

IO = os.path.join("tests", "golden", "io")
base = [l.split() for l in open(os.path.join(IO, "feats.scp"))]
lines, u2s = [], []
plan = [(0, 1), (1, 3), (2, 12)]               # (speaker, utterances)
k = 0
for spk, n in plan:
    for j in range(n):
        utt = "s%d-u%02d" % (spk, j)
        lines.append("%s %s" % (utt, base[k % len(base)][1]))
        u2s.append("%s %d" % (utt, spk))
        k += 1
scp, u2sf = os.path.join(IO, "unbalanced.scp"), os.path.join(IO, "unbalanced.utt2spkid")
open(scp, "w").write("\n".join(lines) + "\n")
open(u2sf, "w").write("\n".join(u2s) + "\n")

out = {}
ds = refds.SequenceDataset(scp, u2sf, [16])
out["v1_len"] = np.array(len(ds))
out["v1_labels"] = np.asarray(ds.labels, dtype=np.int64)
out["v1_rxfiles"] = np.array([str(r) for r in ds.rxfiles])
idx = [0, 5, 6, 7, 9, len(ds) - 1]
out["v1_idx"] = np.array(idx)
for i in idx:
    np.random.seed(100 + i)
    x, y = ds[i]
    out["v1_x%d" % i] = np.ascontiguousarray(x)
    out["v1_y%d" % i] = np.asarray(y)
np.random.seed(5)
dsv = refds.SequenceDataset(scp, u2sf, [12, 20])      # two-element list: per-sample lengths drawn at construction
out["v1_var_seq_len"] = np.asarray(dsv.seq_len, dtype=np.int64)

ds2 = refds.SequenceDataset2(scp, u2sf, 14)
out["v2_len"] = np.array(len(ds2))
out["v2_labels"] = np.asarray(ds2.labels, dtype=np.int64)
out["v2_repetition"] = np.array(ds2.repetition)
idx2 = list(range(0, len(ds2), 2))
out["v2_idx"] = np.array(idx2)
for i in idx2:
    np.random.seed(200 + i)
    x, y = ds2[i]
    out["v2_x%d" % i] = np.ascontiguousarray(x)
    out["v2_y%d" % i] = np.asarray(y)
np.savez_compressed(os.path.join("tests", "golden", "datasets.npz"), **out)
print("v1 len", len(ds), "labels", out["v1_labels"], "v2 len", len(ds2), "rep", ds2.repetition)
