#!/usr/bin/env python3
"""Synthetic fbank corpus in Kaldi format: N(0,1) frames + a per-speaker mean offset,
one FM ark + scp lists, utt2spkid, an all-pairs-subsample trials file."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_kaldi_resnet_amd  # noqa: E402,F401
from pytorch_kaldi_resnet_amd import kaldi_io  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--out", required=True)
ap.add_argument("--speakers", type=int, default=10)
ap.add_argument("--utts-per-speaker", type=int, default=100)
ap.add_argument("--min-frames", type=int, default=200)
ap.add_argument("--max-frames", type=int, default=260)
ap.add_argument("--feat-dim", type=int, default=80)
ap.add_argument("--seed", type=int, default=1234)
ap.add_argument("--trials", type=int, default=20000)
a = ap.parse_args()
os.makedirs(a.out, exist_ok=True)
rs = np.random.RandomState(a.seed)
off = 0.5 * rs.randn(a.speakers, a.feat_dim).astype(np.float32)
lines, u2s = [], []
ark = os.path.abspath(os.path.join(a.out, "feats.ark"))
with open(ark, "wb") as f:
    for s in range(a.speakers):
        for u in range(a.utts_per_speaker):
            utt = "spk%04d-utt%04d" % (s, u)
            T = rs.randint(a.min_frames, a.max_frames + 1)
            o = kaldi_io.write_mat(f, rs.randn(T, a.feat_dim).astype(np.float32) + off[s], key=utt)
            lines.append("%s %s:%d" % (utt, ark, o))
            u2s.append("%s %d" % (utt, s))
idx = rs.permutation(len(lines))
ncv = max(1, len(lines) // 20)
cv = set(idx[:ncv].tolist())
open(os.path.join(a.out, "all.scp"), "w").write("\n".join(lines) + "\n")
open(os.path.join(a.out, "train.scp"), "w").write("\n".join(l for i, l in enumerate(lines) if i not in cv) + "\n")
open(os.path.join(a.out, "cv.scp"), "w").write("\n".join(l for i, l in enumerate(lines) if i in cv) + "\n")
open(os.path.join(a.out, "utt2spkid"), "w").write("\n".join(u2s) + "\n")
utts = [l.split()[0] for l in lines]
with open(os.path.join(a.out, "trials"), "w") as f:
    for _ in range(a.trials):
        i, j = rs.randint(0, len(utts), 2)
        if i == j:
            continue
        f.write("%s %s %s\n" % (utts[i], utts[j], "target" if utts[i][:7] == utts[j][:7] else "nontarget"))
print("wrote", len(lines), "utterances to", a.out)
