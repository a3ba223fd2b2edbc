#!/usr/bin/env python3
"""Cosine scoring with mean subtraction (reference scripts/cosine_score.py, same flags and file formats)."""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_kaldi_resnet_amd  # noqa: E402,F401
from pytorch_kaldi_resnet_amd import kaldi_io, scoring  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--mean", type=str)
    ap.add_argument("--enroll", type=str)
    ap.add_argument("--test", type=str)
    ap.add_argument("--trials", type=str)
    ap.add_argument("--score-file", type=str)
    ap.add_argument("--backend", choices=["host", "hip"], default="host", help="hip: normalisation and the trial dot products run on the GPU")
    a = ap.parse_args()
    if not (a.mean and os.path.exists(a.mean)):
        print("mean file missing")
        sys.exit(0)
    mean = kaldi_io.read_vec_flt(a.mean)
    print("loaded mean from {}".format(a.mean))
    en = scoring.read_embeddings(a.enroll)
    te = en if a.test == a.enroll else scoring.read_embeddings(a.test)
    scoring.cosine_score(en, te, a.trials, mean, a.score_file, backend=a.backend)
    print("saved scores of {} in {}".format(a.trials, a.score_file))
