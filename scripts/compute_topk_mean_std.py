#!/usr/bin/env python3
"""Cohort top-300 score statistics per utterance for adaptive S-norm (reference scripts/compute_topk_mean_std.py,
same flags and file format; --backend hip runs the normalisation, the cohort GEMM and the top-k on the GPU)."""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_kaldi_resnet_amd  # noqa: E402,F401
from pytorch_kaldi_resnet_amd import kaldi_io, scoring  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser("Configuration for data preparation")
    ap.add_argument("--mean", type=str, help="mean vec file")
    ap.add_argument("--ark-file", type=str, help="test embeddings file")
    ap.add_argument("--cohort-file", type=str, help="cohort embeddings file")
    ap.add_argument("--mean-std-file", type=str, help="file to save mean and std")
    ap.add_argument("--topk", type=int, default=300)
    ap.add_argument("--backend", choices=["host", "hip"], default="host")
    a = ap.parse_args()
    if not (a.mean and os.path.exists(a.mean)):
        print("mean file missing")
        sys.exit(0)
    mean = kaldi_io.read_vec_flt(a.mean)
    print("loaded mean from {}".format(a.mean))
    stats = scoring.topk_mean_std(scoring.read_embeddings(a.ark_file), scoring.read_embeddings(a.cohort_file), mean,
                                  a.topk, a.backend)
    scoring.write_mean_std(stats, a.mean_std_file)
    print("saved speaker mean in {}".format(a.mean_std_file))
