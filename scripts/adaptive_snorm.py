#!/usr/bin/env python3
"""Adaptive S-norm of a score file (reference scripts/adaptive_snorm.py, same flags and file formats)."""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_kaldi_resnet_amd  # noqa: E402,F401
from pytorch_kaldi_resnet_amd import scoring  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser("Configuration for data preparation")
    ap.add_argument("--enroll", type=str, help="enroll topk mean and std file")
    ap.add_argument("--test", type=str, help="test topk mean and std file")
    ap.add_argument("--score-in", type=str, help="score in file")
    ap.add_argument("--score-out", type=str, help="score out file")
    a = ap.parse_args()
    scoring.adaptive_snorm(scoring.read_mean_std(a.enroll), scoring.read_mean_std(a.test), a.score_in, a.score_out)
    print("saved adaptive S-norm scores in {}".format(a.score_out))
