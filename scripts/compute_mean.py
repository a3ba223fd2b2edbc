#!/usr/bin/env python3
"""usage: compute_mean.py <embeddings text ark> <mean.vec>   (reference scripts/compute_mean.py)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_kaldi_resnet_amd  # noqa: E402,F401
from pytorch_kaldi_resnet_amd import scoring  # noqa: E402

if __name__ == "__main__":
    mean = scoring.compute_mean(sys.argv[1], sys.argv[2])
    print("saved mean of {} in {}".format(sys.argv[1], sys.argv[2]))
