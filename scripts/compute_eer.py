#!/usr/bin/env python3
"""usage: compute_eer.py <scores> <trials>  -> prints EER as 'x.xx%' (reference scripts/compute_eer.py)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_kaldi_resnet_amd  # noqa: E402,F401
from pytorch_kaldi_resnet_amd import scoring  # noqa: E402

if __name__ == "__main__":
    trials = {}
    for line in open(sys.argv[2]):
        a, b, t = line.rstrip().split()
        trials[a + " " + b] = t
    scores, labels = [], []
    for line in open(sys.argv[1]):
        a, b, s = line.rstrip().split()
        if a + " " + b not in trials:
            raise Exception("Missing entry for " + a + " and " + b + " " + sys.argv[1])
        scores.append(float(s))
        labels.append(1 if trials[a + " " + b] == "target" else 0)
    eer = scoring.compute_eer(scores, labels)
    sys.stdout.write("{0:.2%}\n".format(eer))
    sys.stderr.write("eer is {0:.2%}\n".format(eer))
