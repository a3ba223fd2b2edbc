#!/usr/bin/env python3
"""Run a few convolution launches of given shapes (for rocprofv3 --pmc / kernel-trace passes)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_kaldi_resnet_amd  # noqa: E402,F401
from pytorch_kaldi_resnet_amd import ops  # noqa: E402

B = int(os.environ.get("B", "256"))
reps = int(os.environ.get("REPS", "3"))
shapes = {"L1": (32, 32, 80, 300), "L2": (64, 64, 40, 150), "L3": (128, 128, 20, 75), "L4": (256, 256, 10, 38)}
which = sys.argv[1:] or ["L1", "L3", "L4"]
for name in which:
    Cin, Cout, H, W = shapes[name]
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    wpk = ops.pack_conv_weight(w)
    dy = torch.randn(B, H, W, Cout, device="cuda")
    dw = torch.empty_like(w)
    out = torch.empty(B, H, W, Cout, device="cuda")
    for _ in range(reps):
        ops.conv_fwd(x, wpk, Cout, 3, 1, stats=True, out=out)
        ops.conv_wgrad(x, dy, dw, 3, 1)
    torch.cuda.synchronize()
print("done")
