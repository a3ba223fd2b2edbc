#!/usr/bin/env python3
"""Loss trajectory of the bench's training step over many steps on its ONE fixed synthetic batch, per operand mode - found while
sampling board power over 500 replays: the replay check of bench.py reported packed weights != a fresh pack after ~250 steps.
Prints the loss every 10 steps, the first step with a non-finite loss or parameter, and the largest |parameter| at the end.
usage: python3 tools/long_run.py [steps] [mode ...]      modes: f16x3 (default) f32 bf16x6"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_kaldi_resnet_amd  # noqa: E402,F401
from pytorch_kaldi_resnet_amd import ops  # noqa: E402
from pytorch_kaldi_resnet_amd.model import NeuralSpeakerModel  # noqa: E402
from pytorch_kaldi_resnet_amd.optim import FlatSGD  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
modes = sys.argv[2:] or ["f16x3", "f32"]
LR = float(os.environ.get("LR", "0.1"))
B = int(os.environ.get("B", "256"))
dev = torch.device("cuda", 0)
for mode in modes:
    ops.SPLIT = ops.MFMA_MODES[mode]
    torch.manual_seed(0)
    model = NeuralSpeakerModel(1211, 80, "mean+std", "AAM", 0.2, 30, arch="resnet34").to(dev)
    model.train()
    opt = FlatSGD(model, LR, momentum=0.9, weight_decay=5e-4)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    x = torch.randn(B, 80, 300, device=dev, generator=gen)
    y = torch.randint(0, 1211, (B,), device=dev, generator=gen)
    eng = model.engine()
    bad = None
    traj = []
    for i in range(steps):
        opt.zero_grad(set_to_none=True)
        loss, _, _ = eng.loss_and_grad(x, y, None)
        opt.step()
        if i % 10 == 0 or i == steps - 1:
            lv = float(loss)
            traj.append("%d:%.4g" % (i, lv))
            finite = all(bool(torch.isfinite(p).all()) for p in model.parameters())
            if bad is None and (lv != lv or abs(lv) == float("inf") or not finite):
                bad = i
                names = [n for n, p in model.named_parameters() if not bool(torch.isfinite(p).all())]
                print("%s: first non-finite state seen at step %d (loss %s); non-finite parameters: %s" % (mode, i, lv, names[:6]))
                break
    print("%s lr %g: %s" % (mode, LR, " ".join(traj)))
    if bad is None:
        print("%s: finite through %d steps; max |parameter| %.3g" % (mode, steps, max(float(p.abs().max()) for p in model.parameters())))
