#!/usr/bin/env python3
"""Run one training step + one predict for several (arch, feat, frames, batch) shapes with heuristic tiles
(shapes outside tile_table.json) and report step time - guards the tile chooser's limits."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_kaldi_resnet_amd  # noqa: E402,F401
from pytorch_kaldi_resnet_amd.model import NeuralSpeakerModel  # noqa: E402
from pytorch_kaldi_resnet_amd.optim import FlatSGD  # noqa: E402

cases = [("resnet34", 80, 200, 256), ("resnet34", 80, 400, 128), ("resnet34", 80, 333, 64), ("resnet34", 40, 200, 256),
         ("resnet34", 80, 1001, 8), ("resnet34", 80, 64, 32), ("resnet101", 80, 300, 64), ("resnet101", 80, 203, 32),
         ("resnet18", 40, 120, 16), ("resnet50", 80, 200, 16)]
for arch, F, T, B in cases:
    m = NeuralSpeakerModel(1211, F, "mean+std", "AAM", arch=arch).cuda().train()
    opt = FlatSGD(m, 0.01, momentum=0.9, weight_decay=5e-4)
    x = torch.randn(B, F, T, device="cuda")
    y = torch.randint(0, 1211, (B,), device="cuda")
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        loss, _, _ = m.engine().loss_and_grad(x, y)
        opt.step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        loss, _, _ = m.engine().loss_and_grad(x, y)
        opt.step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    m.eval()
    with torch.no_grad():
        e = m.predict(x)
    torch.cuda.synchronize()
    print("%-9s F=%d T=%4d B=%3d  train %.1f ms (%.0f utt/s)  loss %.3f  emb finite %s" % (
        arch, F, T, B, dt * 1e3, B / dt, float(loss), bool(torch.isfinite(e).all())), flush=True)
    del m, opt, x, y
    torch.cuda.empty_cache()
