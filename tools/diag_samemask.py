#!/usr/bin/env python3
"""Per-tensor same-mask gradient error (HIP and CPU fp32 against the fp64 oracle under each one's own ReLU masks).
usage: python tools/diag_samemask.py [case] [mode]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import masked, spk_oracle as O, weights as W  # noqa: E402
import pytorch_kaldi_resnet_amd  # noqa: E402,F401
from pytorch_kaldi_resnet_amd import ops  # noqa: E402
import test_model_gpu as T  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "r34_softmax_mean_f40"
if len(sys.argv) > 2:
    ops.SPLIT = ops.MFMA_MODES[sys.argv[2]]
gold = os.path.join(ROOT, "tests", "golden")
meta = json.load(open(os.path.join(gold, name + ".json")))
m, npst = T.build(None, meta)
x, y = W.make_input(meta["seed"] + 1, meta["batch"], meta["feat_dim"], meta["frames"], meta["spk_num"])
m.train()
loss, hip, masks_hip = T.hip_step_with_masks(m, torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
kw = dict(pooling=meta["pooling"], loss=meta["loss"], arch=meta["arch"])
st = O.to_torch_state(npst)
keys = O.trainable_keys(st)
for k in keys:
    st[k].requires_grad_(True)
lo, masks32 = masked.record_masks(st, torch.from_numpy(x), torch.from_numpy(y), **kw)
g32 = dict(zip(keys, [g.double() for g in torch.autograd.grad(O.cross_entropy(lo, torch.from_numpy(y)), [st[k] for k in keys])]))
_, rh = masked.grads(npst, x, y, masks=masks_hip, **kw)
_, r32 = masked.grads(npst, x, y, masks=masks32, **kw)
tot = float(torch.cat([rh[k].reshape(-1) for k in keys]).norm())
print("%-40s %10s %10s %10s" % ("tensor", "|g|/|G|", "hip", "cpu32"))
for k in keys:
    n = float(rh[k].norm())
    print("%-40s %10.2e %10.2e %10.2e" % (k, n / tot, float((hip[k] - rh[k]).norm()) / tot, float((g32[k] - r32[k]).norm()) / tot))
