#!/usr/bin/env python3
"""Per-parameter gradient error of the HIP model vs the fp64 CPU oracle (diagnostic, GPU box)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import spk_oracle as O  # noqa: E402
from oracle import weights as W  # noqa: E402
import pytorch_kaldi_resnet_amd  # noqa: E402,F401
from pytorch_kaldi_resnet_amd.model import NeuralSpeakerModel  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "r34_softmax_mean_f40"
meta = json.load(open(os.path.join(ROOT, "tests/golden", name + ".json")))
npst = W.make_state(meta["seed"], meta["spk_num"], meta["feat_dim"], meta["pooling"], meta["loss"], meta["arch"])
m = NeuralSpeakerModel(meta["spk_num"], meta["feat_dim"], meta["pooling"], meta["loss"], 0.2, 30, arch=meta["arch"])
m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in npst.items()})
m = m.cuda().train()
x, y = W.make_input(meta["seed"] + 1, meta["batch"], meta["feat_dim"], meta["frames"], meta["spk_num"])
xg, yg = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
loss = torch.nn.functional.cross_entropy(m(xg, yg), yg)
loss.backward()
hip = {n: p.grad.detach().cpu().double() for n, p in m.named_parameters()}
kw = dict(pooling=meta["pooling"], loss=meta["loss"], arch=meta["arch"])


def oracle_grads(dtype):
    st = O.to_torch_state(npst)
    st = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in st.items()}
    keys = O.trainable_keys(st)
    for k in keys:
        st[k].requires_grad_(True)
    lo = O.forward(st, torch.from_numpy(x).to(dtype), torch.from_numpy(y), train=True, **kw)
    lv = O.cross_entropy(lo, torch.from_numpy(y))
    gs = torch.autograd.grad(lv, [st[k] for k in keys])
    return {k: v.double() for k, v in zip(keys, gs)}


g32, g64 = oracle_grads(torch.float32), oracle_grads(torch.float64)
print("%-42s %10s %10s %10s" % ("param", "|g64|", "hip_err", "cpu32_err"))
for n in hip:
    nr = float(g64[n].norm())
    print("%-42s %10.3e %10.3e %10.3e" % (n, nr, float((hip[n] - g64[n]).norm()) / (nr + 1e-30),
                                          float((g32[n] - g64[n]).norm()) / (nr + 1e-30)))
