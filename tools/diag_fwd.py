#!/usr/bin/env python3
"""Where does the HIP training forward lose accuracy?  (diagnostic, GPU box)

For every BatchNorm of the trunk: the normalised pre-activation z = bn(conv(...)) of the HIP path and of the fp32 CPU
oracle against the fp64 oracle - max |dz|, rms |dz| and the number of elements whose SIGN differs from fp64 (each such
element is a ReLU-mask flip: it moves that layer's gradient by one whole element, ~1/sqrt(N) of its norm).
Usage: python tools/diag_fwd.py [golden case] [f32|bf16x6]
"""
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
from pytorch_kaldi_resnet_amd import ops  # noqa: E402
from pytorch_kaldi_resnet_amd.model import NeuralSpeakerModel  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "r34_softmax_mean_f40"
if len(sys.argv) > 2:
    ops.SPLIT = ops.MFMA_MODES[sys.argv[2]]
meta = json.load(open(os.path.join(ROOT, "tests/golden", name + ".json")))
npst = W.make_state(meta["seed"], meta["spk_num"], meta["feat_dim"], meta["pooling"], meta["loss"], meta["arch"])
x, y = W.make_input(meta["seed"] + 1, meta["batch"], meta["feat_dim"], meta["frames"], meta["spk_num"])


def oracle_z(dtype):
    """z of every F.batch_norm call of the oracle's training forward, in call order."""
    st = O.to_torch_state(npst)
    st = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in st.items()}
    zs = []
    real = O.F.batch_norm

    def tap(*a, **k):
        out = real(*a, **k)
        if out.dim() == 4:
            zs.append(out.detach().double())
        return out

    O.F.batch_norm = tap
    try:
        O.forward(st, torch.from_numpy(x).to(dtype), torch.from_numpy(y), meta["pooling"], meta["loss"], meta["arch"], train=True)
    finally:
        O.F.batch_norm = real
    return zs


z64, z32 = oracle_z(torch.float64), oracle_z(torch.float32)

m = NeuralSpeakerModel(meta["spk_num"], meta["feat_dim"], meta["pooling"], meta["loss"], 0.2, 30, arch=meta["arch"])
m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in npst.items()})
m = m.cuda().train()
eng = m.engine()
with torch.no_grad():
    _, saved = eng.forward_train(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())


def z_of(raw, t4):
    mean, invstd, scale, shift = [t.double() for t in t4]
    g = scale / invstd
    b = shift + mean * scale
    z = (raw.double() - mean) * invstd * g + b
    return z.permute(0, 3, 1, 2).cpu()


zh, labels = [], []
zh.append(z_of(saved["raw0"], eng.stem_bn.t4))
labels.append("stem.bn1")
bi = 0
for li in range(1, 5):
    for k in range(len(getattr(m.res, "layer%d" % li))):
        b, rec = eng.blocks[bi], saved["blocks"][bi]
        for i, (raw, bn) in enumerate(zip(rec["raws"], b.bns)):
            zh.append(z_of(raw, bn.t4))
            labels.append("layer%d.%d.bn%d" % (li, k, i + 1))
        if b.ds is not None:
            zh.append(z_of(rec["rawd"], b.ds[1].t4))
            labels.append("layer%d.%d.ds" % (li, k))
        bi += 1
assert len(zh) == len(z64) == len(z32), (len(zh), len(z64), len(z32))
print("%-18s %9s | %10s %10s %6s | %10s %10s %6s" % ("bn", "N", "hip max", "hip rms", "flips", "cpu32 max", "cpu32 rms", "flips"))
tot_h = tot_c = 0
for lb, a, b, c in zip(labels, zh, z32, z64):
    dh, dc = (a - c), (b - c)
    fh = int(((a > 0) != (c > 0)).sum())
    fc = int(((b > 0) != (c > 0)).sum())
    tot_h += fh
    tot_c += fc
    print("%-18s %9d | %10.2e %10.2e %6d | %10.2e %10.2e %6d" % (lb, c.numel(), float(dh.abs().max()), float(dh.pow(2).mean().sqrt()), fh,
                                                                 float(dc.abs().max()), float(dc.pow(2).mean().sqrt()), fc))
print("total sign flips vs fp64: hip %d, cpu fp32 %d" % (tot_h, tot_c))
