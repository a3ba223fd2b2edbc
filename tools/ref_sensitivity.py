#!/usr/bin/env python3
"""How far does the REFERENCE's own 5-step loss curve move under perturbations at the fp32 rounding level?
(build container only: imports /root/reference/scripts/model.py unmodified, like tools/make_golden.py)

For each golden loss-curve case the same loop as tools/make_golden.py (scripts/train_resnet.py:304-328) is run
  * as recorded (fp32, CPU),
  * in fp64 (the reference module .double()) - the distance fp32 <-> fp64 is the reference's own rounding error,
  * in fp32 with every input value multiplied by (1 + e*u), u uniform in [-1,1], e = 2^-23 (one unit in the last place:
    what a different but equally correct fp32 summation order does to every intermediate) and e = 1e-6, 8 seeds each.
Written to tests/golden/ref_sensitivity.json: per step, |loss - recorded| of every variant, plus the embedding cosine
distance after the 5 steps.  tests/test_model_gpu.py derives its loss-curve budget from these numbers: an
implementation cannot be asked to track the recorded curve more closely than the reference tracks itself.

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/ref_sensitivity.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(REF, "scripts"))
sys.dont_write_bytecode = True

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from oracle import weights as W  # noqa: E402
from tools.make_golden import ref_model  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def curve(meta, dtype=torch.float32, eps=0.0, pseed=0):
    m = ref_model(meta["spk_num"], meta["feat_dim"], meta["pooling"], meta["loss"], meta["arch"], meta["seed"])
    m = m.to(dtype).train()
    crit = nn.CrossEntropyLoss()
    opt = torch.optim.SGD(m.parameters(), meta["lr"], momentum=0.9, weight_decay=meta["wd"])
    rng = np.random.RandomState(1000 + pseed)
    losses = []
    for s in range(meta["steps"]):
        xs, ys = W.make_input(meta["seed"] + 1 + s, meta["batch"], meta["feat_dim"], meta["frames"], meta["spk_num"])
        if eps:
            xs = (xs.astype(np.float64) * (1.0 + eps * rng.uniform(-1, 1, xs.shape))).astype(np.float32)
        xs, ys = torch.from_numpy(xs).to(dtype), torch.from_numpy(ys)
        loss = crit(m(xs, ys), ys)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    x0, _ = W.make_input(meta["seed"] + 1, meta["batch"], meta["feat_dim"], meta["frames"], meta["spk_num"])
    m.eval()
    with torch.no_grad():
        emb = m.predict(torch.from_numpy(x0).to(dtype)).double().numpy()
    return np.array(losses), emb


def cosd(a, b):
    return float((1 - (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))).max())


def main():
    out = {"torch": torch.__version__, "what": __doc__.split("\n")[0], "cases": {}}
    for name in ["c1_r34_aam", "r34_softmax_mean_f40"]:
        meta = json.load(open(os.path.join(GOLD, name + ".json")))
        rec = np.load(os.path.join(GOLD, name + ".npz"))["loss_curve"]
        base, emb0 = curve(meta)
        assert np.abs(base - rec).max() < 1e-6, "the recorded curve is not reproduced: %s vs %s" % (base, rec)
        l64, e64 = curve(meta, torch.float64)
        ent = {"recorded": rec.tolist(), "fp64_minus_recorded": (l64 - rec).tolist(), "emb_cos_fp64": cosd(e64, emb0)}
        for tag, eps in (("ulp", 2.0 ** -23), ("1e-6", 1e-6)):
            d, ec = [], []
            for ps in range(8):
                lp, ep = curve(meta, eps=eps, pseed=ps)
                d.append(np.abs(lp - rec))
                ec.append(cosd(ep, emb0))
            d = np.stack(d)
            ent["perturb_" + tag] = {"eps": eps, "max_abs_dloss": d.max(0).tolist(), "rms_abs_dloss": np.sqrt((d ** 2).mean(0)).tolist(),
                                     "max_emb_cos": max(ec)}
            print(name, tag, "max |dloss| per step", d.max(0), "emb cos", max(ec))
        print(name, "fp64 - fp32", l64 - rec, "emb cos", ent["emb_cos_fp64"])
        out["cases"][name] = ent
    with open(os.path.join(GOLD, "ref_sensitivity.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
