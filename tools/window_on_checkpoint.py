#!/usr/bin/env python3
"""f16x3 operand-scale windows on a TRAINED checkpoint: load a checkpoint written by
scripts/train_resnet.py, take batches of the training corpus and run one training step per batch with the window counters
on (Engine.window_counts -> spk_f16_window_count on every tensor an f16x3 matrix-core kernel stages), then the same-forward
backward comparison of tests/test_fullsize_gpu.py (f16x3 and the native fp32 instruction against the exact split) on these
weights.  GPU box only.  usage: window_on_checkpoint.py <checkpoint> <train.scp> <utt2spkid> [batch] [frames] [nbatches]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_kaldi_resnet_amd  # noqa: E402,F401
from pytorch_kaldi_resnet_amd import ops  # noqa: E402
from pytorch_kaldi_resnet_amd.ingest import NativeTrainLoader  # noqa: E402
from pytorch_kaldi_resnet_amd.model import NeuralSpeakerModel  # noqa: E402

ck_path, scp, u2s = sys.argv[1:4]
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 64
frames = int(sys.argv[5]) if len(sys.argv) > 5 else 200
nb = int(sys.argv[6]) if len(sys.argv) > 6 else 4
ck = torch.load(ck_path, map_location="cpu", weights_only=True)
sd = {k[7:] if k.startswith("module.") else k: v for k, v in ck["state_dict"].items()}
spk = sd["last.weight"].shape[0]
m = NeuralSpeakerModel(spk, 80, "mean+std", "AAM", 0.2, 30, arch=ck.get("arch", "resnet34"))
m.load_state_dict(sd)
m = m.cuda().train()
eng = m.engine()
print("checkpoint %s: epoch %s, %d speakers; best_acc1 %s" % (ck_path, ck.get("epoch"), spk, float(ck.get("best_acc1", -1))))
loader = NativeTrainLoader(scp, u2s, frames, batch, seed=3, threads=4, drop_last=True, device="cuda:0")
buf0 = [b.clone() for b in m.buffers()]


def run(x, y, fwd, bwd=None, count=False):
    ops.SPLIT, ops.SPLIT_BWD = ops.MFMA_MODES[fwd], (ops.MFMA_MODES[bwd] if bwd else None)
    eng.dirty = True
    for b, b0 in zip(m.buffers(), buf0):
        b.copy_(b0)
    for p in m.parameters():
        p.grad = None
    if count:
        eng.window_counts = torch.zeros(4, device="cuda", dtype=torch.int64)
    loss, logits, _ = eng.loss_and_grad(x, y)
    torch.cuda.synchronize()
    c = None
    if count:
        c, eng.window_counts = eng.window_counts.tolist(), None
    return float(loss), m.flat_grads().clone(), c


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


_pool_bwd = ops.stats_pool_bwd


def _spy(x, gout, mode, amax_out=None):
    d = _pool_bwd(x, gout, mode, amax_out=amax_out)
    a = d.abs().flatten()
    q = torch.quantile(a[::97].float(), torch.tensor([0.5, 0.99, 0.9999], device=a.device))
    print("  pooling gradient |d|: max %.3g, 99.99 %% %.3g, 99 %% %.3g, median %.3g, finite %s; min pooled mean %.3g" % (
        float(a.max()), float(q[2]), float(q[1]), float(q[0]), bool(torch.isfinite(a).all()), float(x.mean(dim=2).min())))
    return d


ops.stats_pool_bwd = _spy
tot = [0, 0, 0, 0]
for i, (x, y) in enumerate(loader):
    if i >= nb:
        break
    x, y = x.clone(), y.clone()
    la, g_ex, _ = run(x, y, "bf16x6")
    _, g_h, _ = run(x, y, "bf16x6", "f16x3")
    _, g_f, _ = run(x, y, "bf16x6", "f32")
    eng.bound_log = []
    lb, _, c = run(x, y, "f16x3", count=True)
    print("  first scales (bound, true absmax): " + " ".join("%dch:(%.3g, %.3g)" % (co, float(e.view(torch.float32)), float(t.view(torch.float32))) for co, k, e, t in eng.bound_log[:4]))
    ratios = [(co, k, float(e.view(torch.float32)) / max(float(t.view(torch.float32)), 1e-45)) for co, k, e, t in eng.bound_log]
    eng.bound_log = None
    print("  BatchNorm-backward bound / true absmax per convolution (backward order): " + " ".join("%dch:%.3g" % (co, r) for co, k, r in ratios))
    tot = [a + b for a, b in zip(tot, c)]
    g_hh = m.flat_grads().clone()
    from pytorch_kaldi_resnet_amd.parallel import stage_slices
    print("  per stage (same forward) f16x3 vs exact: " + ", ".join("%s %.2e" % (n, rel(g_h[lo:hi], g_ex[lo:hi])) for n, (lo, hi) in sorted(stage_slices(m).items())))
    print("batch %d: loss %.5f (exact split) %.5f (f16x3); same-forward backward vs exact split: f16x3 %.2e, native fp32 instruction %.2e; "
          "windows: %d staged values, %d saturated, %.3f %% low term subnormal, %.4f %% high term subnormal" % (
              i, la, lb, rel(g_h, g_ex), rel(g_f, g_ex), c[0], c[1], 100.0 * c[2] / c[0], 100.0 * c[3] / c[0]))
print("total: %d staged values, %d saturated, %.3f %% low term subnormal, %.4f %% high term subnormal" % (
    tot[0], tot[1], 100.0 * tot[2] / tot[0], 100.0 * tot[3] / tot[0]))
assert tot[1] == 0, "saturated values under a scale slot"
