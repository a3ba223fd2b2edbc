#!/usr/bin/env python3
"""Per-launch time of the pipelined 3x3 convolutions of ResNet-34 (forward with fused input BN + statistics; pair-input data
gradient with masked add + BatchNorm-backward statistics) in the 32x32x16 and the 16x16x32 form, 60 launches back to back."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytorch_kaldi_resnet_amd  # noqa
from pytorch_kaldi_resnet_amd import ops
ops.SPLIT = 3
B = 256
torch.manual_seed(0)
for C, H, W in ((64, 40, 150), (128, 20, 75), (256, 10, 38)):
    x = torch.randn(B, H, W, C, device="cuda")
    w = torch.randn(C, C, 3, 3, device="cuda") * 0.05
    wpk, wpk_t = ops.pack_conv_weight(w), ops.pack_conv_weight(w, True)
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    xs = ops._amax_fwd_fallback(x, (sc, sh))
    dy = torch.randn(B, H, W, C, device="cuda") * 1e-3
    slot = ops.absmax_into(dy, torch.zeros(1, device="cuda", dtype=torch.int32))
    dyp = dy.clone()      # timing only: any bit pattern stands in for a pair tensor
    dadd = torch.randn(B, H, W, C, device="cuda") * 1e-3
    m2 = torch.randint(-2 ** 31, 2 ** 31 - 1, (B * H * W * (C // 32),), device="cuda", dtype=torch.int32)
    raw = torch.randn(B, H, W, C, device="cuda")
    bn4 = torch.stack([torch.randn(C, device="cuda") * 0.1, torch.rand(C, device="cuda") + 0.5, torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1])
    out = torch.empty(B, H, W, C, device="cuda")
    for m16 in (False, True):
        ops.PIPE_M16 = m16
        for name, fn in (("fwd", lambda: ops.conv_fwd(x, wpk, C, 3, 1, in_affine=(sc, sh), stats=True, in_amax=xs)),
                         ("dgrad-pair", lambda: ops.conv_dgrad(dyp, wpk_t, C, 3, 1, (H, W), add=dadd, add_mask=m2, bn_bwd=(raw, None, bn4, m2), in_amax=slot, in_presplit=True))):
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(60):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 60
            print("C%-3d %dx%d %-10s m16=%d  %.3f ms  %.0f TFLOP/s (fp32 work)" % (C, H, W, name, m16, ms, 2.0 * B * H * W * C * C * 9 / ms * 1e-9), flush=True)
