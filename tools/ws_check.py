#!/usr/bin/env python3
"""conv_ws_kernel (producer / consumer, persistent) against conv_mfma_kernel on the ResNet-34 stride-1 3x3 launch shapes:
bit-equality of every output (same MFMA order per accumulator) and launch time of both.  GPU box.
  python tools/ws_check.py [--batch 256] [--frames 300] [--small]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_kaldi_resnet_amd  # noqa: E402,F401
from pytorch_kaldi_resnet_amd import ops, tiling  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--feat", type=int, default=80)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--layers", default="1,2,3,4")
args = ap.parse_args()
B = args.batch
dev = "cuda"
assert ops.SPLIT, "split operand mode only"


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


dims = [(32, args.feat, args.frames)]
for c in (64, 128, 256):
    h, w = dims[-1][1], dims[-1][2]
    dims.append((c, (h - 1) // 2 + 1, (w - 1) // 2 + 1))
ok_all = True
for li, (C, H, W) in enumerate(dims):
    if str(li + 1) not in args.layers.split(","):
        continue
    torch.manual_seed(li)
    x = torch.randn(B, H, W, C, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    wpk, wpk_t = ops.pack_conv_weight(w), ops.pack_conv_weight(w, True)
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    flops = 2.0 * B * H * W * C * C * 9
    res = {}
    for ws in (False, True):
        ops.WS_CONV = "1" if ws else "0"
        out, st = ops.conv_fwd(x, wpk, C, 3, 1, in_affine=(sc, sh), stats=True)
        t_f = timeit(lambda: ops.conv_fwd(x, wpk, C, 3, 1, in_affine=(sc, sh), stats=True, out=out), args.reps)
        # fused BatchNorm-backward data gradient with sign masks, shortcut add and BN-backward statistics in the epilogue
        torch.manual_seed(100 + li)
        dy = torch.randn(B, H, W, C, device=dev)
        raw, raw_p, dout = (torch.randn(B, H, W, C, device=dev) for _ in range(3))
        g = torch.Generator(device=dev)
        g.manual_seed(5)
        mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (B * H * W * (C // 32),), device=dev, dtype=torch.int32, generator=g)
        mask2 = torch.randint(-2 ** 31, 2 ** 31 - 1, (B * H * W * (C // 32),), device=dev, dtype=torch.int32, generator=g)
        bn4 = torch.stack([torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) + 0.5,
                           torch.randn(C, device=dev) * 0.1])
        coef = torch.stack([torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.01, torch.randn(C, device=dev) * 0.01])
        draw, dx = torch.empty_like(raw), torch.empty_like(raw)

        def dgrad():
            return ops.conv_dgrad(dy, wpk_t, C, 3, 1, (H, W), add=dout, add_mask=mask2, out=dx, bn_bwd=(raw_p, None, bn4, mask2),
                                  in_bnbwd=(raw, None, bn4, coef, mask), side=(draw, None))
        _, part = dgrad()
        t_d = timeit(dgrad, args.reps)
        t_p = timeit(lambda: ops.conv_dgrad(dy, wpk_t, C, 3, 1, (H, W), out=dx), args.reps)
        dxp = ops.conv_dgrad(dy, wpk_t, C, 3, 1, (H, W)).clone()
        dgrad()
        res[ws] = (out.clone(), st.clone(), dx.clone(), draw.clone(), part.clone(), dxp, t_f, t_d, t_p)
    a, b = res[False], res[True]
    names = ["fwd out", "fwd stats", "dgrad dx", "side draw", "bn partial", "plain dgrad"]
    eq = [a[i].shape == b[i].shape and torch.equal(a[i], b[i]) for i in range(6)]
    # the convolution results are bit-identical (same MFMA order per accumulator); the statistics rows are grouped by wave
    # layout, so they are compared after the fixed-order reduction over rows
    eq[1] = bool(torch.allclose(a[1].double().sum(0), b[1].double().sum(0), rtol=1e-4, atol=1e-2))
    eq[4] = bool(torch.allclose(a[4].double().sum(0), b[4].double().sum(0), rtol=1e-4, atol=1e-2))
    print("L%d C=%d %dx%d tile %s ws %s: equal %s" % (li + 1, C, H, W, tiling.conv_tile(H, W, 1, 3, 3, 9, C, split=ops.SPLIT),
                                                      tiling.ws_tile(H, W, 1, 3, 3, 9, C), dict(zip(names, eq))))
    ok_all &= all(eq)
    for j, nm in ((6, "fwd(in-affine, stats)"), (7, "dgrad(fused BN-bwd)"), (8, "dgrad(plain)")):
        print("    %-24s mfma_kernel %.3f ms %6.1f TF | ws_kernel %.3f ms %6.1f TF  (x%.2f)" % (
            nm, a[j], flops / a[j] / 1e9, b[j], flops / b[j] / 1e9, a[j] / b[j]), flush=True)
print("ALL BIT-EQUAL" if ok_all else "MISMATCH")
sys.exit(0 if ok_all else 1)
