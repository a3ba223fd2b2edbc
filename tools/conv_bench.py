#!/usr/bin/env python3
"""Micro-benchmark / tile sweep of the MFMA conv kernels on the ResNet-34 layer shapes (GPU box).

  python tools/conv_bench.py            # time every conv launch shape of one training step with the current tiles
  python tools/conv_bench.py --sweep    # try candidate tiles per shape, print the best and write tile_table.json
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_kaldi_resnet_amd  # noqa: E402,F401
from pytorch_kaldi_resnet_amd import ops, tiling  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sweep", action="store_true")
ap.add_argument("--sweep-bnbwd", action="store_true", help="sweep tiles of the stride-1 data gradients in fused BatchNorm-backward mode and merge them into the table")
ap.add_argument("--no-wgrad", action="store_true", help="with --sweep: leave the weight-gradient table alone")
ap.add_argument("--wgrad-only", action="store_true", help="with --sweep: sweep only the weight-gradient tiles (section wgrad, or wgrad_split in a split operand mode)")
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--feat", type=int, default=80)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--wgrad-blocks", type=int, default=0)
ap.add_argument("--out", default=os.path.join(ROOT, "pytorch-kaldi-resnet_amd", "tile_table.json"))
args = ap.parse_args()
B, F, T = args.batch, args.feat, args.frames
if args.wgrad_blocks:
    tiling.WGRAD_TARGET_BLOCKS = args.wgrad_blocks
dev = "cuda"


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


# (name, Cin, Cout, H, W, k, stride): the distinct conv shapes of ResNet-34 at [B, 80, 300]
H1, W1 = F, T
shapes = []
dims = [(32, H1, W1)]
for c in (64, 128, 256):
    h, w = dims[-1][1], dims[-1][2]
    dims.append((c, (h - 1) // 2 + 1, (w - 1) // 2 + 1))
for i, (c, h, w) in enumerate(dims):
    shapes.append(("L%d 3x3 s1" % (i + 1), c, c, h, w, 3, 1))
    if i > 0:
        pc, ph, pw = dims[i - 1]
        shapes.append(("L%d 3x3 s2" % (i + 1), pc, c, ph, pw, 3, 2))
        shapes.append(("L%d 1x1 s2" % (i + 1), pc, c, ph, pw, 1, 2))


def conv_candidates(OH, OW, IS, ks, Cout):
    return sorted(set(tiling.conv_candidates(OH, OW, IS, ks, ks, ks * ks, Cout, per_config=6, split=ops.split_for(ks))))


# the operand mode comes from SPK_MFMA (ops.SPLIT): split-mode sweeps fill the "conv_split" section of the table
SECTION = "conv_split" if ops.SPLIT else "conv"
FORCE = tiling.FORCE_CONV_SPLIT if ops.SPLIT else tiling.FORCE_CONV
table = json.load(open(args.out)) if os.path.exists(args.out) else {}
for sec in ("conv", "conv_split", "wgrad", "wgrad_split"):
    table.setdefault(sec, {})
rows = []
if args.sweep_bnbwd:
    for name, Cin, Cout, H, W, k, s in shapes:
        if k != 3 or s != 1:
            continue
        C = Cin
        dy = torch.randn(B, H, W, C, device=dev)
        raw, act, raw_p = (torch.randn(B, H, W, C, device=dev) for _ in range(3))
        draw, dz, dx = (torch.empty(B, H, W, C, device=dev) for _ in range(3))
        bn4 = torch.randn(4, C, device=dev)
        coef = torch.randn(3, C, device=dev)
        wpk_t = ops.pack_conv_weight(torch.randn(C, C, 3, 3, device=dev) * 0.05, True)
        flops = 2.0 * B * H * W * C * C * 9
        key = (H, W, 1, 3, 3, 9, C)
        est = ops._amax_fallback(dy, (raw, act, bn4, coef)) if ops.SPLIT == 3 else None
        res = []
        for cand in conv_candidates(H, W, 1, 3, C):
            FORCE[key + (1,)] = cand
            try:
                ms = timeit(lambda: ops.conv_dgrad(dy, wpk_t, C, 3, 1, (H, W), out=dx, bn_bwd=(raw_p, None, bn4),
                                                   in_bnbwd=(raw, act, bn4, coef), side=(draw, dz), in_amax=est), args.reps)
            except RuntimeError:
                continue
            res.append((ms, cand))
        res.sort()
        FORCE[key + (1,)] = res[0][1]
        table[SECTION][",".join(map(str, key + (1,)))] = list(res[0][1])
        cur = [m for m, c in res if tuple(c) == tuple(FORCE.get(key, tiling.FORCE_CONV.get(key, ())))]
        print("%-12s bnbwd-dgrad best %s %.3f ms %.1f TF (plain-table tile: %s) | top: %s" % (
            name, res[0][1], res[0][0], flops / res[0][0] / 1e9, "%.3f ms" % cur[0] if cur else "n/a",
            " ".join("%s:%.3f" % (c, m) for m, c in res[:6])), flush=True)
    json.dump(table, open(args.out, "w"), indent=1)
    print("wrote", args.out)
    sys.exit(0)
for name, Cin, Cout, H, W, k, s in shapes:
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
    wpk = ops.pack_conv_weight(w)
    wpk_t = ops.pack_conv_weight(w, True)
    OH, OW = ops.conv_out_hw(H, W, k, s)
    dy = torch.randn(B, OH, OW, Cout, device=dev)
    dw = torch.empty_like(w)
    out = torch.empty(B, OH, OW, Cout, device=dev)
    dx = torch.empty(B, H, W, Cin, device=dev)
    flops = 2.0 * B * OH * OW * Cout * Cin * k * k
    # absmax slots of the f16x3 operand mode, computed once (the engine hands them from kernel to kernel)
    xa = ops._amax_fwd_fallback(x, None) if ops.SPLIT == 3 else None
    dya = ops.absmax_into(dy, torch.zeros(1, device=dev, dtype=torch.int32)) if ops.SPLIT == 3 else None
    key = (OH, OW, s if k == 3 else 1, k, k, k * k, Cout)      # strided 1x1 launches run as IS = 1 over a strided view
    if args.sweep and not args.wgrad_only and (k == 3 or not ops.SPLIT):
        res = []
        tab = FORCE if k == 3 else tiling.FORCE_CONV
        for cand in conv_candidates(OH, OW, s if k == 3 else 1, k, Cout):
            tab[key] = cand
            try:
                ms = timeit(lambda: ops.conv_fwd(x, wpk, Cout, k, s, stats=True, out=out, in_amax=xa), args.reps)
            except RuntimeError as e:
                continue
            res.append((ms, cand))
        res.sort()
        tab[key] = res[0][1]
        table[SECTION if k == 3 else "conv"][",".join(map(str, key))] = list(res[0][1])
        print("%-12s fwd best %s %.3f ms %.1f TF | top: %s" % (name, res[0][1], res[0][0], flops / res[0][0] / 1e9,
              " ".join("%s:%.3f" % (c, m) for m, c in res[:5])), flush=True)
        if s == 1 and k == 3:
            # the stride-1 data gradient is the same launch shape with Cin/Cout swapped (equal here)
            pass
    if args.sweep and not args.no_wgrad:
        wkey = (OH, OW, Cin, Cout, k, s)
        res = []
        OWe = OW + (OW & 1)
        for WN in ((1,) if Cout == 32 else (1, 2)):
            cands = []
            for TH in range(1, OH + 1):
                for TW in range(2, OWe + 1, 2):
                    if TH * TW > tiling.WGRAD_MAX_TILE[WN] or ((TH - 1) * s + k) * ((TW - 1) * s + k) > tiling.WGRAD_MAX_HALO:
                        continue
                    ty, tx = -(-OH // TH), -(-OW // TW)
                    cands.append((ty * tx * (TH * TW + 24.0), TH, TW))
            cands.sort()
            for _, TH, TW in cands[:16]:
                tiling.FORCE_WGRAD[wkey] = (TH, TW, WN)
                try:
                    ms = timeit(lambda: ops.conv_wgrad(x, dy, dw, k, s, dy_amax=dya, x_amax=xa), args.reps)
                except RuntimeError:
                    continue
                res.append((ms, (TH, TW, WN)))
        res.sort()
        tiling.FORCE_WGRAD[wkey] = res[0][1]
        table["wgrad_split" if (ops.SPLIT and k == 3) else "wgrad"][",".join(map(str, wkey))] = list(res[0][1])
        print("%-12s wgrad best %s %.3f ms %.1f TF | top: %s" % (name, res[0][1], res[0][0], flops / res[0][0] / 1e9,
              " ".join("%s:%.3f" % (c, m) for m, c in res[:5])), flush=True)
    t_fwd = timeit(lambda: ops.conv_fwd(x, wpk, Cout, k, s, stats=True, out=out, in_amax=xa), args.reps)
    t_dg = timeit(lambda: ops.conv_dgrad(dy, wpk_t, Cin, k, s, (H, W), out=dx, in_amax=dya), args.reps)
    t_wg = timeit(lambda: ops.conv_wgrad(x, dy, dw, k, s, dy_amax=dya, x_amax=xa), args.reps)
    rows.append((name, flops, t_fwd, t_dg, t_wg))
    print("%-12s %6.1f GF  fwd %.3f ms %5.1f TF  dgrad %.3f ms %5.1f TF  wgrad %.3f ms %5.1f TF   tiles %s / %s" % (
        name, flops / 1e9, t_fwd, flops / t_fwd / 1e9, t_dg, flops / t_dg / 1e9, t_wg, flops / t_wg / 1e9,
        tiling.conv_tile(*key, split=ops.split_for(k)), tiling.wgrad_tile(OH, OW, Cin, Cout, k, s)), flush=True)
if args.sweep:
    json.dump(table, open(args.out, "w"), indent=1)
    print("wrote", args.out)
