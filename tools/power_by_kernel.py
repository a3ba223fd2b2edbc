#!/usr/bin/env python3
"""Board power and shader clock while ONE kernel class runs back to back for a few seconds each (graph replays of 40 launches),
sampled with rocm-smi from a thread: which kernels of the training step sit at the 1400 W cap, and what clock the chip holds there.
usage: python3 tools/power_by_kernel.py [seconds per case]"""
import os
import re
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_kaldi_resnet_amd  # noqa: E402,F401
from pytorch_kaldi_resnet_amd import ops  # noqa: E402

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
B = 256
dev = torch.device("cuda", 0)
samples = []
stop = [False]


def sampler():
    while not stop[0]:
        try:
            t = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
        except Exception:
            continue
        p = re.search(r"Power \(W\): ([0-9.]+)", t)
        s = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", t)
        if p and s:
            samples.append((time.perf_counter(), float(p.group(1)), int(s.group(1))))


def conv_case(C, H, W, k):
    x = torch.randn(B, H, W, C, device=dev).relu_()
    w = torch.randn(C, C, k, k, device=dev) * (2.0 / (C * k * k)) ** 0.5
    wpk = ops.pack_conv_weight(w)
    aff = (torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1)
    slot = ops._amax_fwd_fallback(x, aff)            # once: the launches below carry their operand-scale slot like the engine's do
    out = torch.empty(B, H, W, C, device=dev)
    return lambda: ops.conv_fwd(x, wpk, C, k, 1, in_affine=aff, stats=True, in_amax=slot, out=out)


def wgrad_case(C, H, W):
    x = torch.randn(B, H, W, C, device=dev).relu_()
    dy = torch.randn(B, H, W, C, device=dev)
    dw = torch.zeros(C, C, 3, 3, device=dev)
    xs, ds = ops._amax_fwd_fallback(x, None), ops.absmax_into(dy, torch.zeros(1, device=dev, dtype=torch.int32))
    return lambda: ops.conv_wgrad(x, dy, dw, 3, 1, dy_amax=ds, x_amax=xs)


def bn_apply_case(C, H, W):
    raw = torch.randn(B, H, W, C, device=dev)
    res = torch.randn(B, H, W, C, device=dev)
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    out = torch.empty_like(raw)
    return lambda: ops.bn_apply(raw, sc, sh, res=res, relu=True, out=out)


cases = [("idle", None),
         ("3x3 forward, 128 channels 20x75 (pipelined, 16x16x32)", conv_case(128, 20, 75, 3)),
         ("3x3 forward, 256 channels 10x38 (pipelined, 16x16x32)", conv_case(256, 10, 38, 3)),
         ("3x3 forward, 32 channels 80x300 (streaming kernel)", conv_case(32, 80, 300, 3)),
         ("3x3 weight gradient, 128 channels 20x75", wgrad_case(128, 20, 75)),
         ("BatchNorm apply + residual + ReLU, 32 channels 80x300 (2 reads + 1 write of 786 MB)", bn_apply_case(32, 80, 300))]
th = threading.Thread(target=sampler, daemon=True)
th.start()
marks = []
for name, fn in cases:
    t0 = time.perf_counter()
    n, ms = 0, 0.0
    if fn is None:
        time.sleep(SECS)
    else:
        for _ in range(2):
            fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(40):
                fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.perf_counter() - t0 < SECS:
            for _ in range(5):
                g.replay()
            n += 200
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / max(n, 1)
    marks.append((name, t0, time.perf_counter(), ms))
stop[0] = True
th.join(timeout=15)
for name, t0, t1, ms in marks:
    # the board's power reading is an average over the last ~second: drop the first 1.5 s of a case
    sel = [(p, s) for t, p, s in samples if t0 + 1.5 <= t <= t1]
    if not sel:
        print("%-90s no samples" % name)
        continue
    pw = sorted(p for p, _ in sel)
    sc = sorted(s for _, s in sel)
    print("%-90s %s  power median %4.0f W (max %4.0f)  sclk median %4d MHz (min %4d)  %d samples" % (
        name, ("%.3f ms per launch" % ms) if ms else "                   ", pw[len(pw) // 2], pw[-1], sc[len(sc) // 2], sc[0], len(sel)))
