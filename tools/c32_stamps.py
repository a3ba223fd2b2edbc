#!/usr/bin/env python3
"""Where wave 0 of a block of the streaming 32-channel 3x3 kernel spends its time (diagnostic build:
tools/variant.sh c32stamps conv3x3_c32_stream.hip -DC32_STAMPS; run with SPK_LIB=.../variants/libspkhip_c32stamps.so).
Prints the share of each phase of a tile in the block's lifetime, averaged over the first 1024 blocks of the last launch."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_kaldi_resnet_amd  # noqa: E402,F401
from pytorch_kaldi_resnet_amd import hip, ops  # noqa: E402

B = int(os.environ.get("B", "256"))
H, W = 80, 300
lib = hip.lib()
reader = getattr(lib, "spk_debug_stamps_c32", None)        # absent from a regular build: launch times only
if reader is not None:
    reader.argtypes = [ctypes.c_void_p, ctypes.c_int]
x = torch.randn(B, H, W, 32, device="cuda")
w = torch.randn(32, 32, 3, 3, device="cuda") * 0.05
wpk = ops.pack_conv_weight(w)
aff = (torch.rand(32, device="cuda") + 0.5, torch.randn(32, device="cuda") * 0.1)
names = ["tile origin + arrive", "barrier 1 (others still read the tile)", "convert + LDS write (waits for the loads)", "barrier 2",
         "issue loads of tile + 2G", "K loop (LDS reads + MFMA)", "epilogue (stores, statistics)"]
for label, a in (("fused BN + ReLU input", aff), ("plain input", None)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    slot = ops._amax_fwd_fallback(x, a)              # once: without it every call below would add an absmax pass over x
    buf_out = torch.empty(B, H, W, 32, device="cuda")
    for i in range(3):
        out, st = ops.conv_fwd(x, wpk, 32, 3, 1, in_affine=a, stats=True, in_amax=slot, out=buf_out)
    torch.cuda.synchronize()
    assert reader is None or reader(None, -1) == 0
    NL = 20                                          # replayed from a captured graph: no host launch latency between the launches
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(NL):
            out, st = ops.conv_fwd(x, wpk, 32, 3, 1, in_affine=a, stats=True, in_amax=slot, out=buf_out)
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / NL
    if reader is None:
        print("%s: %.3f ms per launch (kernel alone)" % (label, ms))
        continue
    nb = min(1024, ops.STREAM_C32_BLOCKS)
    buf = np.zeros((nb, 8), dtype=np.uint64)
    assert reader(buf.ctypes.data, nb) == 0
    s = buf.astype(np.float64)
    life = s[:, 7].mean()
    print("%s: %.3f ms per launch (kernel alone), %d blocks; block lifetime %.0f ticks of s_memtime" % (label, ms, nb, life))
    for i, n in enumerate(names):
        print("   %-48s %5.1f %%" % (n, 100.0 * s[:, i].mean() / life))
