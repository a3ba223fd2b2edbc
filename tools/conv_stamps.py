#!/usr/bin/env python3
"""Phase timeline of the non-pipelined convolution kernel from in-kernel shader-clock stamps (diagnostic build:
tools/variant.sh stamps conv_split.hip -DCONV_STAMPS; run with SPK_LIB=.../variants/libspkhip_stamps.so).
Per case: launches the convolution a few times on random data, reads the stamps of the LAST launch (first 4096 blocks) and prints
the mean cycles of each phase of a block and the block's lifetime.  Cases: l1 (3x3 32->32 at 80x300), c128_1x1, c64_1x1, s2 (3x3 stride 2)."""
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
cases = {"l1": (32, 32, 80, 300, 3, 1), "c128_1x1": (128, 128, 20, 75, 1, 1), "c64_1x1": (64, 64, 40, 150, 1, 1),
         "s2": (32, 64, 80, 300, 3, 2),
         # the pipelined kernel (a second diagnostic library: tools/variant.sh stamps_pipe conv_pipe.hip -DCONV_STAMPS)
         "c64_3x3": (64, 64, 40, 150, 3, 1), "c128_3x3": (128, 128, 20, 75, 3, 1), "c256_3x3": (256, 256, 10, 38, 3, 1)}
lib = hip.lib()
PIPE = any(n.endswith("_3x3") for n in sys.argv[1:])
reader = lib.spk_debug_stamps_pipe if PIPE else lib.spk_debug_stamps
reader.argtypes = [ctypes.c_void_p, ctypes.c_int]
NB = 4096
for name in sys.argv[1:] or [c for c in cases if not c.endswith("_3x3")]:
    Cin, Cout, H, W, k, stride = cases[name]
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    wpk = ops.pack_conv_weight(w)
    aff = (torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(4):
        if i == 3:
            torch.cuda.synchronize()
            assert reader(None, -1) == 0           # zero the stamp array: only the blocks of the launch below are read back
            e0.record()
        out, st = ops.conv_fwd(x, wpk, Cout, k, stride, in_affine=aff, stats=True)
    e1.record()
    torch.cuda.synchronize()
    buf = np.zeros((NB, 16), dtype=np.uint64)
    rc = reader(buf.ctypes.data, NB)
    assert rc == 0, rc
    s = buf.astype(np.int64)
    ok = s[:, 9] > s[:, 0]
    s = s[ok]
    d = lambda i, j: float(np.mean(s[:, i] - s[:, j]))      # noqa: E731
    if PIPE:
        print("%s: %d x %d x %d -> %d ch, 3x3 pipelined: launch %.3f ms, %d blocks stamped" % (name, H, W, Cin, Cout, e0.elapsed_time(e1), len(s)))
        print("   first barrier %6.0f   first plane staged the plain way %8.0f   barrier %6.0f   K loop (in-loop staging) %8.0f   epilogue %8.0f   "
              "block lifetime %8.0f cyc" % (d(1, 0), d(2, 1), d(3, 2), d(4, 3), d(9, 4), d(9, 0)))
        continue
    two = bool((s[:, 5] > 0).all())
    print("%s: %d x %d x %d -> %d ch, %dx%d stride %d: launch %.3f ms, %d blocks stamped" % (name, H, W, Cin, Cout, k, k, stride,
                                                                                       e0.elapsed_time(e1), len(s)))
    print("   wait for the block's turn (first barrier) %8.0f cyc" % d(1, 0))
    print("   chunk 0: stage (loads + convert + LDS)    %8.0f   barrier %6.0f   K loop %6.0f" % (d(2, 1), d(3, 2), d(4, 3)))
    if two:
        print("   chunk 1: barrier %6.0f   stage %8.0f   barrier %6.0f   K loop %6.0f" % (d(5, 4), d(6, 5), d(7, 6), d(8, 7)))
    print("   epilogue %8.0f   block lifetime %8.0f cyc; first block start -> last block end %.0f cyc" % (
        d(9, 8 if two else 4), d(9, 0), float(s[:, 9].max() - s[:, 0].min())))
