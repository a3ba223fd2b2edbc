#!/usr/bin/env python3
"""spk_gemm_f32 on the six GEMM shapes of one training step (fc1 and AAM cosine, forward and both gradients):
correctness against torch fp64 on the host and achieved fp32 TFLOP/s.  GPU box."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_kaldi_resnet_amd  # noqa: E402,F401
from pytorch_kaldi_resnet_amd import ops  # noqa: E402

B, D, F, S = 256, 256, 5120, int(sys.argv[1]) if len(sys.argv) > 1 else 1211
torch.manual_seed(0)
x, w1 = torch.randn(B, F, device="cuda"), torch.randn(D, F, device="cuda") * 0.02
h, wn = torch.randn(B, D, device="cuda"), torch.randn(S, D, device="cuda") * 0.05
dy, dcos = torch.randn(B, D, device="cuda"), torch.randn(B, S, device="cuda")
cases = [
    ("fc1 fwd      x @ W^T", lambda: ops.gemm(x, w1, B, D, F, F, 1, 1, F), lambda: x.double() @ w1.double().T),
    ("cosine       h @ Wn^T", lambda: ops.gemm(h, wn, B, S, D, D, 1, 1, D), lambda: h.double() @ wn.double().T),
    ("d hn         dcos @ Wn", lambda: ops.gemm(dcos, wn, B, D, S, S, 1, D, 1), lambda: dcos.double() @ wn.double()),
    ("d Wn         dcos^T @ h", lambda: ops.gemm(dcos, h, S, D, B, 1, S, D, 1), lambda: dcos.double().T @ h.double()),
    ("fc1 dx       dy @ W", lambda: ops.gemm(dy, w1, B, F, D, D, 1, F, 1), lambda: dy.double() @ w1.double()),
    ("fc1 dW       dy^T @ x", lambda: ops.gemm(dy, x, D, F, B, 1, D, F, 1), lambda: dy.double().T @ x.double()),
]
tot = 0.0
for name, fn, ref in cases:
    out = fn()
    r = ref()
    err = float((out.double() - r).abs().max() / r.abs().max())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    M, N = out.shape
    K = {0: F, 1: D, 2: S, 3: B, 4: D, 5: B}[cases.index((name, fn, ref))]
    tot += ms
    print("%-26s %4dx%4dx%4d  splitk %2d  %.4f ms  %6.2f TFLOP/s  max rel err %.2e" % (
        name, M, N, K, pytorch_kaldi_resnet_amd.hip.lib().spk_gemm_splitk(M, N, K), ms, 2.0 * M * N * K / ms / 1e9, err))
print("sum %.3f ms per step" % tot)
