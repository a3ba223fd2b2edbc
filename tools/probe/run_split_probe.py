import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "split_probe.so"))
lib.split_probe.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_float, ctypes.c_float]
torch.manual_seed(0)
for name, K, mk in (("relu-act x N(0,.03) weights, K=4608", 4608, lambda K: (torch.relu(torch.randn(32, K) * 1.3 + 0.2), torch.randn(K, 32) * 0.03)),
                    ("N(0,1) x N(0,1), K=1152", 1152, lambda K: (torch.randn(32, K), torch.randn(K, 32))),
                    ("tiny gradients x weights: N(0,1e-6) x N(0,.03), K=2304", 2304, lambda K: (torch.randn(32, K) * 1e-6, torch.randn(K, 32) * 0.03)),
                    ("wide dynamic range, K=2304", 2304, lambda K: (torch.randn(32, K) * torch.exp(torch.randn(32, K) * 3), torch.randn(K, 32) * torch.exp(torch.randn(K, 32) * 3)))):
    A, B = mk(K)
    ref = A.double() @ B.double()
    scale = ref.abs().mean()   # absolute error is reported relative to the mean magnitude of the outputs
    cpu32 = (A @ B).double()
    print(name)
    print("  torch CPU fp32 matmul       max %.3e rms %.3e" % (float((cpu32 - ref).abs().max() / scale), float((cpu32 - ref).pow(2).mean().sqrt() / scale)))
    Ad, Bd = A.cuda().contiguous(), B.cuda().contiguous()
    import math
    # power-of-two scales that put the largest magnitude of each operand at ~2^9 (what the kernels do from an absmax)
    sa = 2.0 ** (9 - math.ceil(math.log2(float(A.abs().max()))))
    sb = 2.0 ** (9 - math.ceil(math.log2(float(B.abs().max()))))
    for v, lab in ((0, "native fp32 MFMA 32x32x2"), (9, "bf16 split, 9 terms"), (6, "bf16 split, 6 terms"), (3, "bf16 split, 3 terms"),
                   (23, "fp16 2-term split, 3 products"), (24, "fp16 2-term split, 4 products")):
        C = torch.zeros(32, 32, device="cuda")
        rc = lib.split_probe(Ad.data_ptr(), Bd.data_ptr(), C.data_ptr(), K, v, 1, torch.cuda.current_stream().cuda_stream, sa, sb)
        torch.cuda.synchronize()
        err = (C.cpu().double() - ref).abs()
        print("  %-27s max %.3e rms %.3e (rc %d)" % (lab, float(err.max() / scale), float(err.pow(2).mean().sqrt() / scale), rc))

# fp16 SUBNORMAL operands: does v_mfma_f32_32x32x16_f16 keep them or flush them to zero?  Operands are built from bit patterns
# inside the kernel (the round-2 version of this check passed pointers of temporaries that had already been freed - it printed
# 1.0 for 2^-20 x 1 - and is replaced by this one).  fp16: min normal 2^-14 = 0x0400, subnormals 2^-24 (0x0001) .. 2^-15 (0x0200).
import numpy as np
lib.subnormal_probe.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
out = torch.zeros(1, device="cuda")
ONE = 0x3C00
kept_all = True
for name, bits in (("2^-24 (smallest subnormal)", 0x0001), ("2^-20 (subnormal)", 0x0010), ("2^-15 (largest power-of-two subnormal)", 0x0200),
                   ("2^-14 (smallest normal)", 0x0400), ("1.5 * 2^-16 (subnormal, two mantissa bits)", 0x0180)):
    exact = float(np.array([bits], dtype=np.uint16).view(np.float16)[0])
    res = []
    for a, b in ((bits, ONE), (ONE, bits)):
        out.zero_()
        lib.subnormal_probe(a, b, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        res.append(float(out[0]))
    ok = all(r == exact for r in res)
    if bits < 0x0400:
        kept_all = kept_all and ok
    print("fp16 operand %-44s as A: %.6e  as B: %.6e  (exact %.6e): %s" % (name, res[0], res[1], exact, "kept" if ok else ("FLUSHED" if all(r == 0 for r in res) else "ALTERED")))
print("=> v_mfma_f32_32x32x16_f16 %s fp16 subnormal operands" % ("KEEPS" if kept_all else "does NOT keep"))
