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

# fp16 denormal operands: does the matrix instruction keep them?  a = 2^-20 (subnormal in fp16) x b = 1 over K = 16
A = torch.zeros(32, 16); A[:, 0] = 2.0 ** -20
B = torch.zeros(16, 32); B[0, :] = 1.0
C = torch.zeros(32, 32, device="cuda")
lib.split_probe(A.cuda().data_ptr(), B.cuda().data_ptr(), C.data_ptr(), 16, 23, 1, torch.cuda.current_stream().cuda_stream, 1.0, 1.0)
torch.cuda.synchronize()
print("fp16 subnormal operand 2^-20 x 1 -> %.6e (exact %.6e): %s" % (float(C[0, 0]), 2.0 ** -20, "kept" if float(C[0, 0]) != 0 else "FLUSHED"))
