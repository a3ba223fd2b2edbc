import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "split_probe.so"))
lib.split_probe.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
torch.manual_seed(0)
for name, K, mk in (("relu-act x N(0,.03) weights, K=4608", 4608, lambda K: (torch.relu(torch.randn(32, K) * 1.3 + 0.2), torch.randn(K, 32) * 0.03)),
                    ("N(0,1) x N(0,1), K=1152", 1152, lambda K: (torch.randn(32, K), torch.randn(K, 32))),
                    ("wide dynamic range, K=2304", 2304, lambda K: (torch.randn(32, K) * torch.exp(torch.randn(32, K) * 3), torch.randn(K, 32) * torch.exp(torch.randn(K, 32) * 3)))):
    A, B = mk(K)
    ref = A.double() @ B.double()
    scale = ref.abs().mean()   # absolute error is reported relative to the mean magnitude of the outputs
    cpu32 = (A @ B).double()
    print(name)
    print("  torch CPU fp32 matmul       max %.3e rms %.3e" % (float((cpu32 - ref).abs().max() / scale), float((cpu32 - ref).pow(2).mean().sqrt() / scale)))
    Ad, Bd = A.cuda().contiguous(), B.cuda().contiguous()
    for v, lab in ((0, "native fp32 MFMA 32x32x2"), (9, "bf16 split, 9 terms"), (6, "bf16 split, 6 terms"), (3, "bf16 split, 3 terms")):
        C = torch.zeros(32, 32, device="cuda")
        rc = lib.split_probe(Ad.data_ptr(), Bd.data_ptr(), C.data_ptr(), K, v, 1, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        err = (C.cpu().double() - ref).abs()
        print("  %-27s max %.3e rms %.3e (rc %d)" % (lab, float(err.max() / scale), float(err.pow(2).mean().sqrt() / scale), rc))
