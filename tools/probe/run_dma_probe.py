import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "dma_probe.so"))
n = 6
src = torch.randn(5000, 4, device="cuda")
idx = torch.randint(0, 5000, (n * 256,), device="cuda", dtype=torch.int32)
out = torch.zeros(n * 256, 4, device="cuda")
lib.dma_probe.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p]
rc = lib.dma_probe(src.data_ptr(), idx.data_ptr(), out.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
ref = src[idx.long()]
print("rc", rc, "match", bool(torch.equal(out, ref)), float((out - ref).abs().max()))
