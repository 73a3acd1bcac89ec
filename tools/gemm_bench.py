"""Per-shape timing of the GEMM kernel on the GPU (dev tool): the shapes one T5-small training step launches."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openp5_amd._lib import hip_backend
be = hip_backend(); lib = be.lib
P = lambda t: ctypes.c_void_p(t.data_ptr())
def run(M, N, K, aks, bks, epi=0, c_f32=0, iters=30, dtype=1):
    tt = torch.bfloat16 if dtype else torch.float32
    A = torch.randn((K, M) if aks else (M, K), device='cuda').to(tt)
    B = torch.randn((K, N) if bks else (N, K), device='cuda').to(tt)
    C = torch.zeros(M, N, device='cuda', dtype=torch.float32 if c_f32 else tt)
    aux = torch.randn(M, N, device='cuda').to(tt) if epi in (2, 3) else None
    call = lambda: lib.p5_op_gemm(dtype, P(A), P(B), P(C), P(aux) if aux is not None else None, M, N, K, A.shape[1], B.shape[1], N, N, aks, bks, epi, c_f32, 0 if epi == 4 else 1, 1.0, None, 0, 0.0, be.stream_ptr())
    for _ in range(3): assert call() == 0, lib.p5_last_error()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"M={M:6d} N={N:6d} K={K:6d} aks={aks} bks={bks} epi={epi} f32out={c_f32}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s")
Mt = 8192
for (n, k) in [(1536, 512), (512, 512), (2048, 512), (512, 2048), (1024, 512)]:
    run(Mt, n, k, 0, 0)
run(Mt, 512, 2048, 0, 0, epi=2)
run(Mt, 2048, 512, 0, 0, epi=1)
for (n, k) in [(512, 1536), (512, 512), (512, 2048), (2048, 512), (512, 1024)]:
    run(Mt, n, k, 0, 1)          # dgrad: N = in-features, K = out-features
for (n, k) in [(1536, 512), (512, 512), (2048, 512), (512, 2048), (1024, 512)]:
    run(n, k, Mt, 1, 1, epi=4, c_f32=1)   # wgrad: M=out, N=in, K=tokens
run(512, 32100, 512, 0, 0, c_f32=1)
run(4096, 4096, 4096, 0, 0)
run(8192, 2048, 4096, 0, 0)
run(512, 1536, 512, 0, 0); run(512, 512, 512, 0, 0); run(512, 2048, 512, 0, 0)
run(Mt, 2048, 512, 0, 1, epi=3)      # dgrad through wo with the ReLU/dropout mask epilogue
run(Mt, 512, 2048, 0, 1, epi=0)
