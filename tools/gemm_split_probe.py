"""fp32 GEMM: exact fp32 MFMAs vs products on the f16 matrix cores (two-term split, p5_gemm.h) -- time and error on the verification pass's shapes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openp5_amd._lib import hip_backend
from tests.cases import P
be = hip_backend()
dev = be.device
for (M, N, K) in [(2560, 1536, 512), (2560, 512, 512), (2560, 2048, 512), (2560, 512, 2048), (2560, 6144, 512), (960, 1536, 512), (960, 512, 2048), (960, 32100, 512)]:
    g = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=g); B = torch.randn(N, K, generator=g) * 0.05
    ref = A.double() @ B.double().t()
    Ad, Bd, Cd = A.to(dev), B.to(dev), torch.zeros(M, N, device=dev)
    row = f"{M}x{N}x{K}:"
    for dt in (0, 2):
        def run():
            be.check(be.lib.p5_op_gemm(dt, P(Ad), P(Bd), P(Cd), None, M, N, K, K, K, N, N, 0, 0, 0, 1, 1, 1.0, None, 0, 0.0, be.stream_ptr()), "gemm")
        for _ in range(3): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): run()
        torch.cuda.synchronize(); dt_us = (time.perf_counter() - t0) / 20 * 1e6
        err = ((Cd.cpu().double() - ref).abs().max() / ref.abs().max()).item()
        row += f"  {'split' if dt else 'fp32 '} {dt_us:7.1f} us {2.0*M*N*K/dt_us/1e6:6.1f} TF/s err {err:.1e}"
    print(row)
