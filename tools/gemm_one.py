"""Launches one bf16 GEMM shape a few times -- target for rocprofv3 --pmc.  usage: gemm_one.py M N K [aks bks]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openp5_amd._lib import hip_backend
be = hip_backend(); lib = be.lib
P = lambda t: ctypes.c_void_p(t.data_ptr())
M, N, K = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else (8192, 2048, 512)
aks, bks = ([int(x) for x in sys.argv[4:6]] if len(sys.argv) > 5 else (0, 0))
A = torch.randn((K, M) if aks else (M, K), device='cuda').to(torch.bfloat16); B = torch.randn((K, N) if bks else (N, K), device='cuda').to(torch.bfloat16)
epi, cf = (4, 1) if aks else (0, 0)
C = torch.zeros(M, N, device='cuda', dtype=torch.float32 if cf else torch.bfloat16)
for _ in range(5):
    assert lib.p5_op_gemm(1, P(A), P(B), P(C), None, M, N, K, A.shape[1], B.shape[1], N, 0, aks, bks, epi, cf, 0 if epi == 4 else 1, 1.0, None, 0, 0.0, be.stream_ptr()) == 0
torch.cuda.synchronize()
