"""v1 vs hand-pipelined main loop on the forward GEMM shapes (dev tool)."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openp5_amd._lib import hip_backend
be = hip_backend(); lib = be.lib
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
def run(M, N, K, epi=0, iters=40):
    A = torch.randn(M, K, device='cuda').bfloat16(); B = torch.randn(N, K, device='cuda').bfloat16()
    C = torch.zeros(M, N, device='cuda', dtype=torch.bfloat16)
    aux = torch.randn(M, N, device='cuda').bfloat16() if epi in (2, 3) else None
    ref = (A.float() @ B.float().t())
    if epi == 1: ref = torch.relu(ref)
    if epi == 2: ref = ref + aux.float()
    out = []
    for tile, v2 in [(0, 0), (128, 0), (256, 0)]:
        lib.p5_set_option(b"gemm_tile", tile); lib.p5_set_option(b"gemm_v2", v2)
        call = lambda: lib.p5_op_gemm(1, P(A), P(B), P(C), P(aux), M, N, K, K, K, N, N, 0, 0, epi, 0, 1, 1.0, None, 0, 0.0, be.stream_ptr())
        for _ in range(3): assert call() == 0, lib.p5_last_error()
        torch.cuda.synchronize()
        err = (C.float() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): call()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        out.append(f"[tile {tile:3d} v2={v2}] {us:7.1f} us {2.0*M*N*K/us/1e6:6.0f} TF (err {err:.1e})")
    print(f"M={M} N={N} K={K} epi={epi}: " + "  ".join(out))
for (n, k, epi) in [(2048, 512, 0), (2048, 512, 1), (1536, 512, 0), (512, 2048, 2), (1024, 512, 0), (2048, 2048, 0)]:
    run(8192, n, k, epi)
run(4096, 4096, 4096); run(8192, 8192, 8192, iters=10)
