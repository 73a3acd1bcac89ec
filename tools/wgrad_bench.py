"""wgrad / dgrad shapes under forced tile sizes (dev tool)."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openp5_amd._lib import hip_backend
be = hip_backend(); lib = be.lib
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
def run(M, N, K, aks, bks, epi, c_f32, tile, splitk=0, iters=30, ksdma=1):
    A = torch.randn((K, M) if aks else (M, K), device='cuda').bfloat16()
    B = torch.randn((K, N) if bks else (N, K), device='cuda').bfloat16()
    C = torch.zeros(M, N, device='cuda', dtype=torch.float32 if c_f32 else torch.bfloat16)
    lib.p5_set_option(b"gemm_tile", tile); lib.p5_set_option(b"gemm_ksdma", ksdma)
    call = lambda: lib.p5_op_gemm(1, P(A), P(B), P(C), None, M, N, K, A.shape[1], B.shape[1], N, N, aks, bks, epi, c_f32, splitk, 1.0, None, 0, 0.0, be.stream_ptr())
    for _ in range(3): assert call() == 0, lib.p5_last_error()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    return f"{us:6.1f}us/{2.0*M*N*K/us/1e6:4.0f}TF"
Mt = 8192
def check(M, N, K, aks, bks, tile):
    A = torch.randn((K, M) if aks else (M, K), device='cuda').bfloat16()
    B = torch.randn((K, N) if bks else (N, K), device='cuda').bfloat16()
    C = torch.zeros(M, N, device='cuda', dtype=torch.float32)
    lib.p5_set_option(b"gemm_tile", tile); lib.p5_set_option(b"gemm_ksdma", 1)
    assert lib.p5_op_gemm(1, P(A), P(B), P(C), None, M, N, K, A.shape[1], B.shape[1], N, N, aks, bks, 4, 1, 0, 1.0, None, 0, 0.0, be.stream_ptr()) == 0
    ref = (A.float().t() if aks else A.float()) @ (B.float() if bks else B.float().t())
    return ((C - ref).abs().max() / ref.abs().max()).item()
def run2(m, n, v2, sk):
    lib.p5_set_option(b"gemm_v2", v2)
    r = run(m, n, Mt, 1, 1, 4, 1, 128, sk)
    lib.p5_set_option(b"gemm_v2", 0)
    return r
def cmp(fn):
    out = []
    for rect in (0, 1):
        lib.p5_set_option(b"gemm_xcd_rect", rect)
        out.append(fn())
    return " | ".join(out)
print("wgrad, launcher heuristics: [contiguous runs | rectangles]")
for (m, n) in [(1536, 512), (512, 512), (2048, 512), (512, 2048), (1024, 512)]:
    print(f"  {m}x{n}:", cmp(lambda: run(m, n, Mt, 1, 1, 4, 1, 0, 0)))
print("fwd: [runs | rectangles]")
for (n, k) in [(2048, 512), (1536, 512), (512, 512), (512, 2048), (1024, 512)]:
    print(f"  N={n} K={k}:", cmp(lambda: run(Mt, n, k, 0, 0, 0, 0, 0, 1)))
print("dgrad: [runs | rectangles]")
for (n, k) in [(512, 1536), (512, 512), (512, 2048), (2048, 512)]:
    print(f"  N={n} K={k}:", cmp(lambda: run(Mt, n, k, 0, 1, 0, 0, 0, 1)))
sys.exit(0)
lib.p5_set_option(b"gemm_v2", 4)
print("rel err pipelined wgrad:", [f"{check(*a):.1e}" for a in [(512, 2048, 8192, 1, 1, 128), (520, 264, 1024, 1, 1, 128)]])
lib.p5_set_option(b"gemm_v2", 0)
print("wgrad t128 pipelined [v1 t64 auto | ring3 sk4 | ring4 sk2 | ring4 sk4 | ring4 sk8 | ring4 sk16]")
for (m, n) in [(1536, 512), (512, 512), (2048, 512), (512, 2048), (1024, 512)]:
    print(f"  {m}x{n}:", run(m, n, Mt, 1, 1, 4, 1, 64), "|", run2(m, n, 3, 4), "|", " ".join(run2(m, n, 4, sk) for sk in (2, 4, 8, 16)))
