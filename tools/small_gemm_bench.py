"""Small-problem GEMMs (decoder / decode step): v1 64x64 kernel vs eight-slot ring (dev tool)."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openp5_amd._lib import hip_backend
be = hip_backend(); lib = be.lib
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
def run(M, N, K, aks, bks, epi, c_f32, ring, iters=50):
    A = torch.randn((K, M) if aks else (M, K), device='cuda').bfloat16()
    B = torch.randn((K, N) if bks else (N, K), device='cuda').bfloat16()
    C = torch.zeros(M, N, device='cuda', dtype=torch.float32 if c_f32 else torch.bfloat16)
    aux = torch.randn(M, N, device='cuda').bfloat16() if epi in (2, 3) else None
    lib.p5_set_option(b"gemm_small_ring", ring)
    call = lambda: lib.p5_op_gemm(1, P(A), P(B), P(C), P(aux), M, N, K, A.shape[1], B.shape[1], N, N, aks, bks, epi, c_f32, 0 if epi == 4 else 1, 1.0, None, 0, 0.0, be.stream_ptr())
    for _ in range(3): assert call() == 0, lib.p5_last_error()
    torch.cuda.synchronize()
    ref = (A.float().t() if aks else A.float()) @ (B.float() if bks else B.float().t())
    C.zero_(); call(); torch.cuda.synchronize()
    if epi == 2: ref = ref + aux.float()
    err = ((C.float() - ref).abs().max() / ref.abs().max()).item() if epi in (0, 2, 4) else 0.0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    return f"{e0.elapsed_time(e1) / iters * 1e3:5.1f}us(err {err:.0e})"
print("shape: [v1 | ring8]")
for name, a in [("fwd 512x1536x512", (512, 1536, 512, 0, 0, 0, 0)), ("fwd 512x512x512+res", (512, 512, 512, 0, 0, 2, 0)), ("fwd 512x2048x512 relu", (512, 2048, 512, 0, 0, 1, 0)),
                ("fwd 512x512x2048+res", (512, 512, 2048, 0, 0, 2, 0)), ("dgrad 512x512x1536", (512, 512, 1536, 0, 1, 0, 0)), ("dgrad 512x2048x512 mask", (512, 2048, 512, 0, 1, 3, 0)),
                ("dgrad 512x512x2048", (512, 512, 2048, 0, 1, 0, 0)), ("wgrad 512x512 K512", (512, 512, 512, 1, 1, 4, 1)), ("wgrad 2048x512 K512", (2048, 512, 512, 1, 1, 4, 1)),
                ("decode 200x1536x512", (200, 1536, 512, 0, 0, 0, 0)), ("decode 200x512x512+res", (200, 512, 512, 0, 0, 2, 0)), ("decode 200x512x2048+res", (200, 512, 2048, 0, 0, 2, 0)),
                ("decode lm-head 200x32100x512", (200, 32100, 512, 0, 0, 0, 1))]:
    print(f"  {name:30s}", run(*a, 0), "|", run(*a, 1))
