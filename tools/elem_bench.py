"""Isolated timing of the memory-bound kernels at the encoder shape (dev tool)."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openp5_amd._lib import hip_backend
be = hip_backend(); lib = be.lib
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
def timeit(name, call, nbytes, iters=50):
    for _ in range(3): assert call() == 0, lib.p5_last_error()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"{name:32s} {us:8.1f} us  {nbytes/us/1e6:6.2f} TB/s")
for rows, d in [(8192, 512), (512, 512), (8192, 768)]:
    bf = torch.bfloat16
    x = torch.randn(rows, d, device="cuda").to(bf); dy = torch.randn(rows, d, device="cuda").to(bf)
    w = torch.ones(d, device="cuda"); rstd = torch.rand(rows, device="cuda") + 0.5
    dres_in = torch.randn(rows, d, device="cuda"); dres_out = torch.empty_like(dres_in)
    dy_next = torch.empty_like(x); dw = torch.zeros(d, device="cuda"); y = torch.empty_like(x)
    s = be.stream_ptr()
    timeit(f"rmsnorm_fwd {rows}x{d}", lambda: lib.p5_op_rmsnorm_fwd(1, P(y), P(rstd), P(x), P(w), rows, d, ctypes.c_float(1e-6), s), rows * d * 4)
    timeit(f"rmsnorm_bwd {rows}x{d}", lambda: lib.p5_op_rmsnorm_bwd(1, P(dres_out), P(dy_next), P(dw), P(dy), P(x), P(w), P(rstd), P(dres_in), rows, d, None, s),
           rows * d * (2 + 2 + 4 + 4 + 2))
    scratch = torch.zeros(1024 * d, device="cuda")
    timeit(f"rmsnorm_bwd partials {rows}x{d}", lambda: lib.p5_op_rmsnorm_bwd(1, P(dres_out), P(dy_next), P(dw), P(dy), P(x), P(w), P(rstd), P(dres_in), rows, d, P(scratch), s),
           rows * d * (2 + 2 + 4 + 4 + 2))
