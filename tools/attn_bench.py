"""Isolated timing of the attention kernels at the encoder shape of the benchmark (B=64, H=8, L=128, bf16, dropout 0.1; C5: `attn_bench.py 64 16 512`) -- dev tool.
Event timing of Python-issued launches: anything much under ~40 us is host-bound here, use the rocprofv3 tables for those."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openp5_amd._lib import hip_backend
from openp5_amd.model import relative_position_bucket_lut
be = hip_backend(); lib = be.lib
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
B, H, L = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 8, 128)
inner = H * 64
bf = torch.bfloat16
qkv = (0.5 * torch.randn(B * L, 3 * inner, device="cuda")).to(bf)
dqkv = torch.zeros_like(qkv)
O = torch.zeros(B * L, inner, device="cuda", dtype=bf); dO = torch.randn(B * L, inner, device="cuda").to(bf)
lse = torch.zeros(B * H * L, device="cuda"); Dv = torch.zeros(B * H * L, device="cuda")
table = (0.5 * torch.randn(32, H)).cuda(); dtab = torch.zeros(32, H, device="cuda"); dscr = torch.zeros(B * ((L + 63) // 64), 32 * H, device="cuda")
lut = relative_position_bucket_lut(512, True, 32, 128).cuda()
kmask = torch.ones(B, L, dtype=torch.long, device="cuda"); kmask[:, L - 8:] = 0
rng = torch.tensor([1234, 7], dtype=torch.int32, device="cuda")
s = be.stream_ptr()
f = ctypes.c_float
fwd = lambda: lib.p5_op_attn_fwd(1, P(qkv), P(qkv[:, inner:]), P(qkv[:, 2 * inner:]), P(O), P(lse), P(table), P(lut), 512, P(kmask), B, H, L, L,
                                 3 * inner, 3 * inner, 3 * inner, inner, 0, P(rng), 11, f(0.1), s)
bwd = lambda: lib.p5_op_attn_bwd(1, P(qkv), P(qkv[:, inner:]), P(qkv[:, 2 * inner:]), P(O), P(dO), P(lse), P(Dv), P(dqkv), P(dqkv[:, inner:]),
                                 P(dqkv[:, 2 * inner:]), P(table), P(dtab), P(dscr), 32, P(lut), 512, P(kmask), B, H, L, L, 3 * inner, 3 * inner, 3 * inner,
                                 inner, 3 * inner, 3 * inner, 3 * inner, 0, P(rng), 11, f(0.1), s)
def timeit(name, call, iters=40):
    for _ in range(3): assert call() == 0, lib.p5_last_error()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) / iters * 1e3:8.1f} us", flush=True)
flops = 4.0 * B * H * L * L * 64
if L > 128:
    for head in (1, 0):
        lib.p5_set_option(b"attn_fwd_head", head); lib.p5_set_option(b"attn_bwd_head", head)
        timeit(f"fwd, head-resident={head}", fwd, iters=10)
        timeit(f"bwd (dq + dkv), head-resident={head}", bwd, iters=10)
    lib.p5_set_option(b"attn_fwd_head", 1); lib.p5_set_option(b"attn_bwd_head", 1)
    print(f"(forward {flops / 1e9:.1f} GFLOP, backward {2.5 * flops / 1e9:.1f} GFLOP per call)")
else:
    timeit("fwd", fwd)
    lib.p5_set_option(b"attn_fused", 0); timeit("bwd split (dq + dkv)", bwd)
    lib.p5_set_option(b"attn_fused", 1)
    timeit("bwd fused", bwd)
