"""Host side of tools/probe/mem_probe.hip: streaming read / copy rates and fp32 global-atomic rates on one MI355X."""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libmem_probe.so"))
lib.mem_stream.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
lib.mem_atomic.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
out = torch.zeros(16, dtype=torch.int32, device="cuda")
def timed(f, reps=8):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[2:]); return ts[len(ts) // 2]
for mb in (64, 242, 2048):
    src = torch.empty(mb * 1024 * 1024 // 4, dtype=torch.int32, device="cuda").fill_(3)
    dst = torch.empty_like(src)
    for wgs in (1024, 4096):
        for unroll in (4, 8):
            us = timed(lambda: lib.mem_stream(out.data_ptr(), src.data_ptr(), dst.data_ptr(), src.numel() * 4, wgs, unroll, 0, s))
            print(f"read  {mb:5d} MB  wgs {wgs:5d} unroll {unroll}: {us:8.1f} us  {mb * 1.048576 / us:.2f} TB/s", flush=True)
    us = timed(lambda: lib.mem_stream(out.data_ptr(), src.data_ptr(), dst.data_ptr(), src.numel() * 4, 4096, 4, 1, s))
    print(f"copy  {mb:5d} MB  wgs  4096 unroll 4: {us:8.1f} us  {2 * mb * 1.048576 / us:.2f} TB/s (read + write)", flush=True)
    del src, dst
n = 8192 * 512
dst = torch.zeros(16 * 1024 * 1024, dtype=torch.float32, device="cuda")
for span, what in ((n, "distinct addresses (4.2 M floats)"), (128 * 512, "128 rows x 512 (64 lanes per address)"), (512, "one row of 512")):
    us = timed(lambda: lib.mem_atomic(dst.data_ptr(), n, span, 2048, 8, s))
    print(f"atomicAdd f32: {n} atomics into {what:40s}: {us:8.1f} us  {n / us / 1e3:.1f} G atomics/s", flush=True)
