"""Host side of tools/probe/load_probe.hip: operand-fetch time of the attention forward for two Q/K/V layouts (B=64, H=8, L=128)."""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libload_probe.so"))
B, H, L = 64, 8, 128
buf = torch.randint(0, 2 ** 31 - 1, (B * L * 3 * H * 128 // 4,), dtype=torch.int32, device="cuda")
flush = torch.empty(768 * 1024 * 1024 // 4, dtype=torch.int32, device="cuda")
out = torch.zeros(B * H, dtype=torch.int32, device="cuda")
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for layout, name in ((0, "fused rows [B*L, 3*H*64] (today)"), (1, "head-major [3][B,H,L,64]")):
    for cold in (False, True):
        ts = []
        for _ in range(12):
            if cold: flush.fill_(1)            # evict L2 / MALL between launches
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); lib.load_probe(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(buf.data_ptr()), B, H, L, layout, s); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts[2:])
        mb = 1024 * 40960 / 1e6
        print(f"{name:36s} {'cold (768 MB fill between launches)' if cold else 'warm (25 MB buffer re-read)':38s} median {ts[len(ts)//2]:6.1f} us  "
              f"= {mb / ts[len(ts)//2]:.2f} TB/s over {mb:.0f} MB requested", flush=True)
