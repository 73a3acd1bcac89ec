"""Host side of tools/probe/lds_probe.hip: cycles per wave instruction (4 waves on one CU) for the LDS patterns of p5_attn.h."""
import ctypes, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "liblds_probe.so"))
out = torch.zeros(2048, dtype=torch.int32, device="cuda")
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
names = {0: "tr_b64 pair x4 (8 instr)", 1: "read_b128 x4", 2: "write_b16 x16", 3: "write_b64 x4", 4: "read_u16 diag x16"}
ninstr = {0: 8, 1: 4, 2: 16, 3: 4, 4: 16}
iters = 20000
clk_ghz = 2.4
for pattern in range(5):
    line = f"{names[pattern]:26s}"
    for stride in (128, 136, 144, 160, 176, 272, 288):
        if pattern in (1, 3) and stride % 16 and pattern == 1: 
            line += f"  S={stride}:   -  "; continue
        lib.lds_probe(ctypes.c_void_p(out.data_ptr()), stride, pattern, 200, s); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.lds_probe(ctypes.c_void_p(out.data_ptr()), stride, pattern, iters, s); e1.record(); torch.cuda.synchronize()
        ns = e0.elapsed_time(e1) * 1e6 / iters
        line += f"  S={stride}: {ns * clk_ghz / ninstr[pattern] / 16:5.1f}"
    print(line + "   (LDS-pipe cycles per wave instruction: kernel time / iterations / instructions / 16 waves, at 2.4 GHz)", flush=True)
