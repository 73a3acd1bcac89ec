"""Host harness of tools/probe/tr_wait_probe.hip (build line in its header): every variant at several grid sizes, mismatches counted on the device.
   python tools/probe/tr_wait_probe.py            (on the GPU box)"""
import ctypes, os, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libtr_wait_probe.so"))
lib.tr_wait_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
NAMES = {0: "full wait (control)", 1: "partial waits", 2: "partial + VALU sentinel writes of the destinations", 3: "partial + sentinel + s_load outstanding BEFORE the reads",
         4: "partial + sentinel + s_load issued AFTER the reads", 5: "partial + s_load before", 6: "partial + s_load after", 7: "full wait + sentinel + s_load before"}
src = torch.randn(1024, device="cuda")
total_bad = 0
for variant in range(8):
    line = []
    for wgs, iters in ((1, 20000), (256, 4000), (1024, 2000), (4096, 500)):
        out = torch.zeros(2, dtype=torch.int32, device="cuda")
        rc = lib.tr_wait_probe(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(src.data_ptr()), variant, wgs, iters, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        bad, n = [int(x) & 0xFFFFFFFF for x in out.tolist()]
        assert rc == 0 and n == iters, (rc, n)
        total_bad += bad
        line.append(f"{wgs} wgs x {iters}: {bad}")
    print(f"variant {variant} [{NAMES[variant]}]: mismatching (lane, accumulator) pairs -> " + "; ".join(line), flush=True)
print("TOTAL mismatches", total_bad, "(rounds per variant: 4 waves x wgs x iters)")
