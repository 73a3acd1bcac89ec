"""Per-kernel table (library profiler: HIP events around every launch) of the training step at another BASELINE config.
usage: leg_profile.py c3|c5|c2"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from openp5_amd._lib import hip_backend
be = hip_backend()
which = sys.argv[1]
backbone, B, L, T = {"c2": ("t5-small", 64, 128, 8), "c3": ("t5-base", 64, 128, 8), "c5": ("t5-large", 64, 512, 10)}[which]
cfg, model, opt = bench.build_model(backbone, "bf16", be.device, be, 1, 0)
batch = bench.synth_batch(B, L, T, be.device, 100)
dt, loss = bench.time_training(model, opt, batch, 4, 2, 1, be.device)
print(f"{which} {dt / 4 * 1e3:.3f} ms/step")
rows = bench.kernel_classes(bench.profile_training(be, model, opt, batch, steps=2))
tot = sum(c["us_per_step"] for c in rows)
for c in rows[:28]:
    tf = c["flops_per_step"] / max(c["us_per_step"], 1e-9) / 1e6 if c["flops_per_step"] > 0 else 0
    print(f"  {c['us_per_step']:9.0f} us {100 * c['us_per_step'] / tot:5.1f}%  x{c['launches_per_step']:5.0f}  {tf:6.0f} TF/s  {c['kernel'][:110]}")
    for g in (c.get("grids") or [])[:6]:
        print(f"        {g['grid'][:60]:60s} x{g['launches_per_step']:4.0f} {g['avg_us']:8.1f} us {g.get('tflops') or 0:6.0f} TF/s")
