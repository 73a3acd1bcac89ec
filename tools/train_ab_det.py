"""A/B in one process: deterministic embedding gradients (p5_embed.h) vs the fp32-atomic scatter of rounds 1-3; C2 step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from openp5_amd._lib import hip_backend
be = hip_backend()
dev = be.device
cfg, model, opt = bench.build_model("t5-small", "bf16", dev, be, 1, 0)
batch = bench.synth_batch(64, 128, 8, dev, 100)
def run(name, **opts):
    for k, v in opts.items():
        assert be.lib.p5_set_option(k.encode(), int(v)) == 0, k
    dt, loss = bench.time_training(model, opt, batch, 20, 5, 1, dev)
    print(f"{name:48s} {dt / 20 * 1e3:7.3f} ms/step  loss {loss:.4f}", flush=True)
for rep in range(3):
    run("embedding gradients: segmented sums (default)", embed_det=1)
    run("embedding gradients: fp32 atomic scatter", embed_det=0)
