"""A/Bs in one process (C2 step): routing of the N = d_model GEMMs, the one-launch attention backward for short query blocks."""
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
    print(f"{name:72s} {dt / 20 * 1e3:7.3f} ms/step", flush=True)
for rep in range(3):
    run("defaults", gemm_wide_min_tiles=160, attn_small=1, embed_det=1)
    run("wide kernel from 128 tiles (N = 512 outputs on p5_gemm5<KC>)", gemm_wide_min_tiles=128)
    run("wide kernel from 64 tiles", gemm_wide_min_tiles=64)
    run("defaults, short-block attention backward as dQ + dK/dV launches", gemm_wide_min_tiles=160, attn_small=0)
    run("defaults, embedding gradients by atomic scatter", attn_small=1, embed_det=0)
