"""A/Bs in one process (C2 step): routing knobs of the GEMM family, the one-launch attention backward for short query blocks."""
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
    print(f"{name:76s} {dt / 20 * 1e3:7.3f} ms/step", flush=True)
base = dict(gemm_wide_min_tiles=160, attn_small=1, embed_det=1, gemm_ring128_min_k=1024, gemm_ring128_min_tiles=128)
for rep in range(3):
    run("defaults", **base)
    run("128x128 ring kernel for K >= 512 (o-projection 8192x512x512 and its dgrad)", **dict(base, gemm_ring128_min_k=512))
    run("128x128 ring kernel for K >= 512, from 64 tiles", **dict(base, gemm_ring128_min_k=512, gemm_ring128_min_tiles=64))
    if rep == 0:
        run("wide kernel from 128 tiles (N = 512 outputs on p5_gemm5<KC>)", **dict(base, gemm_wide_min_tiles=128))
        run("short-block attention backward as dQ + dK/dV launches", **dict(base, attn_small=0))
        run("embedding gradients by atomic scatter", **dict(base, embed_det=0))
