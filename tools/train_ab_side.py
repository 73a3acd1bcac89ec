"""C2 training step with / without the engine's side stream (weight gradients of the decoder layers and of the tied head off the main
stream, where the decoder's latency-bound chain of 512-row kernels leaves most CUs idle).   usage: train_ab_side.py <0|1> [wgrad_side bits]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from openp5_amd._lib import hip_backend
from openp5_amd.model import P5T5Native
side = int(sys.argv[1]) if len(sys.argv) > 1 else 0
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 1
P5T5Native.use_side_stream = bool(side)
be = hip_backend()
assert be.lib.p5_set_option(b"wgrad_side", bits) == 0
cfg, model, opt = bench.build_model("t5-small", "bf16", be.device, be, 1, 0)
batch = bench.synth_batch(64, 128, 8, be.device, 100)
res = []
for rep in range(3):
    dt, loss = bench.time_training(model, opt, batch, 20, 5, 1, be.device)
    res.append(dt / 20 * 1e3)
print(f"side_stream={side} wgrad_side={bits}: ms/step {['%.3f' % r for r in res]} loss {loss:.4f}", flush=True)
