"""In-process A/B of engine options on the other BASELINE configs (C3: T5-base B=64 L=128; C5: T5-large B=64 L=512), every variant twice.
usage: leg_ab.py c3|c5 name=opt:val[,opt:val] ..."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from openp5_amd._lib import hip_backend
be = hip_backend()
which = sys.argv[1]
backbone, B, L, T, steps, warm = {"c3": ("t5-base", 64, 128, 8, 8, 3), "c5": ("t5-large", 64, 512, 10, 4, 2)}[which]
cfg, model, opt = bench.build_model(backbone, "bf16", be.device, be, 1, 0)
batch = bench.synth_batch(B, L, T, be.device, 100)
variants = []
for a in sys.argv[2:]:
    name, _, spec = a.partition("=")
    variants.append((name, {k: int(v) for k, v in (kv.split(":") for kv in spec.split(",") if kv)}))
base = dict(variants[0][1])
for rep in range(2):
    for name, opts in variants:
        for k in base:
            assert be.lib.p5_set_option(k.encode(), opts.get(k, base[k])) == 0, k
        model.mark_params_updated()
        dt, loss = bench.time_training(model, opt, batch, steps, warm, 1, be.device)
        print(f"{which} {name:12s} {dt / steps * 1e3:8.3f} ms/step  loss {loss:.4f}", flush=True)
for k, v in base.items():
    be.lib.p5_set_option(k.encode(), v)
