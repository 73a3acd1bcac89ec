"""In-process A/B of engine options on the C2 training step (T5-small, B=64, L=128, T=8, bf16): every variant twice, alternating; prints
ms/step and the step's event-bracketed microseconds of the kernel classes whose name contains one of the --show substrings.
usage: train_ab6.py [--show sub1,sub2,...] name=opt:val[,opt:val] ..."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from openp5_amd._lib import hip_backend
be = hip_backend()
cfg, model, opt = bench.build_model("t5-small", "bf16", be.device, be, 1, 0)
batch = bench.synth_batch(64, 128, 8, be.device, 100)
args = sys.argv[1:]
show = ["gemm5", "gemm2", "p5_gemm_kernel", "rmsnorm_bwd"]
if args and args[0] == "--show":
    show = args[1].split(","); args = args[2:]
variants = []
for a in args:
    name, _, spec = a.partition("=")
    variants.append((name, {k: int(v) for k, v in (kv.split(":") for kv in spec.split(",") if kv)}))
allopts = {}
for _, o in variants:
    for k in o: allopts.setdefault(k, None)
base = dict(variants[0][1])            # the first variant defines the value every other variant resets an option to
for rep in range(2):
    for name, opts in variants:
        for k in allopts:
            v = opts.get(k, base.get(k))
            if v is not None: assert be.lib.p5_set_option(k.encode(), v) == 0, k
        model.mark_params_updated()
        dt, loss = bench.time_training(model, opt, batch, 20, 5, 1, be.device)
        rows = bench.kernel_classes(bench.profile_training(be, model, opt, batch))
        parts = []
        for sub in show:
            sel = [c for c in rows if sub in c["kernel"]]
            parts.append(f"{sub} {sum(c['us_per_step'] for c in sel):5.0f}us({sum(c['launches_per_step'] for c in sel):.0f})")
        print(f"{name:12s} {dt / 20 * 1e3:7.3f} ms/step  loss {loss:.4f}  " + "  ".join(parts), flush=True)
for k, v in base.items():
    be.lib.p5_set_option(k.encode(), v)
