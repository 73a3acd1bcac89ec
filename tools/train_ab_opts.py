"""In-process A/B of engine options on the C2 training step (T5-small, B=64, L=128, T=8, bf16): every variant twice, alternating.
usage: train_ab_opts.py name=opt:val[,opt:val] ...      e.g.  base= n512=gemm_wide_min_tiles:128 adamflat=adam_tiles:0"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from openp5_amd._lib import hip_backend
be = hip_backend()
cfg, model, opt = bench.build_model("t5-small", "bf16", be.device, be, 1, 0)
batch = bench.synth_batch(64, 128, 8, be.device, 100)
DEFAULTS = {"gemm_wide_min_tiles": 160, "adam_tiles": 1, "gemm_ring_n512": 1, "ce_free": 1}
variants = []
for a in sys.argv[1:]:
    name, _, spec = a.partition("=")
    variants.append((name, {k: int(v) for k, v in (kv.split(":") for kv in spec.split(",") if kv)}))
for rep in range(2):
    for name, opts in variants:
        for k, v in {**DEFAULTS, **opts}.items():
            assert be.lib.p5_set_option(k.encode(), v) == 0, k
        model.mark_params_updated()          # (copies rebuilt under the variant's rules)
        dt, loss = bench.time_training(model, opt, batch, 20, 5, 1, be.device)
        rows = bench.kernel_classes(bench.profile_training(be, model, opt, batch))
        kc = [c for c in rows if "gemm5" in c["kernel"] and "KC" in c["kernel"]]
        n5 = [c for c in rows if "128x128 KC" in c["kernel"]]
        ad = [c for c in rows if "adamw" in c["kernel"] or "transpose" in c["kernel"] or "fold_rows" in c["kernel"]]
        hd = [c for c in rows if "cross-entropy" in c["kernel"] or "p5_ce_" in c["kernel"] or ("gemm5" in c["kernel"] and "KC" in c["kernel"] and any("512x32100" in g["grid"] for g in c["grids"]))]
        print(f"{name:10s} {dt / 20 * 1e3:7.3f} ms/step  loss {loss:.4f}  KC {sum(c['us_per_step'] for c in kc):6.0f} us ({sum(c['launches_per_step'] for c in kc):.0f})"
              f"  ring128 {sum(c['us_per_step'] for c in n5):5.0f} us ({sum(c['launches_per_step'] for c in n5):.0f})  adam+copies {sum(c['us_per_step'] for c in ad):5.0f} us  ce {sum(c['us_per_step'] for c in hd if 'gemm5' not in c['kernel'] or 'cross-entropy' in c['kernel']):4.0f} us", flush=True)
for k, v in DEFAULTS.items():
    be.lib.p5_set_option(k.encode(), v)
