"""which gradient tensors differ between the storing backward and the clear-then-accumulate backward (and run to run)"""
import sys, torch
sys.path.insert(0, ".")
from tests import cases
from openp5_amd._lib import hip_backend
from oracle import t5_oracle as O

def run(be, ocfg, mode, B, L, T):
    params = O.init_params(ocfg, 7)
    a = cases.synth_batch(ocfg, B, L, T, 3)
    b = cases.synth_batch(ocfg, B, L, T, 4)
    be.check(be.lib.p5_set_option(b"grad_store_first", mode), "opt")
    m = cases.build_model(be, ocfg, params, "bf16", 0.0)
    m.eval()
    m.loss_and_backward(*a)
    m.zero_grad()
    m.begin_micro_batch(first=True, sync=False)
    m.loss_and_backward(*b)
    torch.cuda.synchronize()
    return m, m._grads.detach().cpu().clone()

be = hip_backend()
ocfg = O.T5Cfg.named("t5-small", dropout=0.0)
m, g1 = run(be, ocfg, 1, 16, 64, 8)
_, g1b = run(be, ocfg, 1, 16, 64, 8)
_, g0 = run(be, ocfg, 0, 16, 64, 8)
_, g0b = run(be, ocfg, 0, 16, 64, 8)
print("store vs store", float((g1 - g1b).abs().max()), "accum vs accum", float((g0 - g0b).abs().max()), "store vs accum", float((g1 - g0).abs().max()))
for name, (off, n, shape) in m._views.items():
    d = float((g1[off:off + n] - g0[off:off + n]).abs().max())
    d2 = float((g0[off:off + n] - g0b[off:off + n]).abs().max())
    if d > 0 or d2 > 0:
        print(f"{name:70s} store-accum {d:.3e}  accum-accum {d2:.3e}  max {float(g0[off:off+n].abs().max()):.3e}")
