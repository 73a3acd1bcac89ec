"""run-to-run reproducibility of one backward on fresh models (first call after other work = fresh allocator blocks)"""
import sys, torch
sys.path.insert(0, ".")
from tests import cases
from openp5_amd._lib import hip_backend
from oracle import t5_oracle as O
be = hip_backend()
cases.model_train_case(be, O.T5Cfg.named("tiny"), 2, 16, 5, "fp32", 0.0, nll_tol=1e-3, grad_tol=1e-2)
junk = [torch.randn(1 << 24, device="cuda") * 1e3 for _ in range(8)]       # poison the allocator's free blocks
del junk
ocfg = O.T5Cfg.named("t5-small", dropout=0.0)
params = O.init_params(ocfg, 7)
a = cases.synth_batch(ocfg, 16, 64, 8, 3)
def run(mode):
    be.check(be.lib.p5_set_option(b"grad_store_first", mode), "opt")
    m = cases.build_model(be, ocfg, params, "bf16", 0.0)
    m.eval()
    loss = m.loss_and_backward(*a)
    torch.cuda.synchronize()
    return float(loss), m._grads.detach().cpu().clone(), m
res = [run(md) for md in (0, 0, 0, 0, 0, 0, 0, 0)]
for i in range(1, len(res)):
    d = (res[i][1] - res[0][1]).abs()
    print(i, "loss", res[i][0], res[0][0], "max grad diff vs run 0", float(d.max()), "n>1e-6", int((d > 1e-6).sum()), flush=True)
    if float(d.max()) > 1e-6:
        m = res[i][2]
        bad = [(n, float(d[o:o + k].max())) for n, (o, k, _) in m._views.items() if float(d[o:o + k].max()) > 1e-6]
        print("   ", [b[0] for b in bad][-12:], len(bad))
