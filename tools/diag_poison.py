"""workspace poisoned with NaN bytes: which gradients / outputs pick up an uninitialised read?"""
import os, sys, torch
os.environ["P5_POISON_WS"] = "1"
sys.path.insert(0, ".")
from tests import cases
from openp5_amd._lib import hip_backend
from oracle import t5_oracle as O
be = hip_backend()
for name, shp in (("t5-small", (16, 64, 8)), ("t5-small", (64, 128, 8)), ("tiny", (4, 16, 16))):
    ocfg = O.T5Cfg.named(name, dropout=0.0)
    params = O.init_params(ocfg, 7)
    a = cases.synth_batch(ocfg, *shp, 3)
    for drop in (0.0, 0.1):
        m = cases.build_model(be, O.T5Cfg(**{**ocfg.__dict__, "dropout": drop}), params, "bf16", drop)
        m.train() if drop > 0 else m.eval()
        if drop > 0:
            m.set_dropout_seed(1234, 0)
        loss = m.loss_and_backward(*a)
        torch.cuda.synchronize()
        g = m._grads
        bad = [(n, int(torch.isnan(g[o:o + k]).sum())) for n, (o, k, _) in m._views.items() if bool(torch.isnan(g[o:o + k]).any())]
        print(name, shp, "dropout", drop, "loss", float(loss), "tensors with NaN:", len(bad), [b for b in bad][-6:], flush=True)
