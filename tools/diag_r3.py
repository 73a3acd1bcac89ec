"""Round-3 diagnostic (dev tool, GPU box): bf16 engine vs fp32 oracle gradients at T5-base dims, per tensor, under the engine's
option switches -- to localise which bf16-only kernel path is wrong at d_model = 768 / 12 heads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import t5_oracle as O
from tests import cases
from openp5_amd._lib import hip_backend

be = hip_backend()
torch.set_num_threads(min(os.cpu_count() or 1, 32))
name = sys.argv[1] if len(sys.argv) > 1 else "t5-base"
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B, L, T = (int(x) for x in (sys.argv[3:6] if len(sys.argv) > 5 else (8, 128, 8)))
cfg = O.T5Cfg.named(name, num_layers=nl, num_decoder_layers=nl, dropout=0.0)
params = O.init_params(cfg, 7)
ids, ww, mask, labels, out_attn = cases.synth_batch(cfg, B, L, T, 3)
Pq = {k: v.clone().requires_grad_(True) for k, v in params.items()}
O.runner_loss(O.p5_forward_nll(Pq, cfg, ids, ww, mask, labels), out_attn).backward()


def run(label, opts):
    for k, v in opts.items():
        be.check(be.lib.p5_set_option(k.encode(), v), k)
    m = cases.build_model(be, cfg, params, "bf16")
    m.eval()
    m.loss_and_backward(ids, ww, mask, labels, out_attn)
    torch.cuda.synchronize()
    rows, whole = cases.grad_agreement(m, Pq)
    rows.sort(reverse=True)
    print(f"[{label}] whole relL2 {whole[0]:.3e} cos {whole[1]:.6f}; worst tensors:")
    for r in rows[:6]:
        print(f"     relL2 {r[0]:.3e} cos {r[1]:.6f} {r[2]}")
    bad = [r[2] for r in rows if r[0] > 0.3]
    print(f"     {len(bad)} tensors with relL2 > 0.3" + (": " + ", ".join(b.replace('.weight', '') for b in bad[:12]) if bad else ""))
    del m
    torch.cuda.empty_cache()


base = {"wgrad_group": 1, "attn_fused": 1, "attn_fwd_wg": 1, "dgrad_t": 1, "gemm_small_ring": 1, "gemm_ring32": 128, "gemm_ring": 1, "gemm_ksdma": 1,
        "gemm_xcd_rect": 1}
run("default (grouped weight gradients)", base)
base0 = dict(base)
base0["wgrad_group"] = 0
run("wgrad_group=0", base0)
for k, v in (("attn_fused", 0), ("attn_fwd_wg", 0), ("dgrad_t", 0), ("gemm_small_ring", 0), ("gemm_ring32", 0), ("gemm_ring", 0),
             ("gemm_ksdma", 0), ("gemm_xcd_rect", 0)):
    o = dict(base0)
    o[k] = v
    run(f"wgrad_group=0 + {k}={v}", o)
o = dict(base0)
o.update({"gemm_ring": 0, "gemm_small_ring": 0, "gemm_ring32": 0})
run("wgrad_group=0, no ring kernels at all", o)
