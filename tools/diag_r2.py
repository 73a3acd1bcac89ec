"""Round-2 diagnostics on the GPU box (dev tool): bf16-vs-oracle gradient errors at the C2 shape, the T5-large fp32
tolerance root cause (fp64 oracle), dataset-level ranking agreement of the bf16 / fp32 engines with the fp32 oracle."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import t5_oracle as O
from tests import cases
from openp5_amd._lib import hip_backend

be = hip_backend()
what = sys.argv[1:] or ["bf16grad", "large", "dataset"]
torch.set_num_threads(min(os.cpu_count() or 1, 32))


def grad_stats(m, Pq, label):
    rows = []
    gmax = max(float(v.grad.abs().max()) for v in Pq.values())
    for name, p in m.named_parameters():
        g, go = p.grad.detach().cpu().double().flatten(), Pq[name].grad.double().flatten()
        rel = float((g - go).norm() / (go.norm() + 1e-30))
        cos = float((g @ go) / (g.norm() * go.norm() + 1e-30))
        mx = float((g - go).abs().max() / (go.abs().max() + 1e-2 * gmax))
        rows.append((rel, cos, mx, name, float(go.norm())))
    rows.sort(reverse=True)
    print(f"[{label}] worst by rel-L2:")
    for r in rows[:8]:
        print(f"   relL2 {r[0]:.3e} cos {r[1]:.6f} maxrel {r[2]:.3e} |g| {r[4]:.3e} {r[3]}")
    print(f"[{label}] median relL2 {sorted(r[0] for r in rows)[len(rows)//2]:.3e}, min cos {min(r[1] for r in rows):.6f}, max maxrel {max(r[2] for r in rows):.3e}")
    allg = torch.cat([p.grad.detach().cpu().double().flatten() for _, p in m.named_parameters()])
    allo = torch.cat([Pq[n].grad.double().flatten() for n, _ in m.named_parameters()])
    print(f"[{label}] whole-gradient relL2 {float((allg-allo).norm()/allo.norm()):.3e} cos {float((allg@allo)/(allg.norm()*allo.norm())):.6f}")


if "bf16grad" in what:
    ocfg = O.T5Cfg.named("t5-small", dropout=0.0)
    params = O.init_params(ocfg, 7)
    ids, ww, mask, labels, out_attn = cases.synth_batch(ocfg, 64, 128, 8, 3)
    t0 = time.time()
    Pq = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    nll_o = O.p5_forward_nll(Pq, ocfg, ids, ww, mask, labels)
    O.runner_loss(nll_o, out_attn).backward()
    print(f"oracle fp32 C2 fwd+bwd {time.time()-t0:.1f}s")
    for dt in ("fp32", "bf16"):
        m = cases.build_model(be, ocfg, params, dt)
        m.eval()
        nll = m(input_ids=ids, whole_word_ids=ww, attention_mask=mask, labels=labels)["loss"]
        O.runner_loss(nll, out_attn.to(nll.device)).backward()
        torch.cuda.synchronize()
        e = (nll.detach().cpu() - nll_o.detach())
        print(f"[{dt}] nll max err {float(e.abs().max()):.3e} mean abs {float(e.abs().mean()):.3e} (nll mean {float(nll_o.mean()):.3f})")
        grad_stats(m, Pq, dt)
        del m

if "large" in what:
    cfg = O.T5Cfg.named("t5-large", num_layers=2, num_decoder_layers=2, dropout=0.0)
    params = O.init_params(cfg, 7)
    ids, ww, mask, labels, out_attn = cases.synth_batch(cfg, 1, 512, 10, 3)
    P32 = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    O.runner_loss(O.p5_forward_nll(P32, cfg, ids, ww, mask, labels), out_attn).backward()
    P64 = {k: v.double().clone().requires_grad_(True) for k, v in params.items()}
    O.runner_loss(O.p5_forward_nll(P64, cfg, ids, ww, mask, labels), out_attn.double()).backward()
    m = cases.build_model(be, cfg, params, "fp32")
    m.eval()
    nll = m(input_ids=ids, whole_word_ids=ww, attention_mask=mask, labels=labels)["loss"]
    O.runner_loss(nll, out_attn.to(nll.device)).backward()
    torch.cuda.synchronize()
    gmax = max(float(v.grad.abs().max()) for v in P64.values())
    worst = []
    for name, p in m.named_parameters():
        g = p.grad.detach().cpu().double()
        g32, g64 = P32[name].grad.double(), P64[name].grad
        den = float(g64.abs().max()) + 1e-2 * gmax
        worst.append((float((g - g64).abs().max()) / den, float((g32 - g64).abs().max()) / den, float((g - g32).abs().max()) / den, name))
    worst.sort(reverse=True)
    print("[large] engine-vs-fp64 | oracle32-vs-fp64 | engine-vs-oracle32   (max abs / (max|g64| + 1% gmax))")
    for w in worst[:10]:
        print(f"   {w[0]:.3e} | {w[1]:.3e} | {w[2]:.3e}  {w[3]}")
    print(f"[large] max engine-vs-fp64 {max(w[0] for w in worst):.3e}; max oracle32-vs-fp64 {max(w[1] for w in worst):.3e}; max engine-vs-oracle32 {max(w[2] for w in worst):.3e}")

if "dataset" in what:
    tmp = tempfile.mkdtemp()
    runner, model, tok, args = cases.make_pipeline(be, tmp, "bf16", flags=["--epochs", "6", "--lr", "1e-3"])
    t0 = time.time()
    losses = runner.train()
    print(f"[dataset] trained 6 epochs in {time.time()-t0:.1f}s, losses {losses}")
    model.eval()
    K = 10
    r_bf16 = cases.collect_rankings(runner, cases.engine_gen_fn(model), K)
    sd = {k: v.detach().cpu().float().clone() for k, v in model.state_dict().items()}
    from openp5_amd.model import P5T5Native
    m32 = P5T5Native(model.config, dtype="fp32", backend=be, seed=1)
    m32.load_state_dict(sd, strict=False)
    m32.eval()
    r_fp32 = cases.collect_rankings(runner, cases.engine_gen_fn(m32), K)
    ocfg = O.T5Cfg.named("t5-small", dropout=0.0, vocab_size=model.config.vocab_size)
    params = {k: sd[k] for k in O.param_shapes(ocfg)}
    t0 = time.time()
    r_or = cases.collect_rankings(runner, cases.oracle_gen_fn(params, ocfg), K)
    print(f"[dataset] oracle evaluation {time.time()-t0:.1f}s")
    print("[dataset] metrics bf16  ", cases.rankings_metrics(r_bf16))
    print("[dataset] metrics fp32  ", cases.rankings_metrics(r_fp32))
    print("[dataset] metrics oracle", cases.rankings_metrics(r_or))
    print("[dataset] fp32 engine vs oracle:", {k: v for k, v in cases.compare_rankings(r_fp32, r_or).items()})
    print("[dataset] bf16 engine vs oracle:", {k: v for k, v in cases.compare_rankings(r_bf16, r_or).items()})
    # score gaps around the gold / rank boundaries in the oracle ranking
    gaps = []
    for users in r_or:
        for gold, ranked, sc in users:
            gaps += [sc[i] - sc[i + 1] for i in range(len(sc) - 1)]
    gaps.sort()
    print(f"[dataset] oracle adjacent-score gaps: min {gaps[0]:.2e} p1 {gaps[len(gaps)//100]:.2e} p10 {gaps[len(gaps)//10]:.2e} median {gaps[len(gaps)//2]:.2e}")
