"""Dev tool (GPU box): how the decision margins of the dataset-level gate (tests/test_gpu_dataset.py) depend on the synthetic signal and
the number of epochs.  usage: gate_explore.py <signal: none|chain> <epochs> [lr] [n_oracle_batches]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import t5_oracle as O
from tests import cases
from openp5_amd._lib import hip_backend
from openp5_amd.model import P5T5Native
from openp5_amd import synth

hip = hip_backend()
signal = None if sys.argv[1] == "none" else sys.argv[1]
epochs = int(sys.argv[2])
lr = sys.argv[3] if len(sys.argv) > 3 else "1e-3"
nob = int(sys.argv[4]) if len(sys.argv) > 4 else 2
orig = synth.make_user_sequences
synth.make_user_sequences = lambda dataset, **kw: orig(dataset, signal=signal, **kw)
tmp = tempfile.mkdtemp()
t0 = time.time()
runner, model, tok, args = cases.make_pipeline(hip, tmp, "bf16", flags=["--epochs", str(epochs), "--lr", lr])
losses = runner.train()
torch.cuda.synchronize()
print(f"[gate {sys.argv[1]} x{epochs} lr {lr}] train {time.time() - t0:.1f}s losses {losses[0]:.3f} -> {losses[-1]:.3f}", flush=True)
model.eval()
K = 10
r16 = cases.collect_rankings(runner, cases.engine_gen_fn(model), K)
sd = {k: v.detach().cpu().float().clone() for k, v in model.state_dict().items()}
m32 = P5T5Native(model.config, dtype="fp32", backend=hip, seed=1)
m32.load_state_dict(sd, strict=False)
m32.eval()
r32 = cases.collect_rankings(runner, cases.engine_gen_fn(m32), K)
c = cases.compare_rankings(r16, r32, tie_tol=0.04)
print("   bf16 vs fp32 engine:", {k: v for k, v in c.items() if k != "diff_users"}, flush=True)
print("   metrics fp32", cases.rankings_metrics(r32), " bf16", cases.rankings_metrics(r16))
gaps = sorted(min((a - b) for a, b in zip(s[:-1], s[1:])) for us in r32 for (_, _, s) in us)
n = len(gaps)
print(f"   min final-score gap per user (fp32 engine), quantiles 10/50/90%: {gaps[n // 10]:.4f} {gaps[n // 2]:.4f} {gaps[9 * n // 10]:.4f}")
# oracle margins on the first `nob` batches of each loader
ocfg = O.T5Cfg.named("t5-small", dropout=0.0, vocab_size=model.config.vocab_size)
params = {k: sd[k] for k in O.param_shapes(ocfg)}
torch.set_num_threads(min(os.cpu_count() or 1, 32))
margins = []
fn = cases.oracle_gen_fn(params, ocfg, margins)
t0 = time.time()
for loader in runner.testloaders:
    trie, ct, _ = runner._dataset_trie(loader.dataset)
    for i, batch in enumerate(loader):
        if i >= nob:
            break
        fn(batch, trie, ct, K, 50)
sm = sorted(m[0] for m in margins)
n = len(sm)
print(f"   oracle set-margins over {n} users ({time.time() - t0:.0f}s): quantiles 10/50/90% {sm[n // 10]:.4f} {sm[n // 2]:.4f} {sm[9 * n // 10]:.4f}; "
      f"robust at 0.04: {sum(1 for m in margins if m[0] > 0.04 and all(g > 0.04 for g in m[1]))}/{n} list, {sum(1 for m in margins if m[0] > 0.04)}/{n} set", flush=True)
