"""How long does the HOST need to issue one training step (no sync)?  If close to the GPU step time we are launch-bound."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from openp5_amd._lib import hip_backend
from openp5_amd.model import P5ModelConfig, P5T5Native
from openp5_amd.optim import FusedAdamW
be = hip_backend()
cfg = P5ModelConfig.from_backbone("t5-small", vocab_size=bench.V, dropout_rate=0.1)
model = P5T5Native(cfg, dtype="bf16", backend=be, seed=2023); model.train()
opt = FusedAdamW(model, lr=1e-3, warmup_steps=10, total_steps=1000)
ids, ww, mask, labels, out_attn = bench.synth_batch(64, 128, 8, be.device, 1)
def step():
    out = model(input_ids=ids, whole_word_ids=ww, attention_mask=mask, labels=labels)
    loss = bench.runner_loss(out["loss"], out_attn); loss.backward(); opt.step(); model.zero_grad()
for _ in range(5): step()
torch.cuda.synchronize()
# GPU-bound rate
t0 = time.perf_counter()
for _ in range(20): step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"issue {t_issue/20*1e3:.2f} ms/step, total {t_all/20*1e3:.2f} ms/step")
# pure host cost: time the python+ctypes+launch path with the GPU idle-ish (sync before each step)
hs = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); hs.append(time.perf_counter() - t0)
print("host-only issue time per step (ms):", [round(h*1e3, 2) for h in hs])
import cProfile, pstats
pr = cProfile.Profile(); torch.cuda.synchronize(); pr.enable()
for _ in range(5): step()
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
