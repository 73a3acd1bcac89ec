"""Feasibility probe: two half-batch (B=32) training steps issued on two streams vs one B=64 step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from openp5_amd._lib import hip_backend
from openp5_amd.model import P5ModelConfig, P5T5Native
from openp5_amd.optim import FusedAdamW
be = hip_backend()
cfg = P5ModelConfig.from_backbone("t5-small", vocab_size=bench.V, dropout_rate=0.1)
def mk(B, seed):
    m = P5T5Native(cfg, dtype="bf16", backend=be, seed=2023); m.train()
    o = FusedAdamW(m, lr=1e-3, warmup_steps=10, total_steps=1000)
    return m, o, bench.synth_batch(B, 128, 8, be.device, seed)
def step(m, o, b, opt=True):
    ids, ww, mask, labels, out_attn = b
    out = m(input_ids=ids, whole_word_ids=ww, attention_mask=mask, labels=labels)
    loss = bench.runner_loss(out["loss"], out_attn); loss.backward()
    if opt: o.step(); m.zero_grad()
m64 = mk(64, 1)
for _ in range(5): step(*m64)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step(*m64)
torch.cuda.synchronize(); print(f"one stream  B=64: {(time.perf_counter()-t0)/20*1e3:.3f} ms/step")
a, b = mk(32, 2), mk(32, 3)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def pair(opt=True):
    with torch.cuda.stream(sa): step(*a, opt=opt)
    with torch.cuda.stream(sb): step(*b, opt=opt)
for _ in range(5): pair()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): pair()
torch.cuda.synchronize(); print(f"two streams 2xB=32 (each with its own optimizer step): {(time.perf_counter()-t0)/20*1e3:.3f} ms/pair")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): pair(opt=False)
torch.cuda.synchronize(); print(f"two streams 2xB=32 fwd+bwd only: {(time.perf_counter()-t0)/20*1e3:.3f} ms/pair")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step(*m64, opt=False)
torch.cuda.synchronize(); print(f"one stream  B=64 fwd+bwd only: {(time.perf_counter()-t0)/20*1e3:.3f} ms/step")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step(*a, opt=False)
torch.cuda.synchronize(); print(f"one stream  B=32 fwd+bwd only: {(time.perf_counter()-t0)/20*1e3:.3f} ms/step")
