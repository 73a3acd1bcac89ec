import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from openp5_amd._lib import hip_backend
from openp5_amd.model import P5ModelConfig, P5T5Native
from openp5_amd.optim import FusedAdamW
be = hip_backend()
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
def run(hi, side):
    P5T5Native.use_side_stream = side
    cfg = P5ModelConfig.from_backbone("t5-small", vocab_size=bench.V, dropout_rate=0.1)
    model = P5T5Native(cfg, dtype="bf16", backend=be, seed=2023); model.train()
    opt = FusedAdamW(model, lr=1e-3, warmup_steps=10, total_steps=1000)
    ids, ww, mask, labels, out_attn = bench.synth_batch(64, 128, 8, be.device, 1)
    def step():
        out = model(input_ids=ids, whole_word_ids=ww, attention_mask=mask, labels=labels)
        loss = bench.runner_loss(out["loss"], out_attn); loss.backward(); opt.step(); model.zero_grad()
    st = torch.cuda.Stream(priority=-1) if hi else torch.cuda.current_stream()
    with torch.cuda.stream(st):
        for _ in range(5): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f"hi_prio_main={hi} side_stream={side}: {dt*1e3:.3f} ms/step")
run(False, True); run(True, True); run(False, False); run(True, True); run(False, True)
