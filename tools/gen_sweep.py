"""Decode-step knob sweep in ONE process (p5_set_option re-keys the captured step graph): beam-10, 20 users, 3416-item trie.
usage: gen_sweep.py [users] [iters]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from openp5_amd._lib import hip_backend
from openp5_amd.model import P5ModelConfig, P5T5Native
from openp5_amd.trie import prefix_allowed_tokens_fn
be = hip_backend()
cfg = P5ModelConfig.from_backbone("t5-small", vocab_size=bench.V, dropout_rate=0.1)
model = P5T5Native(cfg, dtype="bf16", backend=be, seed=2023); model.eval()
fn = prefix_allowed_tokens_fn(bench.synth_item_trie(3416, 7))
gB = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ids, ww, mask, _, _ = bench.synth_batch(gB, 128, 8, be.device, 500)
kw = dict(input_ids=ids, attention_mask=mask, whole_word_ids=ww, max_length=30, prefix_allowed_tokens_fn=fn, num_beams=10, num_return_sequences=10,
          output_scores=True, return_dict_in_generate=True)
DEFAULT = dict(decode_v2=1, dec_cross=3, dec_head=1, dec_nb=0, dec_head_nv=0, dec_fuseq=1, dec_kw=0)
def run(**opts):
    cfgd = dict(DEFAULT); cfgd.update(opts)
    for k, v in cfgd.items():
        assert be.lib.p5_set_option(k.encode(), int(v)) == 0, k
    for _ in range(3):
        o = model.generate(**kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        o = model.generate(**kw)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{str(opts):60s} ms/batch {dt*1e3:7.3f}  items/s {gB*10/dt:8.0f}  len {o['sequences'].shape[1]}", flush=True)
    return o
# fp32 engine with the same weights = the ranking every bf16 variant is measured against
m32 = P5T5Native(cfg, dtype="fp32", backend=be, seed=2023); m32.eval()
with torch.no_grad():
    m32._flat.copy_(model._flat)
m32.mark_params_updated()
truth = m32.generate(**kw)
def agree(o, name):
    same = (truth["sequences"].shape == o["sequences"].shape) and int((truth["sequences"] == o["sequences"]).all(dim=1).sum())
    both = (truth["sequences"] == o["sequences"]).all(dim=1) if same is not False else None
    md = float((truth["sequences_scores"] - o["sequences_scores"])[both].abs().max()) if both is not None and both.any() else float("nan")
    print(f"   vs fp32 engine [{name}]: rows identical {same} / {truth['sequences'].shape[0]}, max |score diff| on identical rows {md:.5f}", flush=True)
ref = run(dec_cross=2, dec_head=0); agree(ref, "scalar cross-attention, materialised logits")
base = run(); agree(base, "defaults")
agree(run(dec_cross=2), "scalar cross-attention, streaming head")
agree(run(dec_head=0), "MFMA cross-attention, materialised logits")
run(dec_head_nv=64)
run(dec_fuseq=0)
run(decode_v2=0, dec_head=0)
run()
kw32 = dict(kw)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    m32.generate(**kw32)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"fp32 parity mode: ms/batch {dt*1e3:7.3f}  items/s {gB*10/dt:8.0f}")
