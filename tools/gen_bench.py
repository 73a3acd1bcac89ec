"""Generation-only benchmark / profiling target (20 users, 3416-item trie): python tools/gen_bench.py [B] [n] [K] ; P5_GEN_DTYPE=bf16|fp32,
P5_GEN_MODE=draft|verified, P5_GEN_EXTRA=<extra draft beams>."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from openp5_amd._lib import hip_backend
from openp5_amd.model import P5ModelConfig, P5T5Native
from openp5_amd.trie import prefix_allowed_tokens_fn
be = hip_backend()
cfg = P5ModelConfig.from_backbone("t5-small", vocab_size=bench.V, dropout_rate=0.1)
model = P5T5Native(cfg, dtype=os.environ.get("P5_GEN_DTYPE", "bf16"), backend=be, seed=2023); model.eval()
model.generation_mode = os.environ.get("P5_GEN_MODE", "draft")
model.verify_extra_beams = int(os.environ.get("P5_GEN_EXTRA", "6"))
fn = prefix_allowed_tokens_fn(bench.synth_item_trie(3416, 7))
gB = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ids, ww, mask, _, _ = bench.synth_batch(gB, 128, 8, be.device, 500)
kw = dict(input_ids=ids, attention_mask=mask, whole_word_ids=ww, max_length=30, prefix_allowed_tokens_fn=fn, num_beams=K, num_return_sequences=K,
          output_scores=True, return_dict_in_generate=True)
for _ in range(2):
    o = model.generate(**kw)
model.time_generate(True)
torch.cuda.synchronize(); t0 = time.perf_counter()
tm = []
for _ in range(n):
    o = model.generate(**kw)
    tm.append(model.last_generate_timing())
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
enc = sorted(t["encode_ms"] for t in tm)[n // 2]; dec = sorted(t["decode_ms"] for t in tm)[n // 2]
lanes = int(os.environ.get("P5_GEN_LANES", "0"))
if lanes > 1:        # several batches in flight (P5T5Native.map_lanes)
    model.time_generate(False)
    list(model.map_lanes(lambda k: model.generate(**k), [kw] * (2 * lanes), lanes=lanes))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    list(model.map_lanes(lambda k: model.generate(**k), [kw] * (4 * n), lanes=lanes))
    torch.cuda.synchronize(); dl = (time.perf_counter() - t0) / (4 * n)
    print(f"{lanes} lanes: {dl*1e3:.3f} ms per batch in flight, {gB*K/dl:.0f} items/s")
print(f"dtype {os.environ.get('P5_GEN_DTYPE', 'bf16')} mode {model.generation_mode}+{model.verify_extra_beams} {model.verify_stats} B={gB} K={K} ms/batch {dt*1e3:.3f} (median device: encode {enc:.3f} decode {dec:.3f}) items/s {gB*K/dt:.0f} "
      f"decoded_len {o['sequences'].shape[1]}")
