"""Generation-only benchmark / profiling target (beam-10, 20 users, 3416-item trie)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from openp5_amd._lib import hip_backend
from openp5_amd.model import P5ModelConfig, P5T5Native
from openp5_amd.trie import prefix_allowed_tokens_fn
be = hip_backend()
cfg = P5ModelConfig.from_backbone("t5-small", vocab_size=bench.V, dropout_rate=0.1)
model = P5T5Native(cfg, dtype=os.environ.get("P5_GEN_DTYPE", "bf16"), backend=be, seed=2023); model.eval()
fn = prefix_allowed_tokens_fn(bench.synth_item_trie(3416, 7))
gB = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ids, ww, mask, _, _ = bench.synth_batch(gB, 128, 8, be.device, 500)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for _ in range(2):
    o = model.generate(input_ids=ids, attention_mask=mask, whole_word_ids=ww, max_length=30, prefix_allowed_tokens_fn=fn, num_beams=10, num_return_sequences=10, output_scores=True, return_dict_in_generate=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    o = model.generate(input_ids=ids, attention_mask=mask, whole_word_ids=ww, max_length=30, prefix_allowed_tokens_fn=fn, num_beams=10, num_return_sequences=10, output_scores=True, return_dict_in_generate=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"B={gB} ms/batch {dt*1e3:.3f} items/s {gB*10/dt:.0f} decoded_len {o['sequences'].shape[1]}")
