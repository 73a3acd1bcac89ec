"""Upper bound of what running two generation batches at once buys: N host threads, each with its OWN model replica, workspaces and HIP stream,
calling generate() in a loop (ctypes and torch release the GIL inside their calls).  python tools/gen_lanes_probe.py [lanes] [n]; P5_GEN_MODE."""
import os, sys, time, threading, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from openp5_amd._lib import hip_backend
from openp5_amd.model import P5ModelConfig, P5T5Native
from openp5_amd.trie import prefix_allowed_tokens_fn
be = hip_backend()
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mode = os.environ.get("P5_GEN_MODE", "verified")
cfg = P5ModelConfig.from_backbone("t5-small", vocab_size=bench.V, dropout_rate=0.1)
fn = prefix_allowed_tokens_fn(bench.synth_item_trie(3416, 7))
ids, ww, mask, _, _ = bench.synth_batch(20, 128, 8, be.device, 500)
models = []
for i in range(lanes):
    m = P5T5Native(cfg, dtype="bf16", backend=be, seed=2023); m.eval(); m.generation_mode = mode
    models.append(m)
kw = dict(input_ids=ids, attention_mask=mask, whole_word_ids=ww, max_length=30, prefix_allowed_tokens_fn=fn, num_beams=10, num_return_sequences=10,
          output_scores=True, return_dict_in_generate=True)
streams = [torch.cuda.Stream() for _ in range(lanes)]
def work(i, reps):
    with torch.cuda.stream(streams[i]):
        for _ in range(reps):
            models[i].generate(**kw)
        streams[i].synchronize()
for i in range(lanes): work(i, 3)
torch.cuda.synchronize()
for nl in sorted({1, lanes}):
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i, n)) for i in range(nl)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"mode {mode}: {nl} lane(s) x {n} batches of 20 users x beam 10: {dt / (nl * n) * 1e3:.3f} ms per batch, {nl * n * 200 / dt:.0f} items/s")
