"""Debug: per-user filtered generation -- per-user Trie vs shared trie + bitmap, batch 1 vs batch 10, graph on/off."""
import os, random, sys, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.utils.data import ConcatDataset, DataLoader
from openp5_amd._lib import hip_backend
from openp5_amd.collator import Collator
from openp5_amd.data import MultiTaskDataset
from openp5_amd.model import P5ModelConfig, P5T5Native
from openp5_amd.runner import DistributedRunner
from openp5_amd.sampler import SingleMultiDataTaskSampler
from openp5_amd.tokenizer import build_offline_tokenizer
from openp5_amd.trie import Trie, prefix_allowed_tokens_fn
from tests.test_host import make_args

be = hip_backend()
tok = build_offline_tokenizer()
tmp = tempfile.mkdtemp()
flags = ["--epochs", "1", "--test_before_train", "0", "--test_epoch", "0", "--metrics", "hit@1,hit@5,ndcg@10", "--batch_size", "16",
         "--sample_num", "1,1", "--max_his", "10", "--test_filtered", "1", "--test_filtered_batch", "0"]
cfg = P5ModelConfig.from_backbone("t5-small", dropout_rate=0.0)
model = P5T5Native(cfg, dtype="fp32", backend=be, seed=5)
model.resize_token_embeddings(len(tok))
model.eval()

def runner_for(extra):
    args = make_args(tmp, flags + extra)
    random.seed(0)
    train = ConcatDataset([MultiTaskDataset(args, "Toy", "train")])
    loader = DataLoader(train, sampler=SingleMultiDataTaskSampler(train, args.batch_size, args.seed), batch_size=args.batch_size, collate_fn=Collator(tok))
    return DistributedRunner(model, tok, loader, None, be.device, args, 0)

r1 = runner_for(["--eval_batch_size", "1"])
r10 = runner_for(["--eval_batch_size", "10"])
K = r1.generate_num
def gen(batch, **kw):
    out = model.generate(input_ids=batch[0], attention_mask=batch[1], whole_word_ids=batch[2], max_length=30, num_beams=K, num_return_sequences=K,
                         output_scores=True, return_dict_in_generate=True, **kw)
    B = batch[0].shape[0]
    S = out["sequences"].view(B, K, -1).cpu()
    S = torch.nn.functional.pad(S, (0, 30 - S.shape[-1]))
    return S, out["sequences_scores"].view(B, K).cpu()

loader1, loader10 = r1.testloaders[0], r10.testloaders[0]
ds = loader1.dataset
_, ct, index = r1._dataset_trie(ds)
per_user = {}
for batch in loader1:
    batch = r1._to_dev(batch)
    u = int(batch[5][0])
    positive = ds.positive[ds.id2user[u]]
    fn = prefix_allowed_tokens_fn(Trie(r1._item_sequences(ds, set(ds.all_items) - positive)))
    a = gen(batch, prefix_allowed_tokens_fn=fn)
    ex = ct.excluded_bitmap([[index[i] for i in positive if i in index]])
    b = gen(batch, trie=ct, excluded=ex)
    per_user[u] = (a, b, batch[0].shape[1])
    if not torch.equal(a[0], b[0]):
        print(f"user {u} L={batch[0].shape[1]}: B=1 per-user-trie != B=1 bitmap; score diff {(a[1]-b[1]).abs().max():.3e}")
        print(" trie :", a[0][0, :3].tolist(), a[1][0, :4].tolist())
        print(" bitmp:", b[0][0, :3].tolist(), b[1][0, :4].tolist())
for batch in loader10:
    batch = r10._to_dev(batch)
    users = batch[5].tolist()
    ex = ct.excluded_bitmap([[index[i] for i in ds.positive[ds.id2user[u]] if i in index] for u in users])
    S, sc = gen(batch, trie=ct, excluded=ex)
    for j, u in enumerate(users):
        a, b, L1 = per_user[u]
        if not torch.equal(S[j], a[0][0]):
            d = (sc[j] - a[1][0]).abs().max()
            print(f"user {u} (L1={L1}, L10={batch[0].shape[1]}): B=10 bitmap != B=1 trie; score diff {d:.3e}; equal to B=1 bitmap: {torch.equal(S[j], b[0][0])}")
            print("  b10:", sc[j, :5].tolist()); print("  b1 :", a[1][0, :5].tolist())
print("done")
