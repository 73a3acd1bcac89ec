"""not-gpu: the runner (train epoch -> evaluate) end to end on toy data with the host-emulated kernels, and the
data-parallel gradient path over gloo with world_size 2."""
import argparse
import os
import random
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch.utils.data import ConcatDataset, DataLoader

from openp5_amd.collator import Collator
from openp5_amd.data import MultiTaskDataset
from openp5_amd.model import P5ModelConfig, P5T5Native
from openp5_amd.optim import FusedAdamW
from openp5_amd.runner import DistributedRunner, masked_mean_loss
from openp5_amd.sampler import DistMultiDataTaskSampler, SingleMultiDataTaskSampler
from openp5_amd.tokenizer import build_offline_tokenizer
from tests.test_host import make_args

VOCAB = 2400


def tiny_model(be, vocab, dropout=0.0, seed=3, dtype="fp32"):
    cfg = P5ModelConfig(vocab_size=vocab, d_model=64, d_ff=128, num_layers=1, num_decoder_layers=1, num_heads=1, dropout_rate=dropout)
    return P5T5Native(cfg, dtype=dtype, backend=be, seed=seed)


def test_runner_train_and_eval(emu, tmp_path):
    tok = build_offline_tokenizer(VOCAB)
    args = make_args(str(tmp_path), ["--epochs", "1", "--test_before_train", "0", "--test_epoch", "0", "--metrics", "hit@5,ndcg@5",
                                     "--eval_batch_size", "6", "--batch_size", "8", "--sample_num", "1,1", "--max_his", "3"])
    args.model_path = str(tmp_path / "m.pt")
    random.seed(0)
    train = ConcatDataset([MultiTaskDataset(args, "Toy", "train")])
    sampler = SingleMultiDataTaskSampler(train, args.batch_size, args.seed)
    loader = DataLoader(train, sampler=sampler, batch_size=args.batch_size, collate_fn=Collator(tok))
    model = tiny_model(emu, len(tok))
    runner = DistributedRunner(model, tok, loader, None, torch.device("cpu"), args, 0)
    losses = runner.train()
    assert len(losses) == 1 and 0 < losses[0] < 20
    assert os.path.exists(args.model_path)
    sd = torch.load(args.model_path)
    assert "lm_head.weight" in sd and "encoder.whole_word_embeddings.weight" in sd     # HF key layout incl. tied duplicates
    res = runner.test()
    assert len(res) == 2 and all(0.0 <= r["hit@5"] <= 1.0 for r in res)
    # the loss goes down when the same batches are revisited
    model2 = tiny_model(emu, len(tok))
    opt = FusedAdamW(model2, lr=3e-3, max_grad_norm=1.0)
    batch = next(iter(loader))
    first = last = None
    model2.train()
    for _ in range(8):
        nll = model2(input_ids=batch[0], whole_word_ids=batch[2], attention_mask=batch[1], labels=batch[3])["loss"]
        loss = masked_mean_loss(nll, batch[4])
        loss.backward()
        opt.step()
        model2.zero_grad()
        first = float(loss) if first is None else first
        last = float(loss)
    assert last < first


def _eval_runner(emu, tmp_path, tok, extra):
    args = make_args(str(tmp_path), ["--epochs", "1", "--test_before_train", "0", "--test_epoch", "0", "--metrics",
                                     "hit@1,hit@5,ndcg@5", "--batch_size", "8", "--sample_num", "1,1", "--max_his", "8"] + extra)
    random.seed(0)
    train = ConcatDataset([MultiTaskDataset(args, "Toy", "train")])
    loader = DataLoader(train, sampler=SingleMultiDataTaskSampler(train, args.batch_size, args.seed), batch_size=args.batch_size,
                        collate_fn=Collator(tok))
    return DistributedRunner(tiny_model(emu, len(tok)), tok, loader, None, torch.device("cpu"), args, 0)


def test_id_metrics_equal_string_metrics(emu, tmp_path):
    """token-id relevance on the device == decode + string equality (DistributedRunner.py:376-387)."""
    tok = build_offline_tokenizer(VOCAB)
    a = _eval_runner(emu, tmp_path, tok, ["--eval_batch_size", "6", "--id_metrics", "1"]).test()
    b = _eval_runner(emu, tmp_path, tok, ["--eval_batch_size", "6", "--id_metrics", "0"]).test()
    assert a == b


def test_filtered_batch_id_metrics_equal_string_metrics(emu, tmp_path):
    """--test_filtered 1 --test_filtered_batch 1 (widened beam, history filtered afterwards): item-index form == string form."""
    tok = build_offline_tokenizer(VOCAB)
    flags = ["--test_filtered", "1", "--test_filtered_batch", "1", "--eval_batch_size", "6"]
    a = _eval_runner(emu, tmp_path, tok, flags + ["--id_metrics", "1"]).test()
    b = _eval_runner(emu, tmp_path, tok, flags + ["--id_metrics", "0"]).test()
    assert a == b and any(v > 0 for r in a for v in r.values())


def test_filtered_protocol_matches_per_user_tries(emu, tmp_path):
    """--test_filtered 1 --test_filtered_batch 0: shared trie + per-user bitmap, at batch size 5, against the reference
    protocol restated literally: batch size 1, a fresh Trie(all_items - positive) per user, string metrics."""
    from openp5_amd import evaluate
    from openp5_amd.trie import Trie, prefix_allowed_tokens_fn
    tok = build_offline_tokenizer(VOCAB)
    flags = ["--test_filtered", "1", "--test_filtered_batch", "0"]
    fast = _eval_runner(emu, tmp_path, tok, flags + ["--eval_batch_size", "5"])
    got = fast.test()
    slow = _eval_runner(emu, tmp_path, tok, flags + ["--eval_batch_size", "1", "--id_metrics", "0"])
    got_b1 = slow.test()
    ref = []
    slow.model.eval()
    for loader in slow.testloaders:
        ds = loader.dataset
        sums, n = 0, 0
        for batch in loader:
            positive = ds.positive[ds.id2user[int(batch[5][0])]]
            fn = prefix_allowed_tokens_fn(Trie(slow._item_sequences(ds, set(ds.all_items) - positive)))
            gold, gen, scores = slow._generate(batch, fn, slow.generate_num, 30)
            rel = evaluate.rel_results(gen, gold, scores, slow.generate_num)
            sums = sums + evaluate.get_metrics_results(rel, slow.metrics)
            n += len(rel)
        ref.append(dict(zip(slow.metrics, (sums / n).tolist())))
    for g, g1, r in zip(got, got_b1, ref):
        for k in r:
            assert abs(g[k] - r[k]) < 1e-12 and abs(g1[k] - r[k]) < 1e-12, (g, g1, r)


def _ddp_worker(rank, world, port, tmp, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.emu.emu_backend import emu_backend
    be = emu_backend()
    tok = build_offline_tokenizer(VOCAB)
    model = tiny_model(be, len(tok), seed=3)
    model.ddp_world = world
    opt = FusedAdamW(model, lr=1e-2, max_grad_norm=1.0)
    g = torch.Generator().manual_seed(100 + rank)            # each rank gets its own shard of the global batch
    ids = torch.randint(3, len(tok), (4, 12), generator=g)
    mask = torch.ones_like(ids)
    labels = torch.randint(3, len(tok), (4, 5), generator=g)
    model.eval()
    nll = model(input_ids=ids, whole_word_ids=torch.zeros_like(ids), attention_mask=mask, labels=labels)["loss"]
    masked_mean_loss(nll, torch.ones_like(labels)).backward()
    opt.step()
    torch.save({"flat": model._flat.clone(), "grads": model._grads.clone(), "ids": ids, "labels": labels}, os.path.join(tmp, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_gloo(emu, tmp_path):
    """world_size 2 over gloo: after one step both ranks hold identical parameters, equal to ONE process stepping on the
    mean of the two shards' gradients (i.e. the gradient all-reduce the reference's DDP wrapper never performs)."""
    world, port = 2, 29000 + random.randint(0, 2000)
    mp.spawn(_ddp_worker, args=(world, port, str(tmp_path), None), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["flat"], r1["flat"]), "ranks diverged"
    assert torch.allclose(r0["grads"], r1["grads"])
    # single-process reference: the summed gradient arena must equal g0 + g1
    tok = build_offline_tokenizer(VOCAB)
    gs = []
    for r in (r0, r1):
        m = tiny_model(emu, len(tok), seed=3)
        m.eval()
        nll = m(input_ids=r["ids"], whole_word_ids=torch.zeros_like(r["ids"]), attention_mask=torch.ones_like(r["ids"]), labels=r["labels"])["loss"]
        masked_mean_loss(nll, torch.ones_like(r["labels"])).backward()
        gs.append(m._grads.clone())
    assert torch.allclose(r0["grads"], gs[0] + gs[1], atol=1e-6, rtol=1e-5)
    m = tiny_model(emu, len(tok), seed=3)
    opt = FusedAdamW(m, lr=1e-2, max_grad_norm=1.0)
    m._grads.copy_(0.5 * (gs[0] + gs[1]))
    opt.step()
    assert torch.allclose(m._flat, r0["flat"], atol=1e-6, rtol=1e-5)
