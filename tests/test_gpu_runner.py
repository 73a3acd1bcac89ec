"""-m gpu: the whole src_t5 pipeline on the MI355X -- synthetic user sequences -> datasets -> sampler -> collator ->
P5T5Native (T5-small dims, bf16) -> fused optimizer -> constrained beam search -> Hit/NDCG, through the runner."""
import os
import random

import pytest
import torch
from torch.utils.data import ConcatDataset, DataLoader

from openp5_amd.collator import Collator
from openp5_amd.data import MultiTaskDataset
from openp5_amd.model import P5ModelConfig, P5T5Native
from openp5_amd.runner import DistributedRunner
from openp5_amd.sampler import SingleMultiDataTaskSampler
from openp5_amd.tokenizer import build_offline_tokenizer
from openp5_amd.utils.initialization import random_initialization
from tests.test_host import make_args

pytestmark = pytest.mark.gpu


def test_runner_end_to_end(hip, tmp_path):
    tok = build_offline_tokenizer()
    args = make_args(str(tmp_path), ["--epochs", "3", "--test_before_train", "0", "--test_epoch", "0", "--metrics", "hit@5,hit@10,ndcg@10",
                                     "--eval_batch_size", "10", "--batch_size", "16", "--sample_num", "2,2", "--max_his", "10", "--lr", "1e-3"])
    args.model_path = str(tmp_path / "toy.pt")
    random.seed(0)
    train = ConcatDataset([MultiTaskDataset(args, "Toy", "train")])
    loader = DataLoader(train, sampler=SingleMultiDataTaskSampler(train, args.batch_size, args.seed), batch_size=args.batch_size,
                        collate_fn=Collator(tok))
    cfg = P5ModelConfig.from_backbone("t5-small", dropout_rate=0.1)
    model = P5T5Native.from_pretrained("t5-small", config=cfg, dtype="bf16", backend=hip, seed=args.seed)
    model.resize_token_embeddings(len(tok))
    assert model.shared.weight.shape == (32100, 512)
    random_initialization(model, tok, "t5-small")
    runner = DistributedRunner(model, tok, loader, None, hip.device, args, 0)
    losses = runner.train()
    assert len(losses) == 3 and losses[-1] < losses[0], losses
    assert runner.samples_per_sec > 0
    res = runner.test()
    assert len(res) == 2 and all(0.0 <= r["hit@10"] <= 1.0 for r in res)
    # the checkpoint written by the runner reloads into a fresh model and reproduces the evaluation exactly
    sd = torch.load(args.model_path)
    assert "lm_head.weight" in sd and sd["shared.weight"].shape[0] == 32100
    model2 = P5T5Native(cfg, dtype="bf16", backend=hip, seed=1)
    model2.resize_token_embeddings(len(tok))
    model2.load_state_dict(sd, strict=False)
    runner2 = DistributedRunner(model2, tok, loader, None, hip.device, args, 0)
    res2 = runner2.test()
    assert res2 == res


def test_filtered_evaluation_on_device(hip, tmp_path):
    """--test_filtered 1 --test_filtered_batch 0 (the released test_command protocol): shared device trie + per-user
    excluded-node bitmaps at batch 10 == the reference's one-trie-per-user protocol at batch 1 with string metrics."""
    from openp5_amd import evaluate
    from openp5_amd.trie import Trie, prefix_allowed_tokens_fn
    tok = build_offline_tokenizer()
    flags = ["--epochs", "1", "--test_before_train", "0", "--test_epoch", "0", "--metrics", "hit@1,hit@5,ndcg@10", "--batch_size", "16",
             "--sample_num", "1,1", "--max_his", "10", "--test_filtered", "1", "--test_filtered_batch", "0"]
    cfg = P5ModelConfig.from_backbone("t5-small", dropout_rate=0.0)
    model = P5T5Native(cfg, dtype="fp32", backend=hip, seed=5)
    model.resize_token_embeddings(len(tok))

    def runner_for(extra):
        args = make_args(str(tmp_path), flags + extra)
        random.seed(0)
        train = ConcatDataset([MultiTaskDataset(args, "Toy", "train")])
        loader = DataLoader(train, sampler=SingleMultiDataTaskSampler(train, args.batch_size, args.seed), batch_size=args.batch_size,
                            collate_fn=Collator(tok))
        return DistributedRunner(model, tok, loader, None, hip.device, args, 0)

    fast = runner_for(["--eval_batch_size", "10"])
    got = fast.test()
    slow = runner_for(["--eval_batch_size", "1", "--id_metrics", "0"])
    ref = []
    for loader in slow.testloaders:
        ds = loader.dataset
        sums, n = 0, 0
        for batch in loader:
            batch = slow._to_dev(batch)
            positive = ds.positive[ds.id2user[int(batch[5][0])]]
            fn = prefix_allowed_tokens_fn(Trie(slow._item_sequences(ds, set(ds.all_items) - positive)))
            gold, gen, scores = slow._generate(batch, fn, slow.generate_num, 30)
            rel = evaluate.rel_results(gen, gold, scores, slow.generate_num)
            sums = sums + evaluate.get_metrics_results(rel, slow.metrics)
            n += len(rel)
        ref.append(dict(zip(slow.metrics, (sums / n).tolist())))
    for g, r in zip(got, ref):
        for k in r:
            assert abs(g[k] - r[k]) < 1e-12, (got, ref)


def test_resume_on_device(hip, tmp_path):
    """f3 on the device (the reference saves weights only, utils/utils.py:119-129): 2 epochs straight == 1 epoch + resume file + everything
    rebuilt + `--resume 1` + the 2nd epoch, bf16 engine with dropout on the MI355X -- optimizer m/v/t, schedule position, dropout
    counter, data order and the bf16 shadow all have to survive the round trip -- BIT FOR BIT, as on the host emulation
    (tests/test_runner_emu.py::test_resume_is_exact), and two straight runs of the same epochs are bit-identical as well.

    Round 3 left this as an open question: straight runs differed by 2.4e-7 .. 5.5e-4 and the resumed run by up to 5.7e-3 (2 x lr).
    Root cause (round 4, tools/diag_repro.py, profiles/r04_repro_before_fix.txt): at step 0 ONLY the gradients that were sums of fp32
    atomics differed between two runs (embedding scatter-adds, T5LayerNorm weight partials reduced by 16 atomically adding
    workgroups, relative-bias tables), in their last bits (1e-10 .. 1e-8) -- and AdamW's update lr * m / (sqrt(v) + eps) is +-lr
    whatever the magnitude of a gradient, so a last-bit difference of a near-zero gradient element becomes an O(lr) difference of
    that parameter.  No race, no lost state.  The fix is to have no fp32 atomics in the training step at all (p5_embed.h,
    rel_bias_grad_flush in p5_attn.h, reduce_rows_16col in p5_elem.h, single-split ungrouped weight gradients), which makes the
    comparison exact."""
    from tests.test_host import SMALL_TOY
    from tests.test_runner_emu import VOCAB, tiny_model
    tok = build_offline_tokenizer(VOCAB)

    def build(epochs, extra=()):
        args = make_args(str(tmp_path), ["--epochs", str(epochs), "--test_before_train", "0", "--test_epoch", "0", "--batch_size", "8",
                                         "--sample_num", "1,1", "--max_his", "3", "--lr", "3e-3"] + list(extra), toy=SMALL_TOY)
        args.model_path = str(tmp_path / "m.pt")
        random.seed(0)
        train = ConcatDataset([MultiTaskDataset(args, "Toy", "train")])
        loader = DataLoader(train, sampler=SingleMultiDataTaskSampler(train, args.batch_size, args.seed), batch_size=args.batch_size,
                            collate_fn=Collator(tok))
        model = tiny_model(hip, len(tok), dropout=0.1, seed=9, dtype="bf16")
        model.set_dropout_seed(77, 0)
        return DistributedRunner(model, tok, loader, None, hip.device, args, 0)

    straight = build(2)
    straight.optimizer.total_steps = 2 * len(straight.train_loader)         # same schedule in all runs
    l2 = straight.train()
    again = build(2)                                                        # the same two epochs once more
    again.optimizer.total_steps = straight.optimizer.total_steps
    l2b = again.train()
    assert torch.equal(again.model._flat, straight.model._flat), float((again.model._flat - straight.model._flat).abs().max())
    assert torch.equal(again.optimizer.m, straight.optimizer.m) and torch.equal(again.optimizer.v, straight.optimizer.v)
    assert l2b == l2
    first = build(1, ["--resume", "1"])
    first.optimizer.total_steps = straight.optimizer.total_steps
    first.optimizer.warmup_steps = straight.optimizer.warmup_steps
    first.train()
    assert os.path.exists(str(tmp_path / "m.pt") + ".resume")
    second = build(2, ["--resume", "1"])
    second.optimizer.total_steps = straight.optimizer.total_steps
    second.optimizer.warmup_steps = straight.optimizer.warmup_steps
    l12 = second.train()
    assert second.optimizer.t == straight.optimizer.t and second.optimizer.sched_steps == straight.optimizer.sched_steps
    diff = float((second.model._flat - straight.model._flat).abs().max())
    print(f"[resume] max |param diff| resumed vs straight {diff:.3e}")
    assert torch.equal(second.model._flat, straight.model._flat), diff
    assert torch.equal(second.optimizer.m, straight.optimizer.m) and torch.equal(second.optimizer.v, straight.optimizer.v)
    assert len(l12) == len(l2) and l12[-1] == l2[-1], (l12, l2)
