"""not-gpu: host-side logic of the path (collator, trie, metrics, prompts, samplers, datasets) -- known-answer values
from SURVEY.md App. D and, when /root/reference is present (build container only), equality with the reference's own
modules run on the same inputs."""
import argparse
import os
import random
import sys

import numpy as np
import pytest
import torch

from openp5_amd import evaluate
from openp5_amd.collator import Collator, TestCollator, calculate_whole_word_ids
from openp5_amd.data import MultiTaskDataset, TestDataset
from openp5_amd.sampler import DistMultiDataTaskSampler, SingleMultiDataTaskSampler, parse_sampler_args
from openp5_amd.synth import write_dataset, write_prompt_file
from openp5_amd.tokenizer import build_offline_tokenizer
from openp5_amd.trie import CompiledTrie, Trie, find_trie, prefix_allowed_tokens_fn
from openp5_amd.utils import utils
from openp5_amd.utils.prompt import get_info_from_prompt, load_prompt_template

REF = "/root/reference/src/src_t5"
HAVE_REF = os.path.isdir(REF)


@pytest.fixture(scope="module")
def tok():
    return build_offline_tokenizer()


def ref_module(name):
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.append(REF)
    import importlib
    return importlib.import_module(name)


def test_whole_word_ids_kat():
    p1 = ['▁ML', '1', 'M', '▁user', '_', '12', '▁item', '_', '100', '1', '</s>', '<pad>', '<pad>']
    assert calculate_whole_word_ids(p1, list(range(13))) == [1, 1, 1, 2, 2, 2, 3, 3, 3, 3, 3, 0, 0]
    p2 = ['▁ML', '1', 'M', '▁user', '_', '12', '▁item', '_', '100', '1', '▁,', '▁x', '</s>']
    assert calculate_whole_word_ids(p2, list(range(13))) == [1, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 5, 0]


def test_collator_matches_scalar_definition(tok):
    batch = [{"input": "What would ML1M user_12 be likely to purchase next after buying ML1M items item_1001 , item_1002 ?", "output": "ML1M item_1038"},
             {"input": "What should we recommend for ML1M user_7 ?", "output": "ML1M item_2"},
             {"input": "ML1M user_3 has purchased ML1M items item_5 , item_77 , item_1234", "output": "ML1M item_999", "user_idx": 3}]
    ids, attn, ww, out_ids, out_attn = Collator(tok)(batch)
    assert ids.dtype == torch.int64 and ids.shape == attn.shape == ww.shape
    for r in range(ids.shape[0]):
        pieces = tok.convert_ids_to_tokens(ids[r].tolist())
        assert ww[r].tolist() == calculate_whole_word_ids(pieces, ids[r].tolist())
    assert out_ids[0].tolist()[-1] in (0, 1) and out_attn.sum() > 0
    for b in batch:
        b.setdefault("user_idx", 0)
    assert TestCollator(tok)(batch)[5].tolist() == [0, 0, 3]


@pytest.mark.skipif(not HAVE_REF, reason="reference tree not present")
def test_whole_word_ids_equals_reference(tok):
    ref = ref_module("processor.Collator")
    rnd = random.Random(0)
    for _ in range(50):
        n = rnd.randint(2, 30)
        ids = [rnd.randint(3, 400) for _ in range(n)] + [1] + [0] * rnd.randint(0, 5)
        pieces = tok.convert_ids_to_tokens(ids)
        assert calculate_whole_word_ids(pieces, ids) == ref.calculate_whole_word_ids(pieces, ids)
        got = Collator(tok).whole_word_ids(np.asarray([ids]))[0].tolist()
        assert got == ref.calculate_whole_word_ids(pieces, ids)


def test_trie_kat_and_csr():
    t = Trie([[0, 5, 6, 1], [0, 5, 7, 1]])
    assert t.get([0]) == [5] and sorted(t.get([0, 5])) == [6, 7] and t.get([0, 5, 6]) == [1]
    assert t.get([0, 5, 6, 1]) == [] and t.get([0, 9]) == [] and len(t) == 2
    ct = CompiledTrie.from_trie(t)
    toks, nodes = ct.children(0)
    assert toks.tolist() == [0]
    toks, nodes = ct.children(int(nodes[0]))
    assert toks.tolist() == [5]
    toks, nodes = ct.children(int(nodes[0]))
    assert toks.tolist() == [6, 7] and ct.max_children == 2
    fn = prefix_allowed_tokens_fn(t)
    assert fn(0, torch.tensor([0, 5])) == t.get([0, 5]) and find_trie(fn) is t


@pytest.mark.skipif(not HAVE_REF, reason="reference tree not present")
def test_trie_equals_reference():
    gt = ref_module("utils.generation_trie")
    rnd = random.Random(1)
    seqs = [[0] + [rnd.randint(2, 9) for _ in range(rnd.randint(1, 5))] + [1] for _ in range(200)]
    a, b = Trie(seqs), gt.Trie(seqs)
    assert len(a) == len(b)
    for _ in range(500):
        pre = rnd.choice(seqs)[: rnd.randint(0, 6)]
        assert sorted(a.get(pre)) == sorted(b.get(pre))
    assert find_trie(gt.prefix_allowed_tokens_fn(b)) is b      # the reference's closure is recognised -> device path


def test_metrics_kat():
    rel = evaluate.rel_results(['a', 'b', 'c', 'd'], ['b', 'x'], [-1, -2, -1.5, -3], 2)
    assert rel == [[0, 1], [0, 0]]
    res = evaluate.get_metrics_results(rel, ['hit@1', 'hit@2', 'ndcg@2'])
    assert np.allclose(res, [0.0, 1.0, 0.63092975])


@pytest.mark.skipif(not HAVE_REF, reason="reference tree not present")
def test_metrics_equal_reference():
    ref = ref_module("utils.evaluate")
    rnd = random.Random(2)
    k, B = 10, 30
    preds = [f"i{rnd.randint(0, 40)}" for _ in range(B * k)]
    gold = [f"i{rnd.randint(0, 40)}" for _ in range(B)]
    scores = [rnd.random() for _ in range(B * k)]
    assert evaluate.rel_results(preds, gold, scores, k) == ref.rel_results(preds, gold, scores, k)
    ms = ['hit@5', 'hit@10', 'ndcg@5', 'ndcg@10']
    rel = evaluate.rel_results(preds, gold, scores, k)
    assert np.allclose(evaluate.get_metrics_results(rel, ms), ref.get_metrics_results(rel, ms))
    pos = {f"u{b}": set(rnd.sample(preds, 5)) for b in range(B)}
    id2user = {b: f"u{b}" for b in range(B)}
    a = evaluate.rel_results_filtered(pos, id2user, list(range(B)), k, preds, gold, scores, 5)
    b_ = ref.rel_results_filtered(pos, id2user, list(range(B)), k, preds, gold, scores, 5)
    assert a == b_


SMALL_TOY = dict(n_users=12, n_items=24, n_inter=90)     # a third of the Toy set: the emulated-kernel runner tests that only need "some epochs"


def make_args(tmp, extra=(), toy=None):
    parser = argparse.ArgumentParser()
    utils.parse_global_args(parser)
    MultiTaskDataset.parse_dataset_args(parser)
    parse_sampler_args(parser)
    from openp5_amd.runner import parse_runner_args
    parse_runner_args(parser)
    prompt = write_prompt_file(os.path.join(tmp, "prompt.txt"))
    write_dataset(os.path.join(tmp, "data"), "Toy", **(toy or {}))
    args = parser.parse_args(["--data_path", os.path.join(tmp, "data"), "--datasets", "Toy", "--tasks", "sequential,straightforward",
                              "--item_indexing", "sequential", "--prompt_file", prompt, "--sample_prompt", "1", "--sample_num", "3,3",
                              "--max_his", "20", "--distributed", "0", "--batch_size", "4", "--eval_batch_size", "5"] + list(extra))
    args.rank = 0
    return args


def test_prompt_and_dataset(tmp_path):
    args = make_args(str(tmp_path))
    tpl = load_prompt_template(args.prompt_file, ["sequential", "straightforward"])
    assert set(tpl) == {"sequential", "straightforward"} and "0" in tpl["sequential"]["seen"] and "0" in tpl["sequential"]["unseen"]
    assert set(get_info_from_prompt(tpl)) == {"dataset", "user_id", "history", "target"}
    random.seed(0)
    ds = MultiTaskDataset(args, "Toy", "train")
    n = len(ds.data_samples)
    assert len(ds) == 6 * n and ds.task_index == [3 * n, 6 * n]
    assert sorted(ds.item_map.values())[0] == "1001"
    s = ds[0]
    assert s["output"].startswith("Toy item_") and "Toy user_1 " in s["input"]
    td = TestDataset(argparse.Namespace(**{**vars(args), "test_filtered": 0}), "Toy", "sequential")
    assert len(td) == 30 and td[0]["output"].startswith("Toy item_")
    # leave-one-out: the test target is the user's last item, validation target the one before
    last = ds.reindex_user_seq_dict["1"][-1]
    assert td[0]["output"] == f"Toy item_{last}"
    assert os.path.exists(os.path.join(args.data_path, "Toy", "user_sequence_sequential_indexing_original.txt"))


def test_samplers(tmp_path):
    from torch.utils.data import ConcatDataset
    args = make_args(str(tmp_path))
    random.seed(0)
    ds = MultiTaskDataset(args, "Toy", "train")
    cat = ConcatDataset([ds])
    n = len(ds.data_samples)
    s = SingleMultiDataTaskSampler(cat, 4, seed=2023)
    s.set_epoch(0)
    idx = list(iter(s))
    assert len(idx) == len(s)
    for i in range(0, len(idx), 4):             # task-homogeneous batches, alternating tasks
        grp = idx[i:i + 4]
        assert all(g < 3 * n for g in grp) or all(g >= 3 * n for g in grp)
        assert (grp[0] >= 3 * n) == ((i // 4) % 2 == 1)
    # disjoint strided shards over two ranks (separate dataset copies, as in real multi-process runs)
    shards = []
    for r in range(2):
        random.seed(0)
        d = MultiTaskDataset(args, "Toy", "train")
        sm = DistMultiDataTaskSampler(ConcatDataset([d]), 4, 2, r, seed=2023)
        sm.set_epoch(0)
        shards.append(list(iter(sm)))
    assert len(shards[0]) == len(shards[1])
    assert set(shards[0]) | set(shards[1]) == set(range(6 * n))
    assert len(set(shards[0]) & set(shards[1])) == 0


@pytest.mark.skipif(not HAVE_REF, reason="reference tree not present")
def test_dataset_and_sampler_equal_reference(tmp_path):
    """Same toy data + flags through the reference's own MultiTaskDataset / TestDataset / samplers."""
    from torch.utils.data import ConcatDataset
    RD = ref_module("data.MultiTaskDataset")
    RT = ref_module("data.TestDataset")
    RS = ref_module("processor.DistMultiDataTaskSampler")
    a1 = make_args(str(tmp_path / "a"))
    a2 = make_args(str(tmp_path / "b"))
    for a in (a1, a2):
        a.test_filtered = 0
    random.seed(5)
    mine = MultiTaskDataset(a1, "Toy", "train")
    random.seed(5)
    ref = RD.MultiTaskDataset(a2, "Toy", "train")
    assert mine.data["input"] == ref.data["input"] and mine.data["output"] == ref.data["output"]
    assert mine.task_index == ref.task_index and mine.item_map == ref.item_map
    tm, tr = TestDataset(a1, "Toy", "straightforward"), RT.TestDataset(a2, "Toy", "straightforward")
    assert tm.data == tr.data and tm.all_items == tr.all_items
    for epoch in (0, 1):
        sm = DistMultiDataTaskSampler(ConcatDataset([mine]), 4, 2, 1, seed=2023)
        sr = RS.DistMultiDataTaskSampler(ConcatDataset([ref]), 4, 2, 1, seed=2023)
        sm.set_epoch(epoch); sr.set_epoch(epoch)
        assert list(iter(sm)) == list(iter(sr)) and len(sm) == len(sr)


def test_word_cache_encoder_equals_tokenizer(tok, tmp_path):
    """the cached word-level tokenisation used by the collator is identical to the tokenizer's own batch encoding"""
    from openp5_amd.collator import WordCacheEncoder
    args = make_args(str(tmp_path))
    random.seed(1)
    ds = MultiTaskDataset(args, "Toy", "train")
    texts = [ds[i]["input"] for i in range(0, len(ds), 7)] + [ds[i]["output"] for i in range(0, len(ds), 11)]
    texts += ["  leading and   multiple   spaces ", "x" * 5 + " " + " ".join(f"item_{i}" for i in range(700))]    # incl. a > 512-token prompt
    enc = WordCacheEncoder(tok)
    assert enc.ok
    ids, mask = enc(texts)
    ref = tok(texts, padding="longest", truncation=True, max_length=512)
    assert ids.tolist() == ref["input_ids"] and mask.tolist() == ref["attention_mask"]
    a = Collator(tok, fast=True)([{"input": t, "output": "Toy item_1001"} for t in texts[:9]])
    b = Collator(tok, fast=False)([{"input": t, "output": "Toy item_1001"} for t in texts[:9]])
    assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_excluded_bitmap_equals_filtered_trie():
    """full CSR trie + excluded-node bitmap allows exactly the tokens of Trie(remaining items) at every prefix."""
    import random as _r
    from openp5_amd.trie import CompiledTrie, Trie
    rnd = _r.Random(4)
    items = sorted({tuple([0, 5] + [rnd.randint(6, 12) for _ in range(rnd.randint(1, 4))] + [1]) for _ in range(200)})
    items = [list(x) for x in items]
    ct = CompiledTrie.from_sequences(items)
    ct.index_items(items)
    assert ct.items_under[ct.item_paths[0][0]] == len(items)
    for frac in (0.0, 0.3, 0.9, 1.0):
        ex = rnd.sample(range(len(items)), int(frac * len(items)))
        bm = ct.excluded_bitmap([ex, []])
        assert bm.shape == (2, (ct.n_nodes + 31) // 32) and not bm[1].any()
        rest = Trie([it for i, it in enumerate(items) if i not in set(ex)])
        stack = [(0, [])]
        while stack:
            node, prefix = stack.pop()
            toks, kids = ct.children(node)
            alive = [(int(t), int(k)) for t, k in zip(toks, kids) if not (bm[0][k >> 5] >> (k & 31)) & 1]
            assert sorted(t for t, _ in alive) == sorted(rest.get(prefix)), prefix
            stack.extend((k, prefix + [t]) for t, k in alive)


def test_id_metrics_match_string_metrics():
    from openp5_amd import evaluate
    torch.manual_seed(0)
    B, K, S = 7, 5, 6
    gold = torch.randint(3, 9, (B, S - 2))
    gold = torch.cat([gold, torch.ones(B, 1, dtype=torch.long), torch.zeros(B, 1, dtype=torch.long)], 1)      # ... </s> <pad>
    seqs = torch.randint(3, 9, (B, K, S + 1))
    seqs[:, :, 0] = 0
    seqs[:, :, -3] = 1
    seqs[:, :, -2:] = 0
    for b in range(B):
        if b % 2 == 0:
            seqs[b, b % K, 1:1 + gold.shape[1]] = gold[b]
            seqs[b, b % K, 1 + gold.shape[1]:] = 0
    scores = torch.randn(B, K)
    scores[3, 1] = scores[3, 2]                                                   # a tie: stable order must be kept
    rel = evaluate.rel_results_ids(seqs.view(B * K, -1), scores.view(-1), gold, K)
    as_str = lambda row: " ".join(str(int(t)) for t in row if int(t) > 1)      # what batch_decode(skip_special_tokens) keeps
    rel_s = evaluate.rel_results([as_str(r) for r in seqs.view(B * K, -1)], [as_str(g) for g in gold], scores.view(-1).tolist(), K)
    assert rel.int().tolist() == rel_s and rel.any()
    metrics = ["hit@1", "hit@3", "ndcg@3", "ndcg@5"]
    a = evaluate.get_metrics_results_ids(rel, metrics)
    b = evaluate.get_metrics_results(rel_s, metrics)
    assert torch.allclose(a, torch.as_tensor(b), atol=1e-12)


def test_collator_solo_rows_equals_batch_of_one():
    from openp5_amd.collator import TestCollator
    from openp5_amd.tokenizer import build_offline_tokenizer
    tok = build_offline_tokenizer(2400)
    rows = [{"input": "Toy user_1 has interacted with Toy items item_1001 , item_1002 ; what next ?", "output": "Toy item_1003", "user_idx": 0},
            {"input": "Toy user_2 likes item_1004", "output": "Toy item_1001", "user_idx": 1},
            {"input": "What would Toy user_3 want after item_1002 , item_1005 ?", "output": "Toy item_1004", "user_idx": 2}]
    batched = TestCollator(tok, solo_rows=True)(rows)
    plain = TestCollator(tok)(rows)
    assert not torch.equal(batched[2], plain[2])                 # the reference's quirk: only the last COLUMN is zeroed
    for i, r in enumerate(rows):
        one = TestCollator(tok)([r])
        n = one[0].shape[1]
        assert torch.equal(batched[0][i, :n], one[0][0]) and torch.equal(batched[2][i, :n], one[2][0])
        assert int(batched[2][i, n:].sum()) == 0 and int(batched[1][i].sum()) == n


def test_filtered_id_metrics_match_string_metrics():
    import numpy as np
    from openp5_amd import evaluate
    rnd = random.Random(3)
    items = [[5, 6, 10 + i, 1] for i in range(12)]                       # token sequences "... </s>"
    text = {i: f"Toy item_{i}" for i in range(12)}
    seq2idx = {tuple(q): i for i, q in enumerate(items)}
    B, width, k, S = 5, 7, 3, 7
    seqs = np.zeros((B, width, S), dtype=np.int64)
    scores = np.zeros((B, width))
    gold_ids = np.zeros((B, 5), dtype=np.int64)
    gen_txt, gold_txt, positive_txt, positive_idx, id2user, user_idx = [], [], {}, [], {}, []
    for b in range(B):
        pick = rnd.sample(range(12), width)
        for r, it in enumerate(pick):
            seqs[b, r, 1:1 + len(items[it])] = items[it]
            gen_txt.append(text[it])
        scores[b] = [rnd.choice([-1.0, -2.0, -3.0]) for _ in range(width)]            # ties included
        pos = set(rnd.sample(range(12), 4))
        gold = rnd.choice([i for i in range(12) if i not in pos])
        gold_ids[b, :len(items[gold])] = items[gold]
        gold_txt.append(text[gold])
        id2user[b] = f"U{b}"
        user_idx.append(b)
        positive_txt[f"U{b}"] = {text[i] for i in pos}
        positive_idx.append(pos)
    a = evaluate.rel_results_filtered_ids(seqs, scores, gold_ids, positive_idx, seq2idx, k)
    ref = evaluate.rel_results_filtered(positive_txt, id2user, user_idx, width, gen_txt, gold_txt, scores.reshape(-1).tolist(), k)
    assert a == ref and any(sum(r) for r in a)


def test_compiled_trie_grafts_an_appended_trie():
    """generation_trie.py:19-21, 47-70: `Trie.append(trie, bos_token_id)`.  The CSR compilation grafts the appended trie wherever
    the main trie has a `bos` child; walking the compiled automaton must allow exactly what `Trie.get` allows after every
    prefix that `Trie.get` itself can generate."""
    import random
    from openp5_amd.trie import Trie, CompiledTrie
    rnd = random.Random(3)
    BOS = 7
    for trial in range(20):
        main_seqs, app_seqs = [], []
        for _ in range(rnd.randint(2, 8)):
            q = [0] + [rnd.randint(2, 12) for _ in range(rnd.randint(1, 4))]
            if rnd.random() < 0.7:
                q.append(BOS)                       # hand-over point
                if rnd.random() < 0.3:
                    q += [rnd.randint(2, 12), 1]    # (the main trie may also continue behind its own bos: never reachable)
            else:
                q.append(1)
            main_seqs.append(q)
        for _ in range(rnd.randint(1, 6)):
            app_seqs.append([rnd.randint(2, 12) for _ in range(rnd.randint(1, 3))] + [1])
        if trial % 4 == 0:
            app_seqs.append([BOS, 3, 1])            # the appended trie's root may have the bos token too
        t, a = Trie(main_seqs), Trie(app_seqs)
        t.append(a, BOS)
        ct = CompiledTrie.from_trie(t)
        assert ct.grafted and ct.max_children >= 1
        # breadth-first over every prefix the host trie can produce
        frontier, seen = [([], 0)], 0
        while frontier:
            nxt = []
            for prefix, node in frontier:
                toks, kids = ct.children(node)
                want = t.get(prefix)
                assert sorted(set(int(x) for x in toks)) == sorted(set(want)), (trial, prefix, list(toks), want)
                assert list(toks) == sorted(toks)
                for tok, kid in zip(toks, kids):
                    nxt.append((prefix + [int(tok)], int(kid)))
                seen += 1
            frontier = nxt
            assert seen < 20000
        assert ct.max_depth >= max(len(q) for q in main_seqs)
    with pytest.raises(NotImplementedError):
        ct.index_items(main_seqs)
