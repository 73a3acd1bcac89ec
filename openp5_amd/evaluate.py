"""Ranking metrics of the evaluation loop -- same functions and conventions as
/root/reference/src/src_t5/utils/evaluate.py (:6-92): per user the K generated items are sorted by sequence score,
relevance = exact string match with the gold item, and Hit@k / NDCG@k are returned as SUMS over users (the runner
divides by the all-reduced user count, DistributedRunner.py:389-395).  Leave-one-out => ideal DCG is 1."""
import math

import numpy as np


def _ranked(predictions, scores, lo, hi):
    pairs = list(zip(predictions[lo:hi], [float(s) for s in scores[lo:hi]]))
    pairs.sort(key=lambda x: x[1], reverse=True)     # stable, like sorted() in the reference
    return [p for p, _ in pairs]


def rel_results(predictions, targets, scores, k):
    out = []
    for b, gold in enumerate(targets):
        out.append([1 if p == gold else 0 for p in _ranked(predictions, scores, b * k, (b + 1) * k)])
    return out


def rel_results_filtered(user_positive, id2user, user_idx, return_num, predictions, targets, scores, k):
    out = []
    for b, gold in enumerate(targets):
        positive = user_positive[id2user[int(user_idx[b])]]
        row = []
        for p in _ranked(predictions, scores, b * return_num, (b + 1) * return_num):
            if p in positive:
                continue
            row.append(1 if p == gold else 0)
            if len(row) >= k:
                break
        out.append(row)
    return out


def hit_at_k(relevance, k):
    return float(sum(1 for row in relevance if sum(row[:k]) > 0))


def ndcg_at_k(relevance, k):
    total = 0.0
    for row in relevance:
        total += sum(r / math.log(i + 2, 2) for i, r in enumerate(row[:k]))
    return total


def get_metrics_results(rel, metrics):
    res = []
    for m in metrics:
        name, k = m.lower().split("@")
        if name.startswith("hit"):
            res.append(hit_at_k(rel, int(k)))
        elif name.startswith("ndcg"):
            res.append(ndcg_at_k(rel, int(k)))
    return np.array(res)


# ---- token-id form of the same metrics: no batch_decode, no strings, no per-batch D2H.  Under the item trie
# a generated sequence IS an item's token sequence, so string equality after decoding (DistributedRunner.py:376-387) and
# token equality agree; the K hypotheses are ranked by score with a stable sort exactly like `_ranked`. ----
def rel_results_ids(sequences, scores, gold_ids, k, pad_id=0):
    """sequences [B*k, S] int (column 0 = decoder start), scores [B*k], gold_ids [B, T] -> bool [B, k] on the same device."""
    import torch
    import torch.nn.functional as F
    B = gold_ids.shape[0]
    gen = sequences.reshape(B, k, -1)[:, :, 1:]
    width = max(gen.shape[-1], gold_ids.shape[-1])
    gen = F.pad(gen, (0, width - gen.shape[-1]), value=pad_id)
    gold = F.pad(gold_ids, (0, width - gold_ids.shape[-1]), value=pad_id)
    rel = (gen == gold[:, None, :]).all(-1)
    order = torch.sort(scores.reshape(B, k), dim=1, descending=True, stable=True).indices
    return torch.gather(rel, 1, order)


def get_metrics_results_ids(rel, metrics):
    """float64 [len(metrics)] sums over users, left on rel's device."""
    import torch
    relf = rel.to(torch.float64)
    disc = 1.0 / torch.log2(torch.arange(rel.shape[1], device=rel.device, dtype=torch.float64) + 2.0)
    out = []
    for m in metrics:
        name, k = m.lower().split("@")
        k = int(k)
        if name.startswith("hit"):
            out.append((relf[:, :k].sum(1) > 0).to(torch.float64).sum())
        elif name.startswith("ndcg"):
            out.append((relf[:, :k] * disc[:k]).sum())
    return torch.stack(out)


def rel_results_filtered_ids(sequences, scores, gold_ids, positive_idx, seq2idx, k):
    """ID form of `rel_results_filtered` (DistributedRunner.py:238-262): the `width` hypotheses of every user, ranked by score
    (stable), minus those that are items of the user's history, cut to k; relevance = same item as the gold one.
    sequences [B, width, S] / gold_ids [B, T] are integer arrays (numpy), column 0 of `sequences` is the decoder start;
    seq2idx maps an item's token tuple (ending with </s>) to its index, positive_idx[b] is a set of such indices."""
    def key(row):
        n = len(row)
        while n > 0 and row[n - 1] == 0:
            n -= 1
        return tuple(int(t) for t in row[:n])
    out = []
    for b in range(len(gold_ids)):
        gold = seq2idx.get(key(gold_ids[b]), -2)
        order = sorted(range(len(scores[b])), key=lambda r: -float(scores[b][r]))      # stable, like sorted() in the reference
        row = []
        for r in order:
            idx = seq2idx.get(key(sequences[b][r][1:]), -1)
            if idx in positive_idx[b]:
                continue
            row.append(1 if idx == gold else 0)
            if len(row) >= k:
                break
        out.append(row)
    return out
