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
