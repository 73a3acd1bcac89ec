"""-m gpu: dataset-level evaluation parity -- north_star: "generated item-ID sequences match the HF-CPU reference within fp32
logit tolerance (ranked Hit@k identical)".  A T5-small-dims model is trained for a few epochs on a synthetic dataset through the
real pipeline (datasets -> sampler -> collator -> runner, bf16 engine), then every test user of both tasks is evaluated three
ways with the SAME weights: bf16 engine, fp32 engine, fp32 CPU oracle (restated HF beam search + Python trie callbacks)."""
import pytest
import torch

from oracle import t5_oracle as O
from tests import cases

pytestmark = pytest.mark.gpu

TIE_TOL = 0.02     # log-prob score gap below which two items count as tied for the bf16 engine (5x the largest score
                   # difference measured between the bf16 engine and the oracle on this set-up, 3.7e-3)


def test_dataset_level_hit_ndcg_equal_oracle(hip, tmp_path):
    runner, model, tok, args = cases.make_pipeline(hip, str(tmp_path), "bf16", flags=["--epochs", "6", "--lr", "1e-3"])
    losses = runner.train()
    assert losses[-1] < 0.7 * losses[0], losses
    model.eval()
    K = 10
    r_bf16 = cases.collect_rankings(runner, cases.engine_gen_fn(model), K)
    sd = {k: v.detach().cpu().float().clone() for k, v in model.state_dict().items()}
    from openp5_amd.model import P5T5Native
    m32 = P5T5Native(model.config, dtype="fp32", backend=hip, seed=1)
    m32.load_state_dict(sd, strict=False)
    m32.eval()
    r_fp32 = cases.collect_rankings(runner, cases.engine_gen_fn(m32), K)
    ocfg = O.T5Cfg.named("t5-small", dropout=0.0, vocab_size=model.config.vocab_size)
    r_or = cases.collect_rankings(runner, cases.oracle_gen_fn({k: sd[k] for k in O.param_shapes(ocfg)}, ocfg), K)
    m_bf16, m_fp32, m_or = cases.rankings_metrics(r_bf16), cases.rankings_metrics(r_fp32), cases.rankings_metrics(r_or)
    c32, c16 = cases.compare_rankings(r_fp32, r_or), cases.compare_rankings(r_bf16, r_or)
    print("[dataset] oracle metrics", m_or)
    print("[dataset] fp32 engine vs oracle", {k: v for k, v in c32.items()})
    print("[dataset] bf16 engine vs oracle", {k: v for k, v in c16.items()})
    assert sum(len(u) for u in r_or) >= 200 and any(v > 0 for m in m_or for v in m.values())
    # fp32 engine: every user's ranked list and every metric identical to the oracle
    assert c32["identical_lists"] == c32["users"] and c32["max_score_diff"] <= 1e-4
    assert m_fp32 == m_or
    # bf16 engine: Hit@5/10 and NDCG@5/10 equal to the oracle's; a user whose gold item sits at a different rank must be a
    # near-tie in the ORACLE's own scores (tie report), and only such users may move a metric
    tie_only = True
    for la, lo in zip(r_bf16, r_or):
        for (g, ra, sa), (_, ro, so) in zip(la, lo):
            ka = ra.index(g) if g in ra else -1
            ko = ro.index(g) if g in ro else -1
            if ka == ko:
                continue
            lo_, hi_ = sorted((ka if ka >= 0 else K - 1, ko if ko >= 0 else K - 1))
            gap = abs(so[lo_] - so[min(hi_, K - 1)])
            print(f"[dataset] tie report: gold rank {ko} (oracle) vs {ka} (bf16), oracle score gap {gap:.4f}")
            tie_only = tie_only and gap <= TIE_TOL
    assert tie_only, "bf16 ranking moved a gold item across a score gap larger than the tie tolerance"
    if c16["same_gold_rank"] == c16["users"]:
        assert m_bf16 == m_or, (m_bf16, m_or)
    assert c16["same_topk_set"][10] >= 0.95 * c16["users"] and c16["max_score_diff"] <= 0.05
