"""-m gpu: dataset-level evaluation parity -- north_star: "generated item-ID sequences match the HF-CPU reference within fp32
logit tolerance (ranked Hit@k identical)".  A T5-small-dims model is trained for a few epochs on a synthetic dataset through the
real pipeline (datasets -> sampler -> collator -> runner, bf16 engine), then every test user of both tasks is evaluated three
ways with the SAME weights: bf16 engine, fp32 engine, fp32 CPU oracle (restated HF beam search + Python trie callbacks)."""
import pytest
import torch

from oracle import t5_oracle as O
from tests import cases

pytestmark = pytest.mark.gpu

FP32_TIE_TOL = 1e-4   # fp32 engine vs oracle: two items whose oracle scores differ by less than the fp32 score tolerance of the
                      # generation tests (1e-4) may swap places (measured: 2 of 240 users, score gaps <= 1.2e-5)
BF16_SCORE_TOL = 0.02   # bf16 engine: ceiling on the largest |score - oracle score| of an item both list (measured 0.003 .. 0.016,
                        # depending on the weights the few training epochs produce)
TIE_TOL = 2.0 * BF16_SCORE_TOL   # FIXED decision margin of the ORACLE below which the bf16 engine may decide differently: score errors
                                 # below BF16_SCORE_TOL per score can flip decisions whose margin is at most twice that.  (Round 2
                                 # scaled this with the error measured in the same run, so a regression widened its own excuse.)


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
    margins = []
    r_or = cases.collect_rankings(runner, cases.oracle_gen_fn({k: sd[k] for k in O.param_shapes(ocfg)}, ocfg, margins), K)
    m_bf16, m_fp32, m_or = cases.rankings_metrics(r_bf16), cases.rankings_metrics(r_fp32), cases.rankings_metrics(r_or)
    c32 = cases.compare_rankings(r_fp32, r_or, tie_tol=FP32_TIE_TOL)
    c16 = cases.compare_rankings(r_bf16, r_or, tie_tol=TIE_TOL)
    print("[dataset] oracle metrics", m_or)
    print("[dataset] bf16 metrics  ", m_bf16)
    print("[dataset] fp32 engine vs oracle", {k: v for k, v in c32.items()})
    print("[dataset] bf16 engine vs oracle", {k: v for k, v in c16.items()})
    assert sum(len(u) for u in r_or) >= 200 and any(v > 0 for m in m_or for v in m.values())
    # fp32 engine: every user's ranked list identical to the oracle's up to swaps of items the ORACLE scores within 1e-4 of each
    # other, the gold item at the same rank for every user, hence every Hit@k / NDCG@k identical
    assert c32["identical_up_to_ties"] == c32["users"] and c32["max_score_diff"] <= 1e-4, c32
    assert c32["identical_lists"] >= 0.98 * c32["users"], c32
    assert c32["same_gold_rank"] == c32["users"] and m_fp32 == m_or
    # ---- teacher-forced check, EVERY user, EVERY returned hypothesis (cases.teacher_forced_check): the oracle scores the very token
    # sequences an engine returned (O.sequence_scores).  This does not depend on the two searches having decided alike, so it is not
    # vacuous on a model whose own decision margins are small: (a) each returned score equals the oracle's score of that sequence
    # within the mode's tolerance, (b) the returned order is the oracle's order of those sequences up to the tie tolerance.
    params_o = {k: sd[k] for k in O.param_shapes(ocfg)}
    tf32 = cases.teacher_forced_check(runner, params_o, ocfg, r_fp32, K, 1e-4, FP32_TIE_TOL, r_or)
    tf16 = cases.teacher_forced_check(runner, params_o, ocfg, r_bf16, K, BF16_SCORE_TOL, TIE_TOL, r_or)
    print("[dataset] teacher-forced, fp32 engine:", {k: v for k, v in tf32.items() if k != "missed"}, "max missed", max(tf32["missed"]))
    print("[dataset] teacher-forced, bf16 engine:", {k: v for k, v in tf16.items() if k != "missed"}, "missed > TIE_TOL:",
          sum(1 for x in tf16["missed"] if x > TIE_TOL), "max", max(tf16["missed"]))
    assert tf32["users"] == c32["users"] and tf32["score_viol"] == 0 and tf32["order_viol"] == 0 and max(tf32["missed"]) <= FP32_TIE_TOL, tf32
    assert tf16["score_viol"] == 0 and tf16["order_viol"] == 0, {k: v for k, v in tf16.items() if k != "missed"}
    # (c) EVERY difference between the bf16 list and the oracle's list must be explained by a tie: either the lists are equal up to swaps of
    # items the oracle scores within TIE_TOL of each other, or they differ at the boundary of the list -- what the oracle lists and the
    # search does not is within TIE_TOL of the search's K-th item, what the search lists and the oracle does not is within TIE_TOL of the
    # oracle's K-th item (all in oracle scores).  What remains unexplained is a near-tie between two PREFIXES whose completions score very
    # differently (the one genuine fragility of beam search): observed 0 and 1 of 240 users; at most 2 % may be.
    flat16, flat_or = [u for us in r_bf16 for u in us], [u for us in r_or for u in us]
    unexplained = []
    for i, ((_, ra, _), (_, ro, so)) in enumerate(zip(flat16, flat_or)):
        tie_swap = cases.lists_equal_up_to_ties(list(ra), list(ro), list(so), TIE_TOL)
        boundary = tf16["missed"][i] <= TIE_TOL and tf16["extra"][i] <= TIE_TOL
        if not (tie_swap or boundary):
            unexplained.append((i, round(tf16["missed"][i], 4), round(tf16["extra"][i], 4)))
    print(f"[dataset] bf16: differences not explained by ties: {len(unexplained)} of {len(flat16)} users {unexplained[:6]}")
    assert len(unexplained) <= 0.02 * len(flat16), unexplained
    # bf16 engine.  Its scores are within BF16_SCORE_TOL of the oracle's; a beam search is a sequence of discrete decisions, so
    # it must reproduce the oracle exactly wherever the oracle took every decision by a margin larger than TIE_TOL
    # (fixed: 2 x the score-error ceiling) and may differ only where the oracle itself was that close to deciding otherwise:
    #   * list-robust users   -> identical ranked lists;
    #   * metric-robust users -> gold item at the same rank, i.e. identical Hit@5/10, NDCG@5/10 contributions;
    #   * the rest is the tie report (printed), and the dataset-level metrics may move by at most those users.
    assert c16["max_score_diff"] <= BF16_SCORE_TOL, c16
    # Floors over ALL users (the raw counts of bit-identical lists, of lists identical up to tie swaps and of identical top-10 sets are
    # printed, not gated: with the oracle's median gap between consecutive final scores at 0.006 they count near-ties at the tail of the
    # list -- six training trajectories of this test gave 158 .. 203 identical lists, 227 .. 240 identical up to swaps, the top-10 set
    # differing for 0 .. 13 users, always with the gold item at the same rank for >= 235 users and the top-5 SET identical for all 240;
    # what every returned list must satisfy is the teacher-forced check and the "every difference is a tie" assertion above.  Those
    # trajectories started from different embeddings -- `random_initialization` draws from torch's device generator, which make_pipeline
    # did not seed; it does now, and the training itself is bit-reproducible, so a given build gives ONE trajectory):
    print(f"[dataset] bf16: {c16['identical_lists']}/{c16['users']} bit-identical lists, {c16['identical_up_to_ties']} identical up to oracle ties <= {TIE_TOL}, "
          f"same top-10 set {c16['same_topk_set'][10]}, same top-5 set {c16['same_topk_set'][5]}, same gold rank {c16['same_gold_rank']}")
    assert c16["same_topk_set"][5] >= 0.95 * c16["users"], c16
    assert c16["same_gold_rank"] >= 0.95 * c16["users"], c16
    for mb, mo in zip(m_bf16, m_or):
        assert abs(mb["hit@5"] - mo["hit@5"]) <= 2.0 / (c16["users"] / len(m_or)) + 1e-12, (mb, mo)
        assert abs(mb["hit@10"] - mo["hit@10"]) <= 2.0 / (c16["users"] / len(m_or)) + 1e-12, (mb, mo)
    rob = cases.robust_users(r_or, margins, TIE_TOL)
    flat_b, flat_o = [u for us in r_bf16 for u in us], [u for us in r_or for u in us]
    n_list = n_metric = n_fragile_moved = 0
    for (list_ok, metric_ok), (g, ra, sa), (_, ro, so), (set_m, gaps) in zip(rob, flat_b, flat_o, margins):
        ka, ko = (ra.index(g) if g in ra else -1), (ro.index(g) if g in ro else -1)
        if list_ok:
            n_list += 1
            assert ra == ro, ("list-robust user differs", set_m, gaps, ra, ro)
        if metric_ok:
            n_metric += 1
            assert ka == ko, ("metric-robust user: gold rank moved", set_m, gaps, ka, ko)
        elif ka != ko:
            n_fragile_moved += 1
            print(f"[dataset] tie report: gold rank {ko} (oracle) vs {ka} (bf16); oracle's smallest set margin {set_m:.4f}, "
                  f"final-score gaps around the gold item {[round(x, 4) for x in gaps[max(0, ko - 1):ko + 1]] if ko >= 0 else '-'}")
    n = len(rob)
    print(f"[dataset] bf16: {n_list}/{n} users list-robust (all identical), {n_metric}/{n} metric-robust (gold rank identical), "
          f"{n - n_metric} fragile of which {n_fragile_moved} moved")
    sm = sorted(m[0] for m in margins)
    print(f"[dataset] TIE_TOL {TIE_TOL:.4f}; oracle set-margin quantiles 10/50/90%: {sm[n // 10]:.4f} {sm[n // 2]:.4f} {sm[9 * n // 10]:.4f}")
    # (the oracle's OWN decision margins on this barely-trained model are small -- median 0.025, 90 % below 0.06 -- so most users are
    # fragile at any tolerance a bf16 score error of 0.005 .. 0.016 allows; the floors asserted above are what holds for ALL users)
    # (how many users are robust at TIE_TOL depends on the trained weights: 18 of 240 on one trajectory of this test, 0 on another --
    #  the per-user exactness above is asserted for whoever is robust; the floors over ALL users are the gate that always applies)
    print(f"[dataset] robust population at TIE_TOL: {n_metric}/{n}")
    for mb, mo in zip(m_bf16, m_or):      # dataset-level metrics: equal up to the fragile users that moved
        for k in mo:
            assert abs(mb[k] - mo[k]) <= n_fragile_moved / (n / len(m_or)) + 1e-12, (k, mb[k], mo[k])
    if n_fragile_moved == 0:
        assert m_bf16 == m_or, (m_bf16, m_or)
