"""not-gpu: pins the oracle (oracle/t5_oracle.py) -- against the committed golden fixtures (made by stock HF T5, see
tests/golden/make_golden.py) and, when transformers is importable, against stock HF T5 run live on fresh inputs."""
import os
import random

import pytest
import torch

from oracle import t5_oracle as O
from openp5_amd.trie import Trie, prefix_allowed_tokens_fn
from tests import cases


@pytest.mark.parametrize("name", ["tiny_relu", "tiny_gated", "t5small_relu"])
def test_oracle_vs_golden(name):
    """fixtures come from stock HF T5 (transformers 5.15 installed here; the reference pins 4.26.0)."""
    fx = torch.load(os.path.join(cases.GOLDEN, name + ".pt"), weights_only=False)
    cfg = O.T5Cfg(**fx["cfg"])
    P = {k: v.requires_grad_(True) for k, v in O.init_params(cfg, fx["params_seed"]).items()}
    nll = O.p5_forward_nll(P, cfg, fx["input_ids"], fx["whole_word_ids"], fx["attention_mask"], fx["labels"])
    assert (nll - fx["nll"]).abs().max() < 1e-5
    loss = O.runner_loss(nll, fx["output_attention"])
    loss.backward()
    for k, ref in fx["grad_norms"].items():
        if k in P:
            assert abs(float(P[k].grad.norm()) - ref) <= 1e-4 * max(1.0, ref), k
    trie = Trie(fx["items"])
    with torch.no_grad():
        s, sc = O.beam_search({k: v.detach() for k, v in P.items()}, cfg, fx["input_ids"], fx["whole_word_ids"], fx["attention_mask"],
                              lambda b, p: trie.get(p.tolist()), fx["num_beams"], fx["max_length"])
    cases.compare_generation(s, sc, fx["sequences"], fx["sequences_scores"], 1e-5)


def test_oracle_vs_hf_live():
    pytest.importorskip("transformers")
    from oracle.hf_ref import build_hf, hf_forward_nll, hf_generate
    cfg = O.T5Cfg.named("tiny", dropout=0.0, num_layers=3, num_decoder_layers=2)
    P = O.init_params(cfg, 21)
    m, wwe = build_hf(cfg, P)
    ids, ww, mask, labels, out_attn = cases.synth_batch(cfg, 4, 33, 7, 9)
    with torch.no_grad():
        nll_hf, _ = hf_forward_nll(m, wwe, ids, ww, mask, labels)
        nll = O.p5_forward_nll(P, cfg, ids, ww, mask, labels)
    assert (nll - nll_hf).abs().max() < 1e-5
    trie = Trie(cases.make_items(60, 4, hi=50))
    fn = lambda b, s: trie.get(s.tolist())   # noqa: E731
    with torch.no_grad():
        s1, sc1 = O.beam_search(P, cfg, ids, ww, mask, fn, 6, 11)
    s2, sc2 = hf_generate(m, wwe, ids, ww, mask, fn, 6, 11)
    cases.compare_generation(s1, sc1, s2, sc2, 1e-5)


def test_oracle_vs_hf_live_t5_small_dims():
    """the same live cross-check at full T5-small dims and V=32100 (transformers 5.15; the reference pins 4.26.0)."""
    pytest.importorskip("transformers")
    from oracle.hf_ref import build_hf, hf_forward_nll, hf_generate
    cfg = O.T5Cfg.named("t5-small", dropout=0.0)
    P = O.init_params(cfg, 23)
    m, wwe = build_hf(cfg, P)
    ids, ww, mask, labels, out_attn = cases.synth_batch(cfg, 2, 40, 7, 9)
    with torch.no_grad():
        nll_hf, _ = hf_forward_nll(m, wwe, ids, ww, mask, labels)
        nll = O.p5_forward_nll(P, cfg, ids, ww, mask, labels)
    assert (nll - nll_hf).abs().max() < 2e-5
    trie = Trie(cases.make_items(120, 4, lo=3000, hi=3100))
    fn = lambda b, s: trie.get(s.tolist())   # noqa: E731
    with torch.no_grad():
        s1, sc1 = O.beam_search(P, cfg, ids, ww, mask, fn, 5, 10)
    s2, sc2 = hf_generate(m, wwe, ids, ww, mask, fn, 5, 10)
    cases.compare_generation(s1, sc1, s2, sc2, 2e-5)


def test_bucket_lut_matches_oracle():
    from openp5_amd.model import relative_position_bucket_lut
    for bidir in (True, False):
        a = relative_position_bucket_lut(512, bidir, 32, 128)
        b = O.bucket_lut(513, bidir).to(torch.int32)
        assert torch.equal(a, b)
    # spot values of HF modeling_t5.py:217-262
    lut = relative_position_bucket_lut(512, True, 32, 128)
    assert lut[512 + 0] == 0 and lut[512 + 1] == 17 and lut[512 - 1] == 1 and lut[512 + 7] == 23 and lut[512 + 127] == 31 and lut[512 - 200] == 15


def test_dropout_mask_statistics():
    keep = O.dropout_keep_mask(1234, O.site_id(0, 2, 5), 200000, 0.1)
    assert abs(float(keep.float().mean()) - 0.9) < 0.005
    k2 = O.dropout_keep_mask(1235, O.site_id(0, 2, 5), 200000, 0.1)
    assert 0.75 < float((keep == k2).float().mean()) < 0.9      # ~0.82 for independent masks


def test_optimizer_matches_published_426_fixture():
    """O.clip_coef / O.linear_schedule_lr / O.adamw_hf_step (the torch-op restatement the GPU trajectory tests compare with) against
    tests/golden/adamw_426.json: the PUBLISHED transformers-4.26 AdamW.step, linear-warmup lambda and torch-1.8.1 clip_grad_norm_ run in
    fp64 scalar Python by tests/golden/make_adamw_426.py (three parameters, five steps across the warm-up boundary, clipped and unclipped
    steps).  Run in float64 the two must agree to rounding."""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "adamw_426.json")))
    h = gold["hyper"]
    names = list(gold["shapes"])
    p = {k: torch.tensor(gold["p0"][k], dtype=torch.float64) for k in names}
    m = {k: torch.zeros_like(p[k]) for k in names}
    v = {k: torch.zeros_like(p[k]) for k in names}
    for t, st in enumerate(gold["steps"], start=1):
        g = {k: torch.tensor(st["grad"][k], dtype=torch.float64) for k in names}
        total, coef = O.clip_coef([g[k] for k in names], h["max_norm"])
        assert abs(total - st["total_norm"]) <= 1e-12 * st["total_norm"] and (coef < 1.0) == st["clipped"]
        lr = O.linear_schedule_lr(h["lr"], t - 1, h["warmup_steps"], h["total_steps"])
        assert abs(lr - st["lr"]) <= 1e-15
        for k in names:
            O.adamw_hf_step(p[k], g[k] * coef, m[k], v[k], t, lr, h["beta1"], h["beta2"], h["eps"], h["weight_decay"])
            for name, got in (("p", p), ("m", m), ("v", v)):
                ref = torch.tensor(st[name][k], dtype=torch.float64)
                assert (got[k] - ref).abs().max() <= 1e-12 * max(1.0, float(ref.abs().max())), (t, k, name)


def test_optimizer_restatement():
    """clip + HF-AdamW + linear warmup against torch.optim reference arithmetic written out by hand."""
    assert O.linear_schedule_lr(1e-3, 0, 10, 100) == 0.0
    assert abs(O.linear_schedule_lr(1e-3, 5, 10, 100) - 5e-4) < 1e-12
    assert abs(O.linear_schedule_lr(1e-3, 55, 10, 100) - 1e-3 * 45 / 90) < 1e-12
    p = torch.tensor([1.0, -2.0]); g = torch.tensor([0.5, 0.25]); m = torch.zeros(2); v = torch.zeros(2)
    O.adamw_hf_step(p, g, m, v, 1, 1e-2)
    # t=1: m = 0.1 g, v = 0.001 g^2 -> update = lr * sqrt(0.001)/0.1 * m/(sqrt(v)+eps) ~= lr * sign(g); then decay lr*wd*p
    exp = torch.tensor([1.0, -2.0]) - 1e-2 * torch.sign(g)
    exp = exp - 1e-2 * 0.01 * exp
    assert torch.allclose(p, exp, atol=1e-6)
