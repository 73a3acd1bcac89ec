"""Generates tests/golden/*.pt from STOCK HuggingFace T5 (installed transformers, eager attention, fp32, CPU)
run in the build container -- the third-party code that holds this path's arithmetic.
The fixtures pin (a) the restated oracle (tests/test_oracle.py) and (b) the HIP path (-m gpu tests) on the GPU
box, where /root/reference and HF-vs-oracle cross-checks are not assumed.
NOTE: the installed transformers is 5.15, the reference pins 4.26.0 (src/src_t5/environment_t5.txt:2), whose source is not on
this box; SURVEY.md 8(c) lists the known behavioural deltas (vectorised beam search, pad fill after </s>).

    python tests/golden/make_golden.py
"""
import os
import random
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import t5_oracle as O          # noqa: E402
from oracle.hf_ref import build_hf, hf_forward_nll, hf_generate   # noqa: E402
from openp5_amd.trie import Trie, prefix_allowed_tokens_fn        # noqa: E402


def make_inputs(cfg, B, L, T, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, cfg.vocab_size, (B, L), generator=g)
    mask = torch.ones(B, L, dtype=torch.long)
    for b in range(1, B):
        n = int(torch.randint(L // 2, L + 1, (1,), generator=g))
        mask[b, n:] = 0
        ids[b, n:] = 0
    ww = torch.cumsum((torch.rand(B, L, generator=g) < 0.4).long(), 1) * mask
    labels = torch.randint(3, cfg.vocab_size, (B, T), generator=g)
    out_attn = torch.ones(B, T, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(2, T + 1, (1,), generator=g))
        labels[b, n - 1] = cfg.eos_id
        labels[b, n:] = 0
        out_attn[b, n:] = 0
    return ids, ww, mask, labels, out_attn


def make_items(n_items, seed, lo=7, hi=40):
    rnd = random.Random(seed)
    items = set()
    while len(items) < n_items:
        n = rnd.randint(2, 4)
        items.add(tuple([0, 5, 6] + [rnd.randint(lo, hi) for _ in range(n)] + [1]))
    return sorted(list(x) for x in items)


def main():
    torch.manual_seed(0)
    for name, cfgname, kw, (B, L, T), K, ML in [
        ("tiny_relu", "tiny", dict(), (3, 20, 6), 5, 12),
        ("tiny_gated", "tiny", dict(ff_act="gated-gelu"), (2, 17, 5), 4, 10),
        # full T5-small dims (d=512, F=2048, H=8, 6+6 layers, V=32100): the oracle and the HIP path are pinned to HF at the
        # size BASELINE.json's headline config uses, not only at toy dims (parameters are regenerated from the seed)
        ("t5small_relu", "t5-small", dict(), (2, 48, 8), 6, 12),
    ]:
        cfg = O.T5Cfg.named(cfgname, dropout=0.0, **kw)
        P = O.init_params(cfg, 11)
        m, wwe = build_hf(cfg, P)
        ids, ww, mask, labels, out_attn = make_inputs(cfg, B, L, T, 5)
        for p in m.parameters():
            p.requires_grad_(True)
        wwe.weight.requires_grad_(True)
        nll, logits = hf_forward_nll(m, wwe, ids, ww, mask, labels)
        loss = O.runner_loss(nll, out_attn)
        loss.backward()
        grads = {k: v.grad.detach().clone() for k, v in m.named_parameters() if v.grad is not None}
        grads["encoder.whole_word_embeddings.weight"] = wwe.weight.grad.detach().clone()
        items = make_items(40, 3) if cfgname == "tiny" else make_items(200, 3, lo=3000, hi=3200)
        fn = prefix_allowed_tokens_fn(Trie(items))
        seqs, scores = hf_generate(m, wwe, ids, ww, mask, fn, K, ML)
        fx = dict(cfg=cfg.__dict__, params_seed=11, input_ids=ids, whole_word_ids=ww, attention_mask=mask, labels=labels,
                  output_attention=out_attn, nll=nll.detach(), loss=loss.detach(), logits_row0=logits[0, 0].detach().clone(),
                  grad_norms={k: float(v.norm()) for k, v in grads.items()},
                  grad_shared=grads["shared.weight"][:16].clone(),
                  grad_enc_rel=grads["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"].clone(),
                  items=items, num_beams=K, max_length=ML, sequences=seqs, sequences_scores=scores,
                  transformers_version=__import__("transformers").__version__)
        torch.save(fx, os.path.join(HERE, name + ".pt"))
        print(name, "nll[:4]", nll[:4].tolist(), "loss", float(loss), "seq", tuple(seqs.shape))


if __name__ == "__main__":
    main()
