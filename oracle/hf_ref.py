"""Stock-HuggingFace T5 run of the same path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Builds `T5ForConditionalGeneration` (installed transformers, eager attention, fp32) carrying the same
parameters as `oracle.t5_oracle`, feeds the encoder `inputs_embeds = shared(ids) + whole_word(ww)`
(P5_T5.py:94-100) and reproduces P5_T5.forward's `reduction="none"` CE (P5_T5.py:368-369) and the
runner's generate() call (DistributedRunner.py:361-371).  Used to (a) pin the restated oracle and
(b) generate tests/golden/*.pt.  SURVEY.md 8(c) explains why the reference's own P5_T5.py cannot be
imported under the installed transformers.
"""
from __future__ import annotations

from typing import Callable, Dict, List

import torch

from .t5_oracle import T5Cfg


def build_hf(cfg: T5Cfg, params: Dict[str, torch.Tensor]):
    from transformers import T5Config, T5ForConditionalGeneration

    assert cfg.num_layers == cfg.num_decoder_layers or True
    hc = T5Config(
        vocab_size=cfg.vocab_size, d_model=cfg.d_model, d_kv=cfg.d_kv, d_ff=cfg.d_ff,
        num_layers=cfg.num_layers, num_decoder_layers=cfg.num_decoder_layers, num_heads=cfg.num_heads,
        relative_attention_num_buckets=cfg.rel_buckets, relative_attention_max_distance=cfg.rel_max_distance,
        dropout_rate=cfg.dropout, layer_norm_epsilon=cfg.eps,
        feed_forward_proj="relu" if cfg.ff_act == "relu" else "gated-gelu",
        decoder_start_token_id=cfg.pad_id, pad_token_id=cfg.pad_id, eos_token_id=cfg.eos_id,
    )
    m = T5ForConditionalGeneration._from_config(hc, attn_implementation="eager")
    sd = {k: v.clone().float() for k, v in params.items() if k != "encoder.whole_word_embeddings.weight"}
    sd["encoder.embed_tokens.weight"] = sd["shared.weight"]
    sd["decoder.embed_tokens.weight"] = sd["shared.weight"]
    sd["lm_head.weight"] = sd["shared.weight"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    m.tie_weights()
    wwe = torch.nn.Embedding(cfg.whole_word_size, cfg.d_model)
    wwe.weight.data.copy_(params["encoder.whole_word_embeddings.weight"].float())
    m.eval()
    return m, wwe


def hf_forward_nll(m, wwe, input_ids, whole_word_ids, attention_mask, labels):
    emb = m.shared(input_ids) + wwe(whole_word_ids)
    out = m(inputs_embeds=emb, attention_mask=attention_mask, labels=labels, return_dict=True)
    logits = out.logits
    nll = torch.nn.functional.cross_entropy(logits.view(-1, logits.shape[-1]), labels.view(-1),
                                            ignore_index=-100, reduction="none")
    return nll, logits


def hf_generate(m, wwe, input_ids, whole_word_ids, attention_mask, allowed_fn: Callable[[int, torch.Tensor], List[int]],
                num_beams: int, max_length: int):
    emb = m.shared(input_ids) + wwe(whole_word_ids)
    with torch.no_grad():
        enc = m.encoder(inputs_embeds=emb, attention_mask=attention_mask, return_dict=True)

        def shim(b, s):
            r = allowed_fn(b, s)
            return r if len(r) else [0]

        out = m.generate(encoder_outputs=enc, attention_mask=attention_mask, max_length=max_length,
                         prefix_allowed_tokens_fn=shim, num_beams=num_beams, num_return_sequences=num_beams,
                         output_scores=True, return_dict_in_generate=True, do_sample=False,
                         early_stopping=False, length_penalty=1.0)
    return out["sequences"], out["sequences_scores"]
