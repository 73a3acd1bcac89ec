"""CPU oracle for the OpenP5 `src_t5` hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (openp5_amd/) never imports it and has no CPU fallback.

What it restates (plain torch ops on CPU, fp32 or fp64, no HuggingFace import):

* P5_T5.forward                       /root/reference/src/src_t5/model/P5_T5.py:275-386
    - JointEncoder embed + whole-word  P5_T5.py:94-100,125
    - T5Stack / T5Block / T5Attention  transformers==4.26.0 (pinned in src/src_t5/environment_t5.txt:2;
      T5LayerNorm / T5LayerFF          third-party, not vendored).  Restated from the installed
                                       transformers 5.15.0 models/t5/modeling_t5.py:50-72 (norm),
                                       :75-141 (FFN), :144-173,:217-369 (attention + buckets)
    - tied lm_head, d^-0.5 scale, CE(reduction=none)   P5_T5.py:352-369
* runner masked loss                   src/src_t5/runner/DistributedRunner.py:72-77
* HF AdamW + linear warmup schedule    src/src_t5/runner/SingleRunner.py:178-219 (transformers.AdamW 4.26
                                       semantics)
* constrained beam search              DistributedRunner.py:361-371 -> HF GenerationMixin beam search,
                                       restated from transformers 5.15.0 generation/utils.py:3008-3560
                                       + PrefixConstrainedLogitsProcessor (logits_process.py:1536-1553)

Pinning: the reference repository ships NO tests / golden vectors for this path,
and its own model file does not import under the installed transformers.  The restatement is therefore
pinned against stock HF-5.15 T5ForConditionalGeneration run in this container
(tests/test_oracle.py::test_oracle_vs_hf_live, fixtures made by tests/golden/make_golden.py) -- i.e. against outputs of
the third-party dependency that holds the arithmetic, not against reference-owned vectors.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# config + parameter init (HF:modeling_t5.py:563-616 `_init_weights`, factor = 1.0)
# --------------------------------------------------------------------------------------
@dataclass
class T5Cfg:
    vocab_size: int = 32100
    d_model: int = 512
    d_kv: int = 64
    d_ff: int = 2048
    num_layers: int = 6
    num_decoder_layers: int = 6
    num_heads: int = 8
    rel_buckets: int = 32
    rel_max_distance: int = 128
    eps: float = 1e-6
    dropout: float = 0.1
    ff_act: str = "relu"            # "relu" | "gated-gelu"
    whole_word_size: int = 512      # P5_T5.py:64
    pad_id: int = 0
    eos_id: int = 1

    @property
    def inner(self) -> int:
        return self.num_heads * self.d_kv

    @staticmethod
    def named(name: str, **kw) -> "T5Cfg":
        presets = {
            "t5-small": dict(d_model=512, d_ff=2048, num_heads=8, num_layers=6, num_decoder_layers=6),
            "t5-base": dict(d_model=768, d_ff=3072, num_heads=12, num_layers=12, num_decoder_layers=12),
            "t5-large": dict(d_model=1024, d_ff=4096, num_heads=16, num_layers=24, num_decoder_layers=24),
            "tiny": dict(vocab_size=300, d_model=128, d_ff=256, num_heads=2, num_layers=2, num_decoder_layers=2),
        }
        d = dict(presets[name])
        d.update(kw)
        return T5Cfg(**d)


def param_shapes(cfg: T5Cfg) -> "Dict[str, Tuple[int, ...]]":
    """HF state-dict key layout, without the duplicated tied keys."""
    d, inner, F, H = cfg.d_model, cfg.inner, cfg.d_ff, cfg.num_heads
    out: Dict[str, Tuple[int, ...]] = {}
    out["shared.weight"] = (cfg.vocab_size, d)
    out["encoder.whole_word_embeddings.weight"] = (cfg.whole_word_size, d)

    def attn(prefix):
        out[prefix + ".q.weight"] = (inner, d)
        out[prefix + ".k.weight"] = (inner, d)
        out[prefix + ".v.weight"] = (inner, d)
        out[prefix + ".o.weight"] = (d, inner)

    def ff(prefix):
        if cfg.ff_act == "relu":
            out[prefix + ".DenseReluDense.wi.weight"] = (F, d)
        else:
            out[prefix + ".DenseReluDense.wi_0.weight"] = (F, d)
            out[prefix + ".DenseReluDense.wi_1.weight"] = (F, d)
        out[prefix + ".DenseReluDense.wo.weight"] = (d, F)

    for i in range(cfg.num_layers):
        p = f"encoder.block.{i}"
        attn(p + ".layer.0.SelfAttention")
        if i == 0:
            out[p + ".layer.0.SelfAttention.relative_attention_bias.weight"] = (cfg.rel_buckets, H)
        out[p + ".layer.0.layer_norm.weight"] = (d,)
        ff(p + ".layer.1")
        out[p + ".layer.1.layer_norm.weight"] = (d,)
    out["encoder.final_layer_norm.weight"] = (d,)
    for i in range(cfg.num_decoder_layers):
        p = f"decoder.block.{i}"
        attn(p + ".layer.0.SelfAttention")
        if i == 0:
            out[p + ".layer.0.SelfAttention.relative_attention_bias.weight"] = (cfg.rel_buckets, H)
        out[p + ".layer.0.layer_norm.weight"] = (d,)
        attn(p + ".layer.1.EncDecAttention")
        out[p + ".layer.1.layer_norm.weight"] = (d,)
        ff(p + ".layer.2")
        out[p + ".layer.2.layer_norm.weight"] = (d,)
    out["decoder.final_layer_norm.weight"] = (d,)
    return out


def init_params(cfg: T5Cfg, seed: int = 2023, dtype=torch.float32) -> Dict[str, Tensor]:
    """Random init with the std's of HF `_init_weights` (factor 1.0); whole-word table N(0,1)
    (P5_T5.py:64-67 keeps nn.Embedding's default init)."""
    g = torch.Generator().manual_seed(seed)
    d, dk, H, F = cfg.d_model, cfg.d_kv, cfg.num_heads, cfg.d_ff
    p: Dict[str, Tensor] = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith("layer_norm.weight"):
            t = torch.ones(shape, dtype=dtype)
        else:
            if name in ("shared.weight", "encoder.whole_word_embeddings.weight"):
                std = 1.0
            elif name.endswith(".q.weight"):
                std = (d * dk) ** -0.5
            elif name.endswith(".k.weight") or name.endswith(".v.weight"):
                std = d ** -0.5
            elif name.endswith(".o.weight"):
                std = (H * dk) ** -0.5
            elif ".wi" in name:
                std = d ** -0.5
            elif name.endswith(".wo.weight"):
                std = F ** -0.5
            elif name.endswith("relative_attention_bias.weight"):
                std = d ** -0.5
            else:
                raise KeyError(name)
            t = (torch.randn(shape, generator=g, dtype=torch.float32) * std).to(dtype)
        p[name] = t
    return p


# --------------------------------------------------------------------------------------
# counter-based dropout RNG shared bit-for-bit with the HIP kernels (openp5_amd/csrc/p5_rng.h)
# --------------------------------------------------------------------------------------
_M32 = 0xFFFFFFFF


def _mix32(x: Tensor) -> Tensor:
    """lowbias32 integer hash on int64 tensors holding uint32 values."""
    x = x & _M32
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    x = x ^ (x >> 16)
    return x


def dropout_keep_mask(seed: int, site: int, numel: int, p: float) -> Tensor:
    """keep[i] for linear element index i; identical to `p5_keep(seed, site, i, thr)` on the device."""
    idx = torch.arange(numel, dtype=torch.int64)
    site_key = (site * 0x85EBCA6B + 0x27D4EB2F) & _M32
    sm = int(_mix32(torch.tensor([((seed & _M32) + site_key) & _M32], dtype=torch.int64))[0])
    h = _mix32((idx & _M32) ^ sm)
    thr = int(p * 16777216.0)
    return (h >> 8) >= thr


class DropoutPlan:
    """Site numbering shared with the engine (openp5_amd/csrc/p5_rng.h `p5_site_id`)."""

    def __init__(self, seed: int, p: float):
        self.seed, self.p = seed, p

    def apply(self, x: Tensor, site: int) -> Tensor:
        if self.p <= 0.0:
            return x
        keep = dropout_keep_mask(self.seed, site, x.numel(), self.p).view(x.shape)
        return x * keep.to(x.dtype) * (1.0 / (1.0 - self.p))


def site_id(stack: int, layer: int, which: int) -> int:
    """stack 0 = encoder, 1 = decoder.  which: 0 embed, 1 self-attn probs, 2 self-attn out,
    3 cross-attn probs, 4 cross-attn out, 5 ffn hidden, 6 ffn out, 7 final norm."""
    return (stack * 64 + layer) * 8 + which


# --------------------------------------------------------------------------------------
# model pieces
# --------------------------------------------------------------------------------------
def rmsnorm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    """HF:modeling_t5.py:59-72 -- fp32 variance, no mean subtraction, no bias."""
    var = x.to(torch.float32 if x.dtype != torch.float64 else torch.float64).pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def relative_position_bucket(rel: Tensor, bidirectional: bool, num_buckets: int, max_distance: int) -> Tensor:
    """HF:modeling_t5.py:217-262 (fp32 log, truncation toward zero)."""
    ret = torch.zeros_like(rel)
    if bidirectional:
        num_buckets //= 2
        ret = ret + (rel > 0).to(torch.long) * num_buckets
        rel = torch.abs(rel)
    else:
        rel = -torch.min(rel, torch.zeros_like(rel))
    max_exact = num_buckets // 2
    is_small = rel < max_exact
    large = max_exact + (
        torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)
    ).to(torch.long)
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return ret + torch.where(is_small, rel, large)


def bucket_lut(max_len: int, bidirectional: bool, num_buckets: int = 32, max_distance: int = 128) -> Tensor:
    """bucket for rel = key - query in [-(max_len-1), max_len-1]; index = rel + max_len - 1."""
    rel = torch.arange(-(max_len - 1), max_len, dtype=torch.long)
    return relative_position_bucket(rel, bidirectional, num_buckets, max_distance)


def compute_bias(table: Tensor, qlen: int, klen: int, bidirectional: bool, cfg: T5Cfg, q_offset: int = 0) -> Tensor:
    """HF:modeling_t5.py:264-279 -> [1,H,q,k]."""
    ctx = torch.arange(qlen, dtype=torch.long)[:, None] + q_offset
    mem = torch.arange(klen, dtype=torch.long)[None, :]
    b = relative_position_bucket(mem - ctx, bidirectional, cfg.rel_buckets, cfg.rel_max_distance)
    return table[b].permute(2, 0, 1).unsqueeze(0)


def attention(x: Tensor, kv: Tensor, P: Dict[str, Tensor], prefix: str, bias: Tensor, cfg: T5Cfg,
              drop: Optional[DropoutPlan], site_probs: int) -> Tensor:
    """HF:modeling_t5.py:144-173,281-369: unscaled QK^T + bias(+mask) -> softmax -> PV -> o."""
    B, Lq, _ = x.shape
    Lk = kv.shape[1]
    H, dk = cfg.num_heads, cfg.d_kv
    q = (x @ P[prefix + ".q.weight"].T).view(B, Lq, H, dk).transpose(1, 2)
    k = (kv @ P[prefix + ".k.weight"].T).view(B, Lk, H, dk).transpose(1, 2)
    v = (kv @ P[prefix + ".v.weight"].T).view(B, Lk, H, dk).transpose(1, 2)
    s = q @ k.transpose(2, 3) + bias
    p = torch.softmax(s, dim=-1)
    if drop is not None:
        p = drop.apply(p, site_probs)
    o = (p @ v).transpose(1, 2).reshape(B, Lq, H * dk)
    return o @ P[prefix + ".o.weight"].T


def gelu_new(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def ffn(x: Tensor, P: Dict[str, Tensor], prefix: str, cfg: T5Cfg, drop: Optional[DropoutPlan], site_h: int,
        taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """HF:modeling_t5.py:75-123.  taps[prefix + ".pre"] = ReLU pre-activations (for mask-flip audits of fp32 comparisons)."""
    if cfg.ff_act == "relu":
        pre = x @ P[prefix + ".DenseReluDense.wi.weight"].T
        if taps is not None:
            taps[prefix + ".pre"] = pre.detach()
        h = torch.relu(pre)
    else:
        h = gelu_new(x @ P[prefix + ".DenseReluDense.wi_0.weight"].T) * (x @ P[prefix + ".DenseReluDense.wi_1.weight"].T)
    if drop is not None:
        h = drop.apply(h, site_h)
    return h @ P[prefix + ".DenseReluDense.wo.weight"].T


def _neg(dtype) -> float:
    return torch.finfo(dtype).min


def encoder_forward(P: Dict[str, Tensor], cfg: T5Cfg, input_ids: Tensor, whole_word_ids: Tensor,
                    attention_mask: Tensor, drop: Optional[DropoutPlan] = None,
                    taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """JointEncoder.forward, P5_T5.py:74-204."""
    E = P["shared.weight"]
    x = E[input_ids] + P["encoder.whole_word_embeddings.weight"][whole_word_ids]       # P5_T5.py:94-100
    if drop is not None:
        x = drop.apply(x, site_id(0, 0, 0))                                            # P5_T5.py:125
    B, L = input_ids.shape
    ext = (1.0 - attention_mask[:, None, None, :].to(x.dtype)) * _neg(x.dtype)          # P5_T5.py:110-112
    bias = compute_bias(P["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"],
                        L, L, True, cfg) + ext                                         # P5_T5.py:136-143
    if taps is not None:
        taps["enc.embed"] = x
    for i in range(cfg.num_layers):
        p = f"encoder.block.{i}"
        n = rmsnorm(x, P[p + ".layer.0.layer_norm.weight"], cfg.eps)
        a = attention(n, n, P, p + ".layer.0.SelfAttention", bias, cfg, drop, site_id(0, i, 1))
        x = x + (drop.apply(a, site_id(0, i, 2)) if drop is not None else a)
        n = rmsnorm(x, P[p + ".layer.1.layer_norm.weight"], cfg.eps)
        f = ffn(n, P, p + ".layer.1", cfg, drop, site_id(0, i, 5), taps)
        x = x + (drop.apply(f, site_id(0, i, 6)) if drop is not None else f)
        if taps is not None:
            taps[f"enc.block{i}"] = x
    x = rmsnorm(x, P["encoder.final_layer_norm.weight"], cfg.eps)
    if drop is not None:
        x = drop.apply(x, site_id(0, 0, 7))                                            # P5_T5.py:179-180
    if taps is not None:
        taps["enc.out"] = x
    return x


def shift_right(labels: Tensor, cfg: T5Cfg) -> Tensor:
    """HF:modeling_t5.py:618-637 == P5_T5.py:329: decoder_start = pad = 0, -100 -> pad."""
    s = labels.new_zeros(labels.shape)
    s[:, 1:] = labels[:, :-1]
    s[:, 0] = cfg.pad_id
    return s.masked_fill(s == -100, cfg.pad_id)


def decoder_forward(P: Dict[str, Tensor], cfg: T5Cfg, dec_ids: Tensor, enc: Tensor, enc_mask: Tensor,
                    drop: Optional[DropoutPlan] = None, taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """T5Stack(decoder), used at P5_T5.py:338-350.  Full (non-cached) causal decode over dec_ids."""
    x = P["shared.weight"][dec_ids]
    if drop is not None:
        x = drop.apply(x, site_id(1, 0, 0))
    B, T = dec_ids.shape
    causal = torch.ones(T, T, dtype=torch.bool).tril()
    cmask = torch.where(causal, torch.zeros((), dtype=x.dtype), torch.full((), _neg(x.dtype), dtype=x.dtype))[None, None]
    sbias = compute_bias(P["decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"],
                         T, T, False, cfg) + cmask
    xbias = (1.0 - enc_mask[:, None, None, :].to(x.dtype)) * _neg(x.dtype)            # zero position bias + enc mask
    for i in range(cfg.num_decoder_layers):
        p = f"decoder.block.{i}"
        n = rmsnorm(x, P[p + ".layer.0.layer_norm.weight"], cfg.eps)
        a = attention(n, n, P, p + ".layer.0.SelfAttention", sbias, cfg, drop, site_id(1, i, 1))
        x = x + (drop.apply(a, site_id(1, i, 2)) if drop is not None else a)
        n = rmsnorm(x, P[p + ".layer.1.layer_norm.weight"], cfg.eps)
        a = attention(n, enc, P, p + ".layer.1.EncDecAttention", xbias, cfg, drop, site_id(1, i, 3))
        x = x + (drop.apply(a, site_id(1, i, 4)) if drop is not None else a)
        n = rmsnorm(x, P[p + ".layer.2.layer_norm.weight"], cfg.eps)
        f = ffn(n, P, p + ".layer.2", cfg, drop, site_id(1, i, 5), taps)
        x = x + (drop.apply(f, site_id(1, i, 6)) if drop is not None else f)
        if taps is not None:
            taps[f"dec.block{i}"] = x
    x = rmsnorm(x, P["decoder.final_layer_norm.weight"], cfg.eps)
    if drop is not None:
        x = drop.apply(x, site_id(1, 0, 7))
    if taps is not None:
        taps["dec.out"] = x
    return x


def lm_logits(P: Dict[str, Tensor], cfg: T5Cfg, dec_out: Tensor) -> Tensor:
    """P5_T5.py:352-361: tied head with d_model^-0.5 rescale."""
    return (dec_out * (cfg.d_model ** -0.5)) @ P["shared.weight"].T


def p5_forward_nll(P: Dict[str, Tensor], cfg: T5Cfg, input_ids: Tensor, whole_word_ids: Tensor,
                   attention_mask: Tensor, labels: Tensor, drop: Optional[DropoutPlan] = None,
                   taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """P5_T5.forward -> flat [B*T] per-token NLL (CrossEntropyLoss(ignore_index=-100, reduction='none'),
    P5_T5.py:368-369; pads are label 0 and are NOT ignored here)."""
    enc = encoder_forward(P, cfg, input_ids, whole_word_ids, attention_mask, drop, taps)
    dec_in = shift_right(labels, cfg)
    dec = decoder_forward(P, cfg, dec_in, enc, attention_mask, drop, taps)
    logits = lm_logits(P, cfg, dec)
    if taps is not None:
        taps["logits"] = logits
    nll = torch.nn.functional.cross_entropy(logits.view(-1, logits.shape[-1]), labels.reshape(-1),
                                            ignore_index=-100, reduction="none")
    return nll


def sequence_scores(P: Dict[str, Tensor], cfg: T5Cfg, input_ids: Tensor, whole_word_ids: Tensor, attention_mask: Tensor,
                    sequences: Tensor, return_token_logprobs: bool = False):
    """Teacher-forced score of GIVEN hypotheses, the quantity HF's beam search ranks by (generation/utils.py:3182 with
    length_penalty 1.0; DistributedRunner.py:361-374 reads it as `sequences_scores`): sum over the generated tokens -- up to and
    including </s> -- of log_softmax(full-vocabulary logits)[token], divided by their number.  `sequences` [B, K, T] start with the
    decoder start token (pad) and are pad-filled after </s>.  Used by the dataset-level gate to check every score a lower-precision
    search returns against this oracle's arithmetic on the SAME sequence, whatever the searches decided on the way.
    return_token_logprobs: also return the per-token log-probabilities [B, K, T-1] (0 after </s>) -- their running sums are the scores
    the search's intermediate decisions compare (beam_search: `running_scores`) -- and the token counts [B, K]."""
    B, K, T = sequences.shape
    labels = sequences[:, :, 1:].reshape(B * K, T - 1)
    rep = lambda t: t.repeat_interleave(K, dim=0)      # noqa: E731
    nll = p5_forward_nll(P, cfg, rep(input_ids), rep(whole_word_ids), rep(attention_mask), labels).view(B * K, T - 1)
    is_eos = labels == cfg.eos_id
    n = torch.where(is_eos.any(dim=1), is_eos.float().argmax(dim=1) + 1, torch.full((B * K,), T - 1))
    keep = torch.arange(T - 1)[None, :] < n[:, None]
    scores = (-(nll * keep).sum(dim=1) / n.clamp(min=1)).view(B, K)
    if return_token_logprobs:
        return scores, (-(nll * keep)).view(B, K, T - 1), n.view(B, K)
    return scores


def runner_loss(nll: Tensor, output_attention: Tensor) -> Tensor:
    """DistributedRunner.py:72-77."""
    B, T = output_attention.shape
    m = (output_attention != 0).to(nll.dtype)
    loss = nll.view(B, T) * m
    return (loss.sum(dim=1) / m.sum(dim=1).clamp(min=1)).mean()


# --------------------------------------------------------------------------------------
# optimizer: clip_grad_norm_(1.0) + transformers.AdamW(4.26) + linear warmup
# --------------------------------------------------------------------------------------
def linear_schedule_lr(base_lr: float, step: int, warmup: int, total: int) -> float:
    """get_linear_schedule_with_warmup lambda evaluated for the step about to be taken
    (`step` = number of scheduler.step() calls so far)."""
    if step < warmup:
        return base_lr * float(step) / float(max(1, warmup))
    return base_lr * max(0.0, float(total - step) / float(max(1, total - warmup)))


def clip_coef(grads: Sequence[Tensor], max_norm: float) -> Tuple[float, float]:
    """torch.nn.utils.clip_grad_norm_: coef = max_norm/(norm+1e-6) clamped to 1."""
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads))
    coef = min(1.0, max_norm / (total + 1e-6))
    return total, coef


def adamw_hf_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, t: int, lr: float, beta1=0.9, beta2=0.999,
                  eps=1e-6, wd=0.01) -> None:
    """transformers.AdamW.step (4.26), correct_bias=True; decay applied AFTER the update."""
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    denom = v.sqrt().add_(eps)
    step_size = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    p.addcdiv_(m, denom, value=-step_size)
    if wd > 0.0:
        p.add_(p, alpha=-lr * wd)


# --------------------------------------------------------------------------------------
# trie-constrained beam search (restated from HF 5.15 generation/utils.py:3008-3560)
# --------------------------------------------------------------------------------------
def beam_search(P: Dict[str, Tensor], cfg: T5Cfg, input_ids: Tensor, whole_word_ids: Tensor,
                attention_mask: Tensor, allowed_fn: Callable[[int, Tensor], List[int]], num_beams: int,
                max_length: int, num_return_sequences: Optional[int] = None,
                length_penalty: float = 1.0, decision_margins: Optional[dict] = None) -> Tuple[Tensor, Tensor]:
    """Returns (sequences [B*K, <=max_length] int64 starting with pad(0), sequences_scores [B*K]).

    decision_margins (test instrumentation, optional dict, filled in place): for every batch item the smallest score margin by
    which this search took any of its discrete decisions -- "set" [B]: membership of the top-2K candidates, the K running beams,
    the rank < K condition on EOS candidates, the top-K merge of finished hypotheses, the early-stop comparison;
    "order" [B, nret-1]: gaps between consecutive final scores; "set_per_token" [B]: as "set", with the comparisons of running SUMS
    (the first three) divided by the number of generated tokens they sum over, i.e. in the unit of the final scores, which are per-token
    means (a lower-precision search's error on a sum grows with the number of terms).  A lower-precision search whose scores stay within half of a
    margin of these cannot decide differently; where it does differ, the margin says whether it was allowed to.

    Per step: log_softmax over the FULL vocab, then -inf outside allowed_fn(batch_id, prefix)
    (not renormalised), + running score, top-2K over K*V, EOS candidates ranked < K finish with
    score / (generated_len ** length_penalty), best K non-finished continue; early_stopping=False heuristic.
    An empty allowed list is replaced by [pad] (SURVEY.md 8(c) delta 1): such beams already carry -inf.
    """
    K = num_beams
    nret = num_return_sequences or K
    B = input_ids.shape[0]
    V = cfg.vocab_size
    enc = encoder_forward(P, cfg, input_ids, whole_word_ids, attention_mask)
    enc_k = enc.repeat_interleave(K, dim=0)
    mask_k = attention_mask.repeat_interleave(K, dim=0)
    beams_to_keep = 2 * K
    top_mask = torch.cat([torch.ones(K, dtype=torch.bool), torch.zeros(K, dtype=torch.bool)])

    running = torch.full((B, K, max_length), cfg.pad_id, dtype=torch.long)
    sequences = running.clone()
    running_scores = torch.zeros(B, K, dtype=enc.dtype)
    running_scores[:, 1:] = -1e9
    beam_scores = torch.full((B, K), -1e9, dtype=enc.dtype)
    is_finished = torch.zeros(B, K, dtype=torch.bool)
    unsat = torch.ones(B, 1, dtype=torch.bool)
    gen_len = torch.zeros(B, K, dtype=torch.long)          # generated length of finished hyps (for cropping)
    run_len_dummy = None
    cur_len = 1
    neg_inf = float("-inf")
    LIVE = -1.0e8                                            # scores above this are real candidates (dead ones carry <= -1e9)
    set_margin = torch.full((B,), float("inf"), dtype=torch.float64)
    set_margin_per_token = torch.full((B,), float("inf"), dtype=torch.float64)

    def _gap(hi: Tensor, lo: Tensor, cond: Optional[Tensor] = None, tokens: int = 1):
        # `tokens`: how many generated tokens the two compared scores sum over (1 for scores that are already per-token means)
        nonlocal set_margin, set_margin_per_token
        ok = (hi > LIVE) & (lo > LIVE)
        if cond is not None:
            ok = ok & cond
        g = torch.where(ok, (hi - lo).abs().to(torch.float64), torch.full_like(hi, float("inf"), dtype=torch.float64))
        set_margin = torch.minimum(set_margin, g)
        set_margin_per_token = torch.minimum(set_margin_per_token, g / tokens)

    while True:
        flat = running[:, :, :cur_len].reshape(B * K, cur_len)
        dec = decoder_forward(P, cfg, flat, enc_k, mask_k)
        logits = lm_logits(P, cfg, dec[:, -1:, :])[:, 0, :].to(torch.float32)
        lp = torch.log_softmax(logits, dim=-1)
        maskv = torch.full_like(lp, neg_inf)
        for r in range(B * K):
            allowed = allowed_fn(r // K, flat[r])
            if len(allowed) == 0:
                allowed = [cfg.pad_id]
            maskv[r, allowed] = 0.0
        lp = (lp + maskv).to(enc.dtype)
        lp = lp.view(B, K, V) + running_scores[:, :, None]
        lp = lp.view(B, K * V)
        if decision_margins is not None:
            wide = torch.topk(lp, k=min(beams_to_keep + 1, lp.shape[1]))[0]
            if wide.shape[1] > beams_to_keep:
                _gap(wide[:, beams_to_keep - 1], wide[:, beams_to_keep], unsat[:, 0], cur_len)      # membership of the top-2K
        topk_lp, topk_idx = torch.topk(lp, k=beams_to_keep)
        topk_beam = topk_idx // V
        topk_tok = topk_idx % V
        topk_seq = torch.take_along_dim(running, topk_beam[:, :, None], dim=1).clone()
        topk_seq[:, :, cur_len] = topk_tok
        hits = (topk_tok == cfg.eos_id) | (cur_len + 1 >= max_length)
        # e. running beams for next iteration
        run_lp = topk_lp + hits.to(topk_lp.dtype) * -1.0e9
        nxt = torch.topk(run_lp, k=K)[1]
        if decision_margins is not None:
            rs = torch.sort(run_lp, dim=1, descending=True)[0]
            _gap(rs[:, K - 1], rs[:, K], unsat[:, 0], cur_len)                                     # the K running beams
            _gap(topk_lp[:, K - 1], topk_lp[:, K], unsat[:, 0] & (hits[:, K - 1] | hits[:, K]), cur_len)   # EOS candidate ranked < K or not
        running = torch.take_along_dim(topk_seq, nxt[:, :, None], dim=1)
        running_scores = torch.take_along_dim(run_lp, nxt, dim=1)
        # f. finished beams
        just = hits & top_mask[None, :]
        fin_lp = topk_lp / ((cur_len + 1 - 1) ** length_penalty)
        fin_lp = fin_lp + (~unsat).to(fin_lp.dtype) * -1.0e9
        fin_lp = fin_lp + (~just).to(fin_lp.dtype) * -1.0e9
        m_seq = torch.cat([sequences, topk_seq], dim=1)
        m_sc = torch.cat([beam_scores, fin_lp], dim=1)
        m_fin = torch.cat([is_finished, just], dim=1)
        m_len = torch.cat([gen_len, torch.full((B, beams_to_keep), cur_len, dtype=torch.long)], dim=1)
        sel = torch.topk(m_sc, k=K)[1]
        if decision_margins is not None:
            ms = torch.sort(m_sc, dim=1, descending=True)[0]
            _gap(ms[:, K - 1], ms[:, K])                                                           # top-K merge of finished hypotheses
        sequences = torch.take_along_dim(m_seq, sel[:, :, None], dim=1)
        beam_scores = torch.take_along_dim(m_sc, sel, dim=1)
        is_finished = torch.take_along_dim(m_fin, sel, dim=1)
        gen_len = torch.take_along_dim(m_len, sel, dim=1)
        cur_len += 1
        # g. early-stop heuristic (early_stopping=False)
        best_possible = running_scores[:, :1] / ((cur_len - 1) ** length_penalty)
        worst_fin = torch.where(is_finished, beam_scores.min(dim=1, keepdim=True)[0], torch.tensor(-1.0e9, dtype=beam_scores.dtype))
        if decision_margins is not None:
            _gap(best_possible[:, 0], worst_fin.min(dim=1)[0], unsat[:, 0])                        # early-stop comparison (all K finished)
        unsat = unsat & torch.any(best_possible > worst_fin, dim=-1, keepdim=True)
        if not (bool(unsat.any()) and not bool(hits.all())):
            break
    seqs = sequences[:, :nret, :].reshape(B * nret, max_length)
    scores = beam_scores[:, :nret].reshape(B * nret)
    out_len = 1 + int(gen_len[:, :nret].max())
    if decision_margins is not None:
        fs = beam_scores[:, :nret].to(torch.float64)
        og = (fs[:, :-1] - fs[:, 1:]).abs()
        og = torch.where((fs[:, :-1] > LIVE) & (fs[:, 1:] > LIVE), og, torch.full_like(og, float("inf")))
        decision_margins["set"] = set_margin
        decision_margins["set_per_token"] = set_margin_per_token
        decision_margins["order"] = og
    return seqs[:, :out_len], scores
