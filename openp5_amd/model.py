"""`P5T5Native` -- drop-in for OpenP5's `P5_T5` model object (/root/reference/src/src_t5/model/P5_T5.py:207)
backed by the HIP engine in libp5hip.so.

It satisfies exactly the uses the reference's launcher/runner make of the model:
  * `forward(input_ids, whole_word_ids, attention_mask, labels, alpha=..., return_dict=True)["loss"]` is the flat
    [B*T] fp32 per-token NLL (P5_T5.py:368-369, consumed at DistributedRunner.py:63-77), differentiable;
  * `generate(input_ids, attention_mask, whole_word_ids, max_length, prefix_allowed_tokens_fn, num_beams,
    num_return_sequences, output_scores, return_dict_in_generate)` -> {"sequences", "sequences_scores"}
    (DistributedRunner.py:361-374);
  * `nn.Module` protocol with HF T5 state-dict keys incl. the duplicated tied keys;
    `shared.weight` is the single tied [V, d] tensor, writable in place (utils/initialization.py:27-29);
  * `resize_token_embeddings(n)` (main.py:193), `.train()/.eval()/.zero_grad()/.parameters()`.

All parameters are views into ONE flat fp32 arena (and all gradients views into a second arena of the same
layout), which is what lets clip + AdamW be two flat kernels and the data-parallel all-reduce a handful of
contiguous buckets issued while the backward is still running.
"""
from __future__ import annotations

import contextlib
import warnings
import os
import ctypes
import threading
import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _abi
from .trie import CompiledTrie, Trie, find_trie


@dataclass
class P5ModelConfig:
    """The T5Config fields the path reads (HF configuration_t5.py:44-62 defaults = t5-small)."""
    vocab_size: int = 32128
    d_model: int = 512
    d_kv: int = 64
    d_ff: int = 2048
    num_layers: int = 6
    num_decoder_layers: Optional[int] = None
    num_heads: int = 8
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    dropout_rate: float = 0.1
    layer_norm_epsilon: float = 1e-6
    feed_forward_proj: str = "relu"
    whole_word_size: int = 512           # P5_T5.py:64
    pad_token_id: int = 0
    eos_token_id: int = 1
    decoder_start_token_id: int = 0

    @staticmethod
    def from_backbone(name: str, **kw) -> "P5ModelConfig":
        presets = {
            "t5-small": dict(d_model=512, d_ff=2048, num_heads=8, num_layers=6),
            "t5-base": dict(d_model=768, d_ff=3072, num_heads=12, num_layers=12),
            "t5-large": dict(d_model=1024, d_ff=4096, num_heads=16, num_layers=24),
        }
        key = name.split("/")[-1]
        if key not in presets:
            raise ValueError(f"unknown backbone {name!r} (known: {sorted(presets)})")
        d = dict(presets[key])
        d.update(kw)
        return P5ModelConfig(**d)

    @staticmethod
    def from_hf(cfg) -> "P5ModelConfig":
        g = lambda k, dflt=None: getattr(cfg, k, dflt)
        return P5ModelConfig(
            vocab_size=g("vocab_size"), d_model=g("d_model"), d_kv=g("d_kv"), d_ff=g("d_ff"), num_layers=g("num_layers"),
            num_decoder_layers=g("num_decoder_layers"), num_heads=g("num_heads"),
            relative_attention_num_buckets=g("relative_attention_num_buckets", 32),
            relative_attention_max_distance=g("relative_attention_max_distance", 128),
            dropout_rate=g("dropout_rate", 0.1), layer_norm_epsilon=g("layer_norm_epsilon", 1e-6),
            feed_forward_proj=g("feed_forward_proj", "relu"), pad_token_id=g("pad_token_id", 0) or 0,
            eos_token_id=g("eos_token_id", 1), decoder_start_token_id=g("decoder_start_token_id", 0) or 0)


def relative_position_bucket_lut(half: int, bidirectional: bool, num_buckets: int, max_distance: int) -> torch.Tensor:
    """bucket(rel) for rel = key_pos - query_pos in [-half, half]; same op sequence (fp32 log, truncation) as
    HF modeling_t5.py:217-262 so the table is bit-identical to what T5Attention.compute_bias indexes with."""
    rel = torch.arange(-half, half + 1, dtype=torch.long)
    ret = torch.zeros_like(rel)
    nb = num_buckets
    if bidirectional:
        nb //= 2
        ret = ret + (rel > 0).to(torch.long) * nb
        rel = torch.abs(rel)
    else:
        rel = -torch.min(rel, torch.zeros_like(rel))
    max_exact = nb // 2
    is_small = rel < max_exact
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, nb - 1))
    return (ret + torch.where(is_small, rel, large)).to(torch.int32)


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class _P5LossFn(torch.autograd.Function):
    """autograd seam: forward = p5_forward (activations stay in the engine workspace), backward = p5_backward
    writing straight into the gradient arena."""

    @staticmethod
    def forward(ctx, anchor, model, input_ids, whole_word_ids, attention_mask, labels):
        ctx.model = model
        return model._engine_forward(input_ids, whole_word_ids, attention_mask, labels)

    @staticmethod
    def backward(ctx, dnll):
        ctx.model._engine_backward(dnll)
        return None, None, None, None, None, None


class _GenLane:
    """What ONE in-flight generate() call owns: its engines (the bf16 / fp32 search engine and, for verified generation, the fp32 verification
    engine -- all bound to the model's ONE set of parameter arenas), its workspaces, its pinned read-back buffer, its HIP stream.  Lane 0 is
    the model's own engine on the caller's stream; further lanes let several batches be in flight at once (`P5T5Native.map_lanes`)."""
    __slots__ = ("idx", "engine", "engine_v", "fold_v", "fold_v_dirty", "ws", "ver_hdr", "forced", "stream")

    def __init__(self, idx, engine, stream=None):
        self.idx, self.engine, self.stream = idx, engine, stream
        self.engine_v, self.fold_v, self.fold_v_dirty, self.ver_hdr = None, None, True, None
        self.ws, self.forced = {}, ([], [])


class P5T5Native(nn.Module):
    LUT_HALF = 512
    # Engine-internal second HIP stream (weight gradients, K/V projections, clears off the main stream).  OFF since round 3: with
    # the layer-grouped weight-gradient launches every phase of the encoder backward fills the GPU from ONE stream, and the ~50
    # cross-stream event waits per step cost more than the overlap returns (MI355X, C2 step: 4.74 ms without, 5.07-5.4 ms with).
    # Data-parallel gradient exchange uses its own communication stream either way (`_engine_backward`).
    use_side_stream = False
    use_transposed_weights = True     # bf16: keep W^T of the layer weights for the data gradients (p5_engine_bind_transposed)
    fuse_decode_norms = True      # generate(): fold the decoder RMSNorms into the GEMMs around them
    # generate() of a bf16 model: "verified" = the bf16 search (with `verify_extra_beams` more beams) proposes, one fp32 pass decides -- the
    # returned lists and scores are the fp32 search's (include/p5hip.h, csrc/p5_verify.h); "draft" = the plain bf16 search.  An fp32 model
    # always runs the plain (fp32) search.
    generation_mode = "verified"
    verify_extra_beams = 6
    VERIFY_MAX_K = 22             # the replay's candidate pool (K x 2K <= 1024 entries of LDS, csrc/p5_verify.h)
    VERIFY_MAX_ROWS = 512         # rows per user of the fp32 pass = queries per (user, head) of its cross-attention launch
    verify_escalation = (22,)     # extra beams of the wider draft a FLAGGED user gets before the plain fp32 search is the last resort
    verify_share_encoder = True   # verified mode: the draft starts from the verification pass's fp32 encoder output (one encoder pass per batch)
    gen_lanes = 3                 # batches in flight in `map_lanes` (the runner's evaluation loops, bench.py): lanes overlap each other's latency-bound chains
    prefix_fast_forward = True    # the steps every item shares ("<dataset> item _") as one teacher-forced pass (p5_generate_set_forced_prefix)

    def __init__(self, config, dtype: str = "bf16", device=None, backend=None, seed: int = 2023):
        super().__init__()
        if not isinstance(config, P5ModelConfig):
            config = P5ModelConfig.from_hf(config)
        if config.num_decoder_layers is None:
            config.num_decoder_layers = config.num_layers
        self.config = config
        if backend is None:
            from ._lib import hip_backend
            backend = hip_backend(device)
        self._be = backend
        self._lib = backend.lib
        self.compute_dtype = {"bf16": 1, "bfloat16": 1, "fp32": 0, "float32": 0}[str(dtype).replace("torch.", "")]
        self._engine = ctypes.c_void_p()
        self._ws = None
        self._gen_ws = None
        self._anchor = torch.zeros(1, device=backend.device, requires_grad=True)
        self.ddp_world = 1          # set by the runner: gradient all-reduce across ranks during backward
        self.ddp_group = None
        self._ddp_sync = True       # False on all but the last micro-batch of a gradient-accumulation group
        self.ddp_bucket_dtype = "fp32"   # "bf16": gradient buckets travel as bf16 (half the bytes on the xGMI links)
        self.staged_backward = False     # run the stage-by-stage backward (the data-parallel code path) even at world size 1, without collectives
        self.ddp_lazy_wait = True        # gradient buckets are waited for at their first use (the optimizer step), not at the end of the backward
        self._pending_half = False
        self.ddp_timing = False          # record device time the main stream spends waiting for the gradient exchange (bench.py)
        self.ddp_wait_ms = []
        self._pending = []
        self._staged_ranges, self._bucket16 = None, None
        self._side = None
        self._comm = None
        self._fold = None
        self._fold_dirty = True; self._mark_lanes_dirty()
        self._lanes = []            # generation lanes (lane 0 = this model's engine on the caller's stream)
        self._tls = threading.local()
        self._stats_lock = threading.Lock()
        self.verify_stats = {"calls": 0, "users": 0, "escalated_users": 0, "fallback_users": 0, "rows": 0, "rows_per_user_max": 0, "draft_beams": 0,
                             "wide_fp32_users": 0}
        self.last_generate_path = None      # "verified" | "fp32_search" | "draft_bf16": which search the most recent generate() call ran
        self._warned_wide_verified = False
        self._shadow_t = None       # transposed bf16 copy of the layer weights (data gradients run on the forward GEMM kernel)
        self._grads_dead = False    # zero_grad(set_to_none=True) was called and no backward has run since: `.grad` holds stale values
        self._tr_dirty = True
        self._build(seed)

    # ------------------------------------------------------------------ engine / arena plumbing
    def _cfg_struct(self):
        c = self.config
        ff = c.feed_forward_proj
        if ff not in ("relu", "gated-gelu"):
            raise ValueError(f"feed_forward_proj={ff!r} not supported (relu | gated-gelu)")
        return _abi.P5Config(
            vocab_size=c.vocab_size, d_model=c.d_model, d_kv=c.d_kv, d_ff=c.d_ff, n_enc_layers=c.num_layers,
            n_dec_layers=c.num_decoder_layers, n_heads=c.num_heads, rel_buckets=c.relative_attention_num_buckets,
            rel_max_distance=c.relative_attention_max_distance, whole_word_size=c.whole_word_size,
            gated_gelu=1 if ff == "gated-gelu" else 0, dtype=self.compute_dtype, eps=c.layer_norm_epsilon,
            dropout=c.dropout_rate, pad_id=c.pad_token_id, eos_id=c.eos_token_id)

    def _create_engine(self):
        if self._engine:
            self._lib.p5_engine_destroy(self._engine)
        self._engine = ctypes.c_void_p()
        self._fold, self._fold_dirty = None, True       # sized by (and bound to) the engine
        self._drop_lanes()          # lane engines (verification engines, extra search engines) are bound to the old arena: rebuilt on demand
        cfg = self._cfg_struct()
        self._be.check(self._lib.p5_engine_create(ctypes.byref(cfg), ctypes.byref(self._engine)), "p5_engine_create")
        table = []
        name = ctypes.create_string_buffer(256)
        off, rows, cols = ctypes.c_int64(), ctypes.c_int(), ctypes.c_int()
        i = 0
        while self._lib.p5_param_table(self._engine, i, name, 256, ctypes.byref(off), ctypes.byref(rows), ctypes.byref(cols)) == 0:
            table.append((name.value.decode(), off.value, rows.value, cols.value))
            i += 1
        self._table = table
        self._n = int(self._lib.p5_param_count(self._engine))

    def _build(self, seed, old_state: Optional[Dict[str, torch.Tensor]] = None):
        dev = self._be.device
        self._create_engine()
        self._flat = torch.zeros(self._n, dtype=torch.float32, device=dev)
        self._grads = torch.zeros(self._n, dtype=torch.float32, device=dev)
        self._shadow = torch.zeros(self._n, dtype=torch.bfloat16, device=dev) if self.compute_dtype == 1 else None
        c = self.config
        self._lut_enc = relative_position_bucket_lut(self.LUT_HALF, True, c.relative_attention_num_buckets, c.relative_attention_max_distance).to(dev)
        self._lut_dec = relative_position_bucket_lut(self.LUT_HALF, False, c.relative_attention_num_buckets, c.relative_attention_max_distance).to(dev)
        self._rng_cpu = [int(seed) & 0xFFFFFFFF, 0]
        self._rng = torch.tensor(self._rng_cpu, dtype=torch.int64, device=dev).to(torch.int32)
        # drop any previous parameter modules, then (re)register views with HF names
        for k in list(self._modules.keys()):
            del self._modules[k]
        self._views = {}
        for name, off, rows, cols in self._table:
            shape = (cols,) if name.endswith("layer_norm.weight") else (rows, cols)
            view = self._flat[off:off + rows * cols].view(shape)
            p = nn.Parameter(view, requires_grad=True)
            self._register_dotted(name, p)
            self._views[name] = (off, rows * cols, shape)
        self._init_weights(seed)
        if old_state is not None:
            self._copy_in(old_state, strict=False)
        self._bind()
        self._shadow_dirty = True
        self._fold_dirty = True; self._mark_lanes_dirty()

    def _register_dotted(self, name, p):
        parts = name.split(".")
        mod = self
        for part in parts[:-1]:
            if part not in mod._modules:
                mod.add_module(part, nn.Module())
            mod = mod._modules[part]
        mod.register_parameter(parts[-1], p)

    def _bind(self):
        self._be.check(self._lib.p5_engine_bind(self._engine, _ptr(self._flat), _ptr(self._grads), _ptr(self._shadow), _ptr(self._lut_enc),
                                                 _ptr(self._lut_dec), self.LUT_HALF, _ptr(self._rng)), "p5_engine_bind")
        if not self._be.is_emulator and self.use_side_stream:
            # weight-gradient GEMMs run on a second HIP stream, off the dgrad critical path
            if self._side is None:
                self._side = torch.cuda.Stream(device=self._be.device)
            self._lib.p5_engine_set_side_stream(self._engine, ctypes.c_void_p(self._side.cuda_stream))
        if self.compute_dtype == 1 and self.use_transposed_weights:
            nbytes = int(self._lib.p5_transposed_bytes(self._engine))
            self._shadow_t = torch.zeros(nbytes, dtype=torch.uint8, device=self._be.device)
            self._be.check(self._lib.p5_engine_bind_transposed(self._engine, _ptr(self._shadow_t), self._be.stream_ptr()), "p5_engine_bind_transposed")
            self._tr_dirty = True

    @torch.no_grad()
    def _init_weights(self, seed):
        """HF `_init_weights` std's (modeling_t5.py:563-616, factor 1.0); whole-word table N(0,1) (P5_T5.py:64-67)."""
        g = torch.Generator().manual_seed(int(seed))
        c = self.config
        d, dk, H, F = c.d_model, c.d_kv, c.num_heads, c.d_ff
        for name, p in self.named_parameters():
            if name.endswith("layer_norm.weight"):
                p.fill_(1.0)
                continue
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
            else:
                std = d ** -0.5
            p.copy_((torch.randn(p.shape, generator=g) * std).to(p.device))

    def _sync_shadow(self):
        if self.compute_dtype == 1 and self._shadow_dirty:
            self._be.check(self._lib.p5_refresh_shadow(self._engine, self._be.stream_ptr()), "p5_refresh_shadow")
        self._shadow_dirty = False

    def _sync_decode_fold(self):
        """Decode-step weights with the RMSNorm weights folded in (include/p5hip.h: p5_refresh_decode_fold); rebuilt lazily
        from the fp32 master parameters whenever they have changed since the last generate()."""
        if not self.fuse_decode_norms:
            return
        if self._fold is None:
            n = int(self._lib.p5_decode_fold_count(self._engine))
            self._fold = torch.empty(n, dtype=torch.bfloat16 if self.compute_dtype == 1 else torch.float32, device=self._flat.device)
            self._be.check(self._lib.p5_engine_bind_decode_fold(self._engine, _ptr(self._fold)), "p5_engine_bind_decode_fold")
            self._fold_dirty = True; self._mark_lanes_dirty()
        if self._fold_dirty:
            self._be.check(self._lib.p5_refresh_decode_fold(self._engine, self._be.stream_ptr()), "p5_refresh_decode_fold")
            self._fold_dirty = False

    def mark_params_updated(self, shadow_fresh: bool = False, copies_fresh: bool = False):
        """Call after writing parameters outside the fused optimizer (which refreshes the bf16 shadow -- and, `copies_fresh`, the transposed
        and norm-folded copies -- itself)."""
        self._shadow_dirty = not shadow_fresh
        self._fold_dirty = True; self._mark_lanes_dirty()
        self._tr_dirty = not (copies_fresh and shadow_fresh)

    def _sync_transposed(self):
        """W^T of the layer weights for the next backward, and W diag(ln) of the projections behind a T5LayerNorm for the next forward
        (both live in the buffer bound with p5_engine_bind_transposed; one refresh call after every parameter change)."""
        if self._shadow_t is not None and self._tr_dirty:
            self._be.check(self._lib.p5_refresh_transposed(self._engine, self._be.stream_ptr()), "p5_refresh_transposed")
            self._tr_dirty = False

    # ------------------------------------------------------------------ nn.Module protocol
    def _apply(self, fn, recurse=True):
        probe = fn(torch.zeros(1, device=self._flat.device))
        if probe.device != self._flat.device:
            if not self._be.is_emulator and probe.device.type != "cuda":
                raise RuntimeError("P5T5Native lives on the HIP device; there is no CPU path")
            if probe.device != self._be.device:
                raise RuntimeError(f"P5T5Native was built for {self._be.device}; build it with device={probe.device} instead")
        if probe.dtype != torch.float32:
            raise RuntimeError("master parameters are fp32; choose the compute dtype with dtype='bf16'|'fp32'")
        return self

    TIED = ("encoder.embed_tokens.weight", "decoder.embed_tokens.weight", "lm_head.weight")

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        sd = super().state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        shared = sd[prefix + "shared.weight"]
        for k in self.TIED:
            sd[prefix + k] = shared
        return sd

    @torch.no_grad()
    def _copy_in(self, state_dict, strict):
        own = dict(self.named_parameters())
        missing = [k for k in own if k not in state_dict]
        unexpected = []
        for k, v in state_dict.items():
            if k in own:
                if tuple(own[k].shape) != tuple(v.shape):
                    if k in ("shared.weight",) and v.shape[1] == own[k].shape[1]:
                        n = min(v.shape[0], own[k].shape[0])
                        own[k][:n].copy_(v[:n].to(own[k].device, torch.float32))
                        continue
                    raise RuntimeError(f"size mismatch for {k}: {tuple(v.shape)} vs {tuple(own[k].shape)}")
                own[k].copy_(v.to(own[k].device, torch.float32))
            elif k in self.TIED or k == "decoder.block.0.layer.1.EncDecAttention.relative_attention_bias.weight":
                continue   # tied duplicates / key ignored on load (P5_T5.py:213-215)
            else:
                unexpected.append(k)
        if "shared.weight" not in state_dict:
            for k in self.TIED:
                if k in state_dict:
                    own["shared.weight"].copy_(state_dict[k].to(own["shared.weight"].device, torch.float32))
                    missing = [m for m in missing if m != "shared.weight"]
                    break
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing={missing} unexpected={unexpected}")
        self._shadow_dirty = True
        self._fold_dirty = True; self._mark_lanes_dirty()
        self._tr_dirty = True
        return missing, unexpected

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        missing, unexpected = self._copy_in(state_dict, strict)
        from torch.nn.modules.module import _IncompatibleKeys
        return _IncompatibleKeys(missing, unexpected)

    @classmethod
    def from_pretrained(cls, backbone, config=None, state_dict=None, **kw):
        """`P5_T5.from_pretrained(args.backbone, config=config)` (main.py:176,184).  Weights come from `state_dict`
        or from a local HF checkpoint directory/file (`pytorch_model.bin` / `model.safetensors`); with neither (no
        network in this environment) the HF `_init_weights` distribution is used.  A missing
        `encoder.whole_word_embeddings.weight` keeps its fresh N(0,1) init, as in the reference."""
        import os
        if config is None:
            config = P5ModelConfig.from_backbone(str(backbone))
        model = cls(config, **kw)
        sd = state_dict
        if sd is None and isinstance(backbone, str) and os.path.exists(backbone):
            path = backbone
            if os.path.isdir(path):
                for cand in ("model.safetensors", "pytorch_model.bin"):
                    if os.path.exists(os.path.join(path, cand)):
                        path = os.path.join(path, cand)
                        break
            if path.endswith(".safetensors"):
                from safetensors.torch import load_file
                sd = load_file(path)
            elif os.path.isfile(path):
                sd = torch.load(path, map_location="cpu")
        if sd is not None:
            model.load_state_dict(sd, strict=False)
        return model

    @torch.no_grad()
    def resize_token_embeddings(self, new_num_tokens: int):
        """main.py:193: keep the old rows, new rows ~ N(0, 1) (HF `_init_weights` for the shared table)."""
        old = self.config.vocab_size
        if new_num_tokens == old:
            return self.shared
        state = {k: v.detach().clone() for k, v in super().state_dict().items()}
        old_E = state.pop("shared.weight")
        self.config.vocab_size = int(new_num_tokens)
        seed = self._rng_cpu[0]
        self._build(seed)
        self._copy_in(state, strict=False)
        n = min(old, new_num_tokens)
        self.shared.weight[:n].copy_(old_E[:n])
        self._shadow_dirty = True
        self._fold_dirty = True; self._mark_lanes_dirty()
        return self.shared

    def get_input_embeddings(self):
        return self.shared

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def zero_grad(self, set_to_none: bool = True):
        """set_to_none=True (torch's default, what the reference loop's optimizer.zero_grad() does, DistributedRunner.py:93): the
        gradients are DEAD until the next backward, which overwrites them -- a first micro-batch STORES every Linear gradient and
        clears only the ~1 MB of atomically accumulated ones -- so there is no 242 MB fill and no read-modify-write of zeros.  The
        `.grad` views stay attached (re-attaching ~130 of them every step is host time); their contents are undefined until that
        backward, where torch would show None.  set_to_none=False: the arena is cleared now."""
        if set_to_none:
            self._be.check(self._lib.p5_engine_discard_grads(self._engine), "discard_grads")
            self._grads_dead = True         # (FusedAdamW.step skips the update until a backward has rewritten them, as torch skips grad=None)
        else:
            self._be.check(self._lib.p5_engine_clear_grads(self._engine, self._be.stream_ptr()), "clear_grads")
            self._grads_dead = False        # real zeros: a step would apply them (weight decay only), as torch does

    def tie_weights(self):
        return None

    def begin_micro_batch(self, first: bool, sync: bool):
        """Gradient accumulation (--gradient_accumulation_steps > 1): every gradient kernel of the engine ADDS into the
        arena (split-K atomics, partial-sum reductions, `+=`), the only overwrite is the clear at the start of a backward --
        so a micro-batch other than the first tells the engine to skip that clear; `sync` = exchange gradients across ranks
        in this backward (only the last micro-batch of a group does)."""
        if not first:
            self._lib.p5_engine_grads_zeroed(self._engine)
        self._ddp_sync = bool(sync)

    # ------------------------------------------------------------------ RNG for dropout
    def set_dropout_seed(self, seed: int, step: int = 0):
        self._rng_cpu = [int(seed) & 0xFFFFFFFF, int(step) & 0xFFFFFFFF]
        self._rng.copy_(torch.tensor(self._rng_cpu, dtype=torch.int64).to(torch.int32))

    # ------------------------------------------------------------------ forward / backward
    def _workspace(self, nbytes, which="_ws"):
        cur = getattr(self, which)
        if cur is None or cur.numel() < nbytes:
            raw = torch.empty(int(nbytes * 1.05) + 512, dtype=torch.uint8, device=self._be.device)
            if os.environ.get("P5_POISON_WS"):      # debugging aid: NaN patterns in every byte the engine has not written yet
                raw.fill_(0xFF)
            skew = (-raw.data_ptr()) % 256          # the engine wants a 256-byte aligned base
            cur = raw[skew:skew + int(nbytes * 1.05) + 255]
            setattr(self, which, cur)
        return cur

    @staticmethod
    def _i64(t, device):
        return t.to(device=device, dtype=torch.int64).contiguous()

    def _engine_forward(self, input_ids, whole_word_ids, attention_mask, labels):
        dev = self._be.device
        B, L = input_ids.shape
        T = labels.shape[1]
        self._sync_shadow()
        self._sync_transposed()     # (no-op unless the parameters changed; a backward may follow this forward)
        ws = self._workspace(self._lib.p5_train_workspace_bytes(self._engine, B, L, T))
        nll = torch.empty(B * T, dtype=torch.float32, device=dev)
        training = 1 if (self.training and self.config.dropout_rate > 0) else 0
        if training:
            self._advance_dropout_step()
        self._saved_inputs = (input_ids, whole_word_ids, attention_mask, labels)   # keep device buffers alive
        self._be.check(self._lib.p5_forward(self._engine, _ptr(input_ids), _ptr(whole_word_ids), _ptr(attention_mask), _ptr(labels), B, L, T,
                                            training, _ptr(nll), _ptr(ws), ws.numel(), self._be.stream_ptr()), "p5_forward")
        return nll

    def _engine_backward(self, dnll):
        """dnll = gradient of the per-token NLL (autograd path), or None after `p5_forward_loss`: the engine then seeds
        the backward with the gradient of the runner's masked-mean loss itself."""
        if dnll is not None:
            dnll = dnll.to(torch.float32).contiguous()
        self.finish_exchange()          # (an exchange nobody consumed: its buckets must land before this backward rewrites the arena)
        lib, eng, sp = self._lib, self._engine, self._be.stream_ptr()
        exchange = self.ddp_world > 1 and self._ddp_sync
        if exchange or self.staged_backward:
            import torch.distributed as dist
            # ONE library call enqueues every stage and records an event behind each gradient range that became final (the staged backward
            # issues the same grouped weight-gradient launches as the single-GPU step, so ranges leave in two-layer groups); the exchange
            # of range k goes to a communication stream that waits for event k -- it overlaps the rest of the backward on the device, and
            # the host makes one call per step instead of one per stage
            nst = lib.p5_backward_num_stages(eng)
            if self._staged_ranges is None or len(self._staged_ranges) < 2 * nst:
                self._staged_ranges = (ctypes.c_int64 * (2 * nst))()
            nr = ctypes.c_int(0)
            self._be.check(lib.p5_backward_staged(eng, _ptr(dnll), sp, self._staged_ranges, nst, ctypes.byref(nr)), "p5_backward_staged")
            self._staged_n = nr.value
            self._pending = []
            half = str(self.ddp_bucket_dtype).replace("torch.", "") in ("bf16", "bfloat16")
            if exchange:
                comm = self._side
                if comm is None and self._flat.is_cuda:
                    if self._comm is None:
                        self._comm = torch.cuda.Stream(device=self._be.device)
                    comm = self._comm
                if half and (self._bucket16 is None or self._bucket16.numel() != self._grads.numel()):
                    self._bucket16 = torch.empty(self._grads.numel(), dtype=torch.bfloat16, device=self._grads.device)      # allocated once, not per step
                ctx = torch.cuda.stream(comm) if comm is not None else contextlib.nullcontext()
                with ctx:
                    for k in range(nr.value):
                        b0, e0 = int(self._staged_ranges[2 * k]), int(self._staged_ranges[2 * k + 1])
                        if comm is not None:
                            self._be.check(lib.p5_backward_staged_wait(eng, k, ctypes.c_void_p(comm.cuda_stream)), "p5_backward_staged_wait")
                        seg = self._grads[b0:e0]
                        if half:                     # bf16 bucket: cast, reduce, cast back (below)
                            buf = self._bucket16[b0:e0]
                            buf.copy_(seg)
                        else:
                            buf = seg
                        self._pending.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.ddp_group, async_op=True), buf, seg))
            self._pending_half = half
            # the buckets are waited for where the gradients are first READ (FusedAdamW.step -> finish_exchange), bucket by bucket, not here:
            # with a host-blocking backend (gloo) the host goes on to the next batch's collation while the last buckets travel, and nothing
            # between the end of the backward and the optimizer step touches the gradient arena.  `ddp_lazy_wait = False` restores the wait
            # at the end of the backward (a caller that reads .grad before stepping must call finish_exchange() itself).
            if not (exchange and self.ddp_lazy_wait):
                self.finish_exchange()
        else:
            self._be.check(lib.p5_backward(eng, _ptr(dnll), sp), "p5_backward")
        self._grads_dead = False
        for name, p in self.named_parameters():
            if p.grad is None:
                off, n, shape = self._views[name]
                p.grad = self._grads[off:off + n].view(shape)

    def finish_exchange(self):
        """Wait (bucket by bucket, in issue order) for the gradient all-reduces the last backward enqueued; a no-op when none is pending.
        Called by FusedAdamW.step before the gradient norm is taken; call it yourself before reading `.grad` of a data-parallel model."""
        if not self._pending:
            return
        exchange = True
        timing = self.ddp_timing and self._flat.is_cuda
        if timing:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        for w, buf, seg in self._pending:
            w.wait()
            if self._pending_half:
                seg.copy_(buf)      # every rank holds the same bf16 sums -> identical fp32 gradients -> identical updates
        self._pending = []
        for cs in (self._side, self._comm):
            if cs is not None and exchange and self._flat.is_cuda:
                # NCCL/RCCL's wait() already orders the CURRENT stream after the collective; backends that complete on the
                # stream they were issued from (gloo on device tensors) need the explicit edge comm -> main
                torch.cuda.current_stream().wait_stream(cs)
        if timing:
            ev1.record()
            self.ddp_wait_ms.append((ev0, ev1))     # (read by bench.py after a synchronize: main-stream time spent waiting for the buckets)

    def loss_and_backward(self, input_ids, whole_word_ids, attention_mask, labels, output_attention):
        """Fused form of the reference loop body DistributedRunner.py:63-80: forward, masked-mean loss
        (`(nll.view(B,T) * m).sum(1) / m.sum(1).clamp(min=1)).mean()`) and backward, all inside the engine -- the loss is
        reduced behind the cross-entropy kernel and the backward is seeded from the label mask, so no torch autograd graph and
        none of the ~10 elementwise launches of the generic path.  Returns the loss as a 0-dim tensor; gradients are in `.grad`."""
        dev = self._be.device
        input_ids = self._i64(input_ids, dev)
        B, L = input_ids.shape
        whole_word_ids = self._i64(whole_word_ids if whole_word_ids is not None else torch.zeros_like(input_ids), dev)
        attention_mask = self._i64(attention_mask if attention_mask is not None else (input_ids != self.config.pad_token_id).long(), dev)
        labels = self._i64(labels, dev)
        output_attention = self._i64(output_attention, dev)
        T = labels.shape[1]
        self._sync_shadow()
        self._sync_transposed()
        ws = self._workspace(self._lib.p5_train_workspace_bytes(self._engine, B, L, T))
        out = torch.empty(B * T + 1, dtype=torch.float32, device=dev)
        training = 1 if (self.training and self.config.dropout_rate > 0) else 0
        if training:
            self._advance_dropout_step()
        self._saved_inputs = (input_ids, whole_word_ids, attention_mask, labels, output_attention)
        self._be.check(self._lib.p5_forward_loss(self._engine, _ptr(input_ids), _ptr(whole_word_ids), _ptr(attention_mask), _ptr(labels),
                                                 _ptr(output_attention), B, L, T, training, _ptr(out), ctypes.c_void_p(out.data_ptr() + 4 * B * T),
                                                 _ptr(ws), ws.numel(), self._be.stream_ptr()), "p5_forward_loss")
        self._engine_backward(None)
        return out[B * T]

    def _advance_dropout_step(self):
        self._rng_cpu[1] = (self._rng_cpu[1] + 1) & 0xFFFFFFFF
        self._rng[1] = self._rng_cpu[1] if self._rng_cpu[1] < 2 ** 31 else self._rng_cpu[1] - 2 ** 32

    def forward(self, input_ids=None, whole_word_ids=None, attention_mask=None, labels=None, alpha=None, return_dict=True, **unused):
        """P5_T5.forward (P5_T5.py:275-386): returns {"loss": flat [B*T] per-token NLL}; `alpha` is accepted and ignored
        exactly as in the reference (P5_T5.py:294)."""
        if labels is None:
            raise ValueError("P5T5Native.forward needs labels (the runner always passes them)")
        dev = self._be.device
        input_ids = self._i64(input_ids, dev)
        if whole_word_ids is None:
            whole_word_ids = torch.zeros_like(input_ids)
        whole_word_ids = self._i64(whole_word_ids, dev)
        if attention_mask is None:
            attention_mask = (input_ids != self.config.pad_token_id).long()
        attention_mask = self._i64(attention_mask, dev)
        labels = self._i64(labels, dev)
        nll = _P5LossFn.apply(self._anchor, self, input_ids, whole_word_ids, attention_mask, labels)
        out = {"loss": nll}
        return out if return_dict else (nll,)

    # ------------------------------------------------------------------ generation lanes
    def _drop_lanes(self):
        for ln in getattr(self, "_lanes", []):
            if ln.engine_v:
                self._lib.p5_engine_destroy(ln.engine_v)
            if ln.idx > 0 and ln.engine:
                self._lib.p5_engine_destroy(ln.engine)
        self._lanes = []

    def _mark_lanes_dirty(self):
        for ln in getattr(self, "_lanes", []):
            ln.fold_v_dirty = True

    def _lane(self, i):
        """lane i, created on demand: lane 0 = the model's own engine; lane i > 0 = a second search engine over the SAME parameter arenas,
        bf16 shadow, folded decode weights and transposed / norm-folded copies (read-only during generation) with its own HIP stream."""
        while len(self._lanes) <= i:
            k = len(self._lanes)
            if k == 0:
                self._lanes.append(_GenLane(0, self._engine))
                continue
            cfg = self._cfg_struct()
            eng = ctypes.c_void_p()
            self._be.check(self._lib.p5_engine_create(ctypes.byref(cfg), ctypes.byref(eng)), "p5_engine_create (lane)")
            self._be.check(self._lib.p5_engine_bind(eng, _ptr(self._flat), _ptr(self._grads), _ptr(self._shadow), _ptr(self._lut_enc), _ptr(self._lut_dec),
                                                     self.LUT_HALF, _ptr(self._rng)), "p5_engine_bind (lane)")
            if self._fold is not None:
                self._be.check(self._lib.p5_engine_bind_decode_fold(eng, _ptr(self._fold)), "p5_engine_bind_decode_fold (lane)")
            if self._shadow_t is not None:
                self._be.check(self._lib.p5_engine_bind_transposed(eng, _ptr(self._shadow_t), self._be.stream_ptr()), "p5_engine_bind_transposed (lane)")
            self._lanes.append(_GenLane(k, eng, torch.cuda.Stream(device=self._be.device) if self._flat.is_cuda else None))
        return self._lanes[i]

    def _cur_lane(self):
        return getattr(self._tls, "lane", None) or self._lane(0)

    def _lane_workspace(self, lane, nbytes, key):
        cur = lane.ws.get(key)
        if cur is None or cur.numel() < nbytes:
            raw = torch.empty(int(nbytes * 1.05) + 512, dtype=torch.uint8, device=self._be.device)
            skew = (-raw.data_ptr()) % 256
            cur = raw[skew:skew + int(nbytes * 1.05) + 255]
            lane.ws[key] = cur
        return cur

    def map_lanes(self, fn, items, lanes=None):
        """`fn(item)` for every item, IN ORDER, with up to `lanes` calls in flight: each worker thread owns a generation lane (its own search /
        verification engines, workspaces and HIP stream over the model's one set of weights), so the latency-bound kernel chain of one
        batch's beam search overlaps the next batch's -- on one MI355X two lanes run beam-10 generation at 1.5-1.6 x the items/s of one
        (DESIGN.md 3.8).  `fn` typically calls `model.generate(...)` and post-processes its result; results come back as a generator."""
        lanes = int(self.gen_lanes if lanes is None else lanes)
        if lanes <= 1 or not self._flat.is_cuda:
            for it in items:
                yield fn(it)
            return
        import collections
        import concurrent.futures
        # parameter copies the searches read are refreshed HERE, once, on the caller's stream: no lane does it under another lane's feet
        self._sync_shadow()
        self._sync_transposed()
        self._sync_decode_fold()
        pool = [self._lane(i + 1) for i in range(lanes)]          # lanes 1 .. n; lane 0 stays with the calling thread's own generate() calls
        main = torch.cuda.current_stream()
        for ln in pool:
            ln.stream.wait_stream(main)
        q, lock, tls = collections.deque(pool), threading.Lock(), self._tls

        dev = self._be.device

        def init():
            torch.cuda.set_device(dev)          # (the current device is per host thread; rank r of a multi-GPU job is not on device 0)
            with lock:
                tls.lane = q.popleft()

        def run(it):
            ln = tls.lane
            with torch.cuda.stream(ln.stream), torch.no_grad():
                out = fn(it)
                ln.stream.synchronize()
            return out

        with concurrent.futures.ThreadPoolExecutor(max_workers=lanes, initializer=init) as ex:
            pending = collections.deque()
            for it in items:
                pending.append(ex.submit(run, it))
                while len(pending) >= 2 * lanes:
                    yield pending.popleft().result()
            while pending:
                yield pending.popleft().result()
        for ln in pool:
            main.wait_stream(ln.stream)

    # ------------------------------------------------------------------ generation
    @torch.no_grad()
    def generate(self, input_ids=None, attention_mask=None, whole_word_ids=None, max_length: int = 20,
                 prefix_allowed_tokens_fn: Optional[Callable] = None, num_beams: int = 1, num_return_sequences: Optional[int] = None,
                 output_scores: bool = False, return_dict_in_generate: bool = False, trie=None, roots=None, excluded=None, **unused):
        """Constrained beam search (DistributedRunner.py:361-371).  `prefix_allowed_tokens_fn` made by
        `openp5_amd.trie.prefix_allowed_tokens_fn` -- or by the reference's own generation_trie.prefix_allowed_tokens_fn,
        whose closure holds the Trie -- runs fully on the device.  `trie` may be passed directly (Trie / CompiledTrie).
        `excluded`: optional uint32 [B, words] bitmap from `CompiledTrie.excluded_bitmap` -- per-user history exclusion."""
        dev = self._be.device
        K = int(num_beams)
        nret = int(num_return_sequences or K)
        if trie is None and prefix_allowed_tokens_fn is not None:
            trie = find_trie(prefix_allowed_tokens_fn)
            if trie is None:
                trie = self._explore_callable(prefix_allowed_tokens_fn, input_ids.shape[0], max_length)
                roots = trie._roots
        if trie is None:
            raise ValueError("generate() needs a trie / prefix_allowed_tokens_fn (OpenP5 always decodes under the item trie)")
        if isinstance(trie, Trie) or (hasattr(trie, "trie_dict") and not isinstance(trie, CompiledTrie)):
            app = getattr(trie, "append_trie", None)
            key = (getattr(trie, "len", None), id(app), getattr(app, "len", None), getattr(trie, "bos_token_id", None))
            cache = getattr(trie, "_p5_compiled", None)
            if cache is None or cache[0] != key:
                cache = (key, CompiledTrie.from_trie(trie))        # (grafts an appended trie, trie.py)
                try:
                    trie._p5_compiled = cache
                except Exception:
                    pass
            trie = cache[1]
        off, tok, nxt = trie.device_arrays(dev)
        input_ids = self._i64(input_ids, dev)
        B, L = input_ids.shape
        if whole_word_ids is None:
            whole_word_ids = torch.zeros_like(input_ids)
        whole_word_ids = self._i64(whole_word_ids, dev)
        if attention_mask is None:
            attention_mask = (input_ids != self.config.pad_token_id).long()
        attention_mask = self._i64(attention_mask, dev)
        roots_t = None
        if roots is not None:
            roots_t = torch.as_tensor(roots, dtype=torch.int32, device=dev).contiguous()
        self._sync_shadow()
        self._sync_transposed()     # (also refreshes the folded copy W diag(ln) the encoder pass multiplies the raw residual stream with)
        self._sync_decode_fold()
        maxc = max(1, trie.max_children)
        excl_t, excl_words = None, 0
        if excluded is not None:
            excl_np = np.ascontiguousarray(excluded.cpu().numpy() if torch.is_tensor(excluded) else excluded, dtype=np.uint32)
            if excl_np.ndim != 2 or excl_np.shape[0] != B or excl_np.shape[1] * 32 < trie.n_nodes:
                raise ValueError(f"excluded bitmap must be [B={B}, >= {(trie.n_nodes + 31) // 32}] uint32, got {excl_np.shape}")
            excl_words = int(excl_np.shape[1])
            excl_t = torch.from_numpy(excl_np.view(np.int32)).to(dev)
        # no hypothesis is longer than the deepest trie path, and the search stops on the device (no per-step read-back):
        # enqueue exactly as many decode steps as can do work.  (With max_length == depth the forced finish at max_length
        # coincides with the leaves' </s>, so results are those of the unbounded call.)
        max_length = max(2, min(int(max_length), int(trie.max_depth)))
        # forced-prefix fast-forward (include/p5hip.h): the chain every item shares behind the start token, as long as no user's history
        # exclusion touches it (an excluded node on the chain would leave that user without candidates: the plain search handles that)
        lane = self._cur_lane()
        lane.forced = ([], [])
        if self.prefix_fast_forward and roots_t is None:
            ftok, fnode = trie.forced_prefix(self.config.decoder_start_token_id, self.config.eos_token_id)
            n = min(len(ftok), max_length - 2)
            if n >= 2 and excluded is not None:
                words = excl_np[:, [x >> 5 for x in fnode[:n]]]
                bits = np.asarray([x & 31 for x in fnode[:n]], dtype=np.uint32)
                if bool(((words >> bits[None, :]) & 1).any()):
                    n = 0
            if n >= 2:
                lane.forced = (ftok[:n], fnode[:n])
        mode = unused.get("generation_mode", self.generation_mode)
        if mode not in ("verified", "draft"):
            raise ValueError(f"generation_mode={mode!r} (verified | draft)")
        args = (input_ids, whole_word_ids, attention_mask, B, L, K, max_length, off, tok, nxt, roots_t, excl_t, excl_words, maxc)
        if self.compute_dtype == 1 and mode == "verified" and K <= self.VERIFY_MAX_K:
            seq, score, ln = self._generate_verified(*args)
            path = "verified"
        elif self.compute_dtype == 1 and mode == "verified":
            # the replay's candidate pool (K x 2K entries in LDS, csrc/p5_verify.h P5_VERIFY_POOL) ends at K = 22: wider searches run as the
            # plain fp32 search on the verification engine -- still the fp32 search's lists, at its speed -- and say so
            if not self._warned_wide_verified:
                warnings.warn(f"generate(num_beams={K}): verified generation covers num_beams <= {self.VERIFY_MAX_K}; running the plain fp32 beam search "
                              f"instead (same ranked lists, slower).  generation_mode='draft' selects the plain bf16 search.", RuntimeWarning, stacklevel=2)
                self._warned_wide_verified = True
            seq, score, ln = self._search_fp32(*args)
            with self._stats_lock:
                self.verify_stats["wide_fp32_users"] += B
            path = "fp32_search"
        else:
            seq, score, ln = self._search(lane.engine, "gen", *args)
            path = "draft_bf16" if self.compute_dtype == 1 else "fp32_search"
        self.last_generate_path = path
        out_len = 1 + int(ln[:, :nret].max().item())
        sequences = seq[:, :nret, :out_len].reshape(B * nret, out_len).to(torch.int64)
        scores = score[:, :nret].reshape(B * nret)
        if return_dict_in_generate:
            return {"sequences": sequences, "sequences_scores": scores if output_scores else None}
        return sequences

    def _search(self, engine, ws_attr, input_ids, whole_word_ids, attention_mask, B, L, K, max_length, off, tok, nxt, roots_t, excl_t, excl_words,
                maxc, hist=None):
        """One device beam search on `engine` (p5_generate; with `hist`, p5_generate_draft records what the search kept alive)."""
        dev = self._be.device
        lane = self._cur_lane()
        ftok, fnode = lane.forced
        if ftok:
            arr = (ctypes.c_int * len(ftok))
            self._be.check(self._lib.p5_generate_set_forced_prefix(engine, arr(*ftok), arr(*fnode), len(ftok)), "p5_generate_set_forced_prefix")
        ws = self._lane_workspace(lane, self._lib.p5_generate_workspace_bytes(engine, B, L, K, max_length, maxc, excl_words), ws_attr)
        seq = torch.zeros(B, K, max_length, dtype=torch.int32, device=dev)
        score = torch.zeros(B, K, dtype=torch.float32, device=dev)
        ln = torch.zeros(B, K, dtype=torch.int32, device=dev)
        if hist is None:
            self._be.check(self._lib.p5_generate(engine, _ptr(input_ids), _ptr(whole_word_ids), _ptr(attention_mask), B, L, K, max_length,
                                                 _ptr(off), _ptr(tok), _ptr(nxt), _ptr(roots_t), _ptr(excl_t), excl_words, maxc, _ptr(seq), _ptr(score), _ptr(ln),
                                                 _ptr(ws), ws.numel(), self._be.stream_ptr()), "p5_generate")
        else:
            self._be.check(self._lib.p5_generate_draft(engine, _ptr(input_ids), _ptr(whole_word_ids), _ptr(attention_mask), B, L, K, max_length,
                                                       _ptr(off), _ptr(tok), _ptr(nxt), _ptr(roots_t), _ptr(excl_t), excl_words, maxc, _ptr(seq), _ptr(score),
                                                       _ptr(ln), _ptr(hist), _ptr(ws), ws.numel(), self._be.stream_ptr()), "p5_generate_draft")
        return seq, score, ln

    # ------------------------------------------------------------------ verified generation (bf16 drafts, fp32 decides)
    def _verify_engine(self, lane):
        """fp32 engine over the SAME master parameter arena (no copy of the weights; its kernels read `_flat` directly), one per lane."""
        if not lane.engine_v:
            cfg = self._cfg_struct()
            cfg.dtype = 0
            lane.engine_v = ctypes.c_void_p()
            self._be.check(self._lib.p5_engine_create(ctypes.byref(cfg), ctypes.byref(lane.engine_v)), "p5_engine_create (verify)")
            self._be.check(self._lib.p5_engine_bind(lane.engine_v, _ptr(self._flat), _ptr(self._grads), None, _ptr(self._lut_enc), _ptr(self._lut_dec),
                                                     self.LUT_HALF, _ptr(self._rng)), "p5_engine_bind (verify)")
        return lane.engine_v

    def _generate_verified(self, input_ids, whole_word_ids, attention_mask, B, L, K, max_length, off, tok, nxt, roots_t, excl_t, excl_words, maxc, level=0):
        """include/p5hip.h "verified generation": the bf16 search with `verify_extra_beams` more beams proposes, ONE teacher-forced fp32
        pass over the distinct prefixes it kept alive scores them, and HF's beam search of the real width is replayed on those fp32 numbers.
        Users whose replay needed a prefix the draft had dropped are re-run through the plain fp32 search (counted in `verify_stats`)."""
        lib, dev, sp = self._lib, self._be.device, self._be.stream_ptr()
        extra = (int(self.verify_extra_beams),) + tuple(int(x) for x in self.verify_escalation)
        Kw = min(64, K + max(0, extra[min(level, len(extra) - 1)]))
        lane = self._cur_lane()
        ev = self._verify_engine(lane)
        common = (input_ids, whole_word_ids, attention_mask, B, L)
        trie_args = (off, tok, nxt, roots_t, excl_t, excl_words, maxc)
        ftok, fnode = lane.forced
        if ftok:      # (the replay skips the forced steps as the draft does)
            arr = (ctypes.c_int * len(ftok))
            self._be.check(lib.p5_generate_set_forced_prefix(ev, arr(*ftok), arr(*fnode), len(ftok)), "p5_generate_set_forced_prefix (verify)")
        ws = self._lane_workspace(lane, lib.p5_verify_workspace_bytes(ev, B, L, K, Kw, max_length, maxc, excl_words), "ver")
        self._be.check(lib.p5_verify_begin(ev, B, L, K, Kw, max_length, _ptr(off), _ptr(tok), _ptr(nxt), _ptr(roots_t), maxc, excl_words, _ptr(ws), ws.numel()),
                       "p5_verify_begin")
        # ONE encoder pass per batch: the fp32 one; the draft starts from its output
        self._be.check(lib.p5_verify_encode(ev, _ptr(input_ids), _ptr(whole_word_ids), _ptr(attention_mask), sp), "p5_verify_encode")
        if self.verify_share_encoder:
            self._be.check(lib.p5_generate_set_encoder_output(lane.engine, ctypes.c_void_p(lib.p5_verify_encoder_output(ev))), "p5_generate_set_encoder_output")
        # (carved out of the lane's workspace: its address is part of the decode-step graph's key, a fresh tensor per call would re-capture it)
        nh = int(lib.p5_generate_history_count(B, Kw, max_length))
        hist = self._lane_workspace(lane, nh * 4, "hist")[:nh * 4].view(torch.int32)
        hist.zero_()
        self._search(lane.engine, "gen", *common, Kw, max_length, *trie_args, hist=hist)
        self._be.check(lib.p5_verify_plan(ev, _ptr(hist), sp), "p5_verify_plan")
        hdr_off = int(lib.p5_verify_plan_header(ev)) - ws.data_ptr()
        hdr_dev = ws[hdr_off:hdr_off + 16].view(torch.int32)
        if ws.is_cuda:
            if lane.ver_hdr is None:
                lane.ver_hdr = torch.zeros(4, dtype=torch.int32).pin_memory()
            lane.ver_hdr.copy_(hdr_dev, non_blocking=True)      # the ONE number the host needs: rows per user of this batch
            torch.cuda.current_stream().synchronize()
            hdr = lane.ver_hdr.tolist()
        else:
            hdr = hdr_dev.cpu().tolist()
        if hdr[3]:
            raise RuntimeError("p5_verify_plan: row capacity exceeded")
        cap = int(lib.p5_verify_row_capacity(Kw, max_length))
        PU = min(cap, (max(1, int(hdr[0])) + 15) // 16 * 16)      # rows per user of the fp32 pass: the largest row count of the batch, in whole 16-row tiles
        st = self.verify_stats
        if PU > self.VERIFY_MAX_ROWS:
            # the users' rows are the queries of ONE cross-attention launch per layer (<= 512 queries per user): a deep, wide draft beyond that
            # goes to the plain fp32 search as a whole
            with self._stats_lock:
                st["calls"] += 1; st["users"] += B if level == 0 else 0; st["fallback_users"] += B
            return self._search_fp32(input_ids, whole_word_ids, attention_mask, B, L, K, max_length, *trie_args)
        seq = torch.zeros(B, K, max_length, dtype=torch.int32, device=dev)
        score = torch.zeros(B, K, dtype=torch.float32, device=dev)
        ln = torch.zeros(B, K, dtype=torch.int32, device=dev)
        missing = torch.zeros(B, dtype=torch.int32, device=dev)
        self._be.check(lib.p5_verify_run(ev, PU, _ptr(excl_t), _ptr(seq), _ptr(score), _ptr(ln), _ptr(missing), sp), "p5_verify_run")
        with self._stats_lock:
            st["calls"] += 1; st["users"] += B if level == 0 else 0; st["rows"] += int(hdr[2]); st["rows_per_user_max"] = max(st["rows_per_user_max"], int(hdr[0]))
            if level == 0:
                st["draft_beams"] = Kw
        miss = missing.nonzero().flatten()
        if miss.numel():
            sub = lambda t: None if t is None else t[miss].contiguous()     # noqa: E731
            nb = int(miss.numel())
            sub_args = (sub(input_ids), sub(whole_word_ids), sub(attention_mask), nb, L, K, max_length, off, tok, nxt, sub(roots_t), sub(excl_t), excl_words, maxc)
            if level + 1 < len(extra) and K + extra[level + 1] > Kw:
                # a flagged user first gets a WIDER draft (cheap: a sub-batch, ~2 ms) -- only what that cannot settle goes to the fp32 search
                with self._stats_lock:
                    st["escalated_users"] += nb
                s2, sc2, l2 = self._generate_verified(*sub_args, level=level + 1)
            else:
                # the fp32 search itself for these users (a prefix the fp32 search ranks among its K was not among the draft's Kw, or a
                # row of the split-product pass left the range the two-term fp16 split covers)
                with self._stats_lock:
                    st["fallback_users"] += nb
                s2, sc2, l2 = self._search_fp32(*sub_args)
            seq[miss] = s2; score[miss] = sc2; ln[miss] = l2
        return seq, score, ln

    def _search_fp32(self, *args):
        """The plain fp32 beam search on this lane's verification engine (exact fp32 MFMAs over the master arena): the last resort of the
        verified mode and what a verified call wider than VERIFY_MAX_K beams runs."""
        lib, dev, sp = self._lib, self._be.device, self._be.stream_ptr()
        lane = self._cur_lane()
        ev = self._verify_engine(lane)
        ftok, fnode = lane.forced
        if self.fuse_decode_norms:
            if lane.fold_v is None:
                n = int(lib.p5_decode_fold_count(ev))
                lane.fold_v = torch.empty(n, dtype=torch.float32, device=dev)
                self._be.check(lib.p5_engine_bind_decode_fold(ev, _ptr(lane.fold_v)), "p5_engine_bind_decode_fold (verify)")
                lane.fold_v_dirty = True
            if lane.fold_v_dirty:
                self._be.check(lib.p5_refresh_decode_fold(ev, sp), "p5_refresh_decode_fold (verify)")
                lane.fold_v_dirty = False
        return self._search(ev, "gen_v", *args)

    def time_generate(self, enable: bool = True):
        """Benchmark aid: arm (or disarm) the engine's device-time brackets around the next `generate` calls (two event records per
        call, nothing is waited for until `last_generate_timing` is read)."""
        self._be.check(self._lib.p5_generate_timing(self._engine, 1 if enable else 0, None, None), "p5_generate_timing")
        self._gen_timed = bool(enable)

    def last_generate_timing(self):
        """{"encode_ms", "decode_ms", "forced_prefix_steps"} of the calling thread's most recent `generate` call made while `time_generate()`
        was armed: device time of the encoder pass + cross-attention K/V projections (+ the forced-prefix pass), and of the decode loop
        alone (waits for that call to finish); the number of steps the forced-prefix pass covered.  None if not armed."""
        if not getattr(self, "_gen_timed", False):
            return None
        a, b = ctypes.c_float(0.0), ctypes.c_float(0.0)
        self._be.check(self._lib.p5_generate_timing(self._engine, 1, ctypes.byref(a), ctypes.byref(b)), "p5_generate_timing")
        return {"encode_ms": float(a.value), "decode_ms": float(b.value), "forced_prefix_steps": len(self._cur_lane().forced[0])}

    def _explore_callable(self, fn, B, max_length):
        """Compat path for an arbitrary prefix_allowed_tokens_fn(batch_id, prefix): enumerate it breadth-first into one
        CSR trie per batch item (as many Python calls as trie nodes, once per generate call)."""
        off, tok, nxt, roots = [0], [], [], []
        queue = []
        for b in range(B):
            roots.append(len(queue) + 0)
            queue.append((b, []))
        next_id = len(queue)
        qi = 0
        while qi < len(queue):
            b, prefix = queue[qi]
            qi += 1
            kids = [] if (len(prefix) >= max_length or (prefix and prefix[-1] == self.config.eos_token_id)) else \
                (sorted(set(int(t) for t in fn(b, torch.tensor(prefix, dtype=torch.long)))) if prefix else [self.config.decoder_start_token_id])
            for t in kids:
                tok.append(t)
                nxt.append(next_id)
                next_id += 1
                queue.append((b, prefix + [t]))
            off.append(len(tok))
        ct = CompiledTrie(np.asarray(off), np.asarray(tok), np.asarray(nxt))
        ct._roots = roots
        return ct

    def __del__(self):
        try:
            if self._engine:
                self._lib.p5_engine_destroy(self._engine)
            self._drop_lanes()
        except Exception:
            pass
