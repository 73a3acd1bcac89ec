"""Fused clip + AdamW + linear-warmup schedule over the model's flat parameter arena.

Semantics = what the reference runner does per step (DistributedRunner.py:81,85-87 with SingleRunner.py:178-219):
`clip_grad_norm_(params, clip)`; `transformers.AdamW(lr, eps=adam_eps, betas=(0.9,0.999), correct_bias=True)` with
weight_decay on EVERY parameter (the reference's no_decay name filter matches nothing in T5);
`get_linear_schedule_with_warmup`; `zero_grad`.  Two kernels per step (sum of squares, update) instead of ~10 launches
per tensor; the update also refreshes the bf16 compute shadow, so no separate cast pass is needed.
"""
import ctypes
import math

import torch


def linear_schedule_with_warmup(step: int, warmup: int, total: int) -> float:
    """lr multiplier of transformers.get_linear_schedule_with_warmup after `step` scheduler steps."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    return max(0.0, float(total - step) / float(max(1, total - warmup)))


class FusedAdamW:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01, max_grad_norm=1.0,
                 warmup_steps=0, total_steps=0):
        self.model = model
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.warmup_steps, self.total_steps = warmup_steps, total_steps
        self.t = 0              # optimizer steps taken
        self.sched_steps = 0    # scheduler.step() calls so far (lr for the NEXT step uses this)
        dev = model._flat.device
        self.m = torch.zeros_like(model._flat)
        self.v = torch.zeros_like(model._flat)
        self.sumsq = torch.zeros(1024, dtype=torch.float32, device=dev)    # per-workgroup partial sums of g^2

    def current_lr(self):
        if self.total_steps <= 0:
            return self.lr
        return self.lr * linear_schedule_with_warmup(self.sched_steps, self.warmup_steps, self.total_steps)

    def step(self, grad_accum: int = 1):
        """`grad_accum`: the arena holds the SUM over that many micro-batches (and, after the all-reduce, over ranks); the
        mean is taken by the update kernel's gradient scale."""
        mdl = self.model
        if getattr(mdl, "_grads_dead", False):
            # zero_grad() only marks the gradient arena dead (model.zero_grad docstring).  torch holds p.grad = None here and its AdamW
            # skips every such parameter (no update, no moment change, no per-parameter step count); the reference loop still calls
            # scheduler.step() (DistributedRunner.py:85-86).  Same here: the dead arena is never re-applied.
            self.sched_steps += 1
            return
        if hasattr(mdl, "finish_exchange"):
            mdl.finish_exchange()        # data parallel: the all-reduced buckets are first read here (model._engine_backward)
        be, lib = mdl._be, mdl._lib
        P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
        sp = be.stream_ptr()
        self.t += 1
        world = max(1, int(getattr(mdl, "ddp_world", 1)))
        use_clip = self.max_grad_norm is not None and self.max_grad_norm > 0
        if use_clip:
            be.check(lib.p5_grad_sumsq(P(mdl._grads), mdl._n, P(self.sumsq), sp), "p5_grad_sumsq")
        # over the engine's own arenas: with the transposed / norm-folded bf16 copies bound the update writes them as well (csrc: p5_adamw_tiles_kernel)
        fresh = ctypes.c_int(0)
        be.check(lib.p5_engine_adamw_step(mdl._engine, P(self.m), P(self.v), P(self.sumsq) if use_clip else None, float(self.max_grad_norm or 0.0),
                                          1.0 / (world * max(1, int(grad_accum))), float(self.current_lr()), self.betas[0], self.betas[1], self.eps, self.wd,
                                          self.t, ctypes.byref(fresh), sp), "p5_engine_adamw_step")
        mdl.mark_params_updated(shadow_fresh=mdl._shadow is not None, copies_fresh=bool(fresh.value))
        self.sched_steps += 1   # scheduler.step() (DistributedRunner.py:86)

    def zero_grad(self, set_to_none=True):
        self.model.zero_grad()

    def grad_norm(self):
        return math.sqrt(float(self.sumsq.double().sum().item())) / max(1, int(getattr(self.model, "ddp_world", 1)))

    def state_dict(self):
        return {"m": self.m, "v": self.v, "t": self.t, "sched_steps": self.sched_steps, "warmup_steps": self.warmup_steps,
                "total_steps": self.total_steps}

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"].to(self.m.device))
        self.v.copy_(sd["v"].to(self.v.device))
        self.t, self.sched_steps = int(sd["t"]), int(sd["sched_steps"])
