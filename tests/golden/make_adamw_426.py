"""Writes tests/golden/adamw_426.json: the reference's optimizer step, from the PUBLISHED algorithms, in plain Python floats (fp64).

What the reference runs per step (runner/DistributedRunner.py:81,85-86; runner/SingleRunner.py:191-217):
    torch.nn.utils.clip_grad_norm_(model.parameters(), args.clip)        # torch 1.8.1
    optimizer.step()            # transformers.AdamW(lr, eps=adam_eps, betas=(0.9, 0.999)), correct_bias=True (default), weight_decay on every T5 param
    scheduler.step()            # transformers.get_linear_schedule_with_warmup(optimizer, warmup_steps, total_steps)
Neither library version can be installed here (transformers 5.x has no AdamW; pinned: transformers==4.26.0, torch==1.8.1), so this file
restates their published step() bodies operation by operation -- scalar Python arithmetic only, no torch / numpy, so that it cannot share
a mistake with oracle/t5_oracle.py::adamw_hf_step (torch ops) or with csrc/p5_elem.h::p5_adamw_kernel:

  transformers/optimization.py (v4.26.0) class AdamW, step():
      exp_avg.mul_(beta1).add_(grad, alpha=(1.0 - beta1))
      exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1.0 - beta2)
      denom = exp_avg_sq.sqrt().add_(group["eps"])
      step_size = group["lr"]
      if group["correct_bias"]:
          bias_correction1 = 1.0 - beta1 ** state["step"];  bias_correction2 = 1.0 - beta2 ** state["step"]
          step_size = step_size * math.sqrt(bias_correction2) / bias_correction1
      p.data.addcdiv_(exp_avg, denom, value=-step_size)
      if group["weight_decay"] > 0.0:
          p.data.add_(p.data, alpha=(-group["lr"] * group["weight_decay"]))          # decay AFTER the update, on the updated value
  transformers/optimization.py get_linear_schedule_with_warmup.lr_lambda(current_step):
      if current_step < num_warmup_steps: return float(current_step) / float(max(1, num_warmup_steps))
      return max(0.0, float(num_training_steps - current_step) / float(max(1, num_training_steps - num_warmup_steps)))
      (LambdaLR: the lr of optimizer step number k, 0-based, is base_lr * lr_lambda(k): the scheduler is constructed -- one implicit step --
       before the first optimizer.step() and stepped after each one)
  torch/nn/utils/clip_grad.py (v1.8.1) clip_grad_norm_(parameters, max_norm, norm_type=2.0):
      total_norm = torch.norm(torch.stack([torch.norm(p.grad.detach(), 2.0) for p in parameters]), 2.0)
      clip_coef = max_norm / (total_norm + 1e-6)
      if clip_coef < 1: for p in parameters: p.grad.detach().mul_(clip_coef)

Case: three parameters (a [2,3] matrix, a [5] vector, a [1] scalar), 5 optimizer steps, base lr 1e-2, warmup 2 of 6 total steps (steps 0-1 warm
up -- step 0 has lr 0 --, 2-4 decay), max_norm 1.0: gradients scaled so that steps 0, 2, 4 clip and steps 1, 3 do not.
Run: python tests/golden/make_adamw_426.py
"""
import json
import math
import os

BETA1, BETA2, EPS, WD, BASE_LR, MAX_NORM = 0.9, 0.999, 1e-6, 0.01, 1e-2, 1.0
WARMUP, TOTAL, STEPS = 2, 6, 5
SHAPES = {"a": [2, 3], "b": [5], "c": [1]}


def lcg(seed):
    """deterministic stand-in for a random stream (values in [-1, 1)), exactly representable in fp32"""
    state = seed
    while True:
        state = (state * 1103515245 + 12345) % (1 << 31)
        yield ((state >> 7) % 4096) / 2048.0 - 1.0


def lr_lambda(step):
    if step < WARMUP:
        return float(step) / float(max(1, WARMUP))
    return max(0.0, float(TOTAL - step) / float(max(1, TOTAL - WARMUP)))


def main():
    rnd = lcg(2023)
    n = {k: int(math.prod(v)) for k, v in SHAPES.items()}
    p = {k: [next(rnd) for _ in range(n[k])] for k in SHAPES}
    m = {k: [0.0] * n[k] for k in SHAPES}
    v = {k: [0.0] * n[k] for k in SHAPES}
    out = {"hyper": {"beta1": BETA1, "beta2": BETA2, "eps": EPS, "weight_decay": WD, "lr": BASE_LR, "max_norm": MAX_NORM, "warmup_steps": WARMUP,
                     "total_steps": TOTAL}, "shapes": SHAPES, "p0": {k: list(x) for k, x in p.items()}, "steps": []}
    for step in range(STEPS):
        scale = 3.0 if step % 2 == 0 else 0.125           # |g| ~ 3 * sqrt(12 / 3) > 1 clips; 0.125 * 2 < 1 does not
        g = {k: [scale * next(rnd) for _ in range(n[k])] for k in SHAPES}
        # clip_grad_norm_
        total_norm = math.sqrt(sum(math.sqrt(sum(x * x for x in g[k])) ** 2 for k in SHAPES))
        clip_coef = MAX_NORM / (total_norm + 1e-6)
        gc = {k: ([x * clip_coef for x in g[k]] if clip_coef < 1 else list(g[k])) for k in SHAPES}
        # AdamW.step
        lr = BASE_LR * lr_lambda(step)
        t = step + 1
        for k in SHAPES:
            for i in range(n[k]):
                m[k][i] = m[k][i] * BETA1 + (1.0 - BETA1) * gc[k][i]
                v[k][i] = v[k][i] * BETA2 + (1.0 - BETA2) * gc[k][i] * gc[k][i]
                denom = math.sqrt(v[k][i]) + EPS
                step_size = lr * math.sqrt(1.0 - BETA2 ** t) / (1.0 - BETA1 ** t)
                p[k][i] = p[k][i] + (-step_size) * (m[k][i] / denom)
                p[k][i] = p[k][i] + (-lr * WD) * p[k][i]
        out["steps"].append({"lr": lr, "grad": {k: list(x) for k, x in g.items()}, "total_norm": total_norm, "clipped": clip_coef < 1,
                             "p": {k: list(x) for k, x in p.items()}, "m": {k: list(x) for k, x in m.items()}, "v": {k: list(x) for k, x in v.items()}})
    assert [s["clipped"] for s in out["steps"]] == [True, False, True, False, True], [s["total_norm"] for s in out["steps"]]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "adamw_426.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, "norms", [round(s["total_norm"], 4) for s in out["steps"]], "lrs", [s["lr"] for s in out["steps"]])


if __name__ == "__main__":
    main()
