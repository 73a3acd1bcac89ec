#!/usr/bin/env python
"""bench.py -- BASELINE.json metric on the MI355X: training samples/s (+ beam-10 items/s) of the OpenP5 src_t5 hot path,
T5-small on ML-1M-shaped synthetic data (B=64/GPU, sequential prompts L=128, T=8, bf16,
dropout 0.1, V=32100), one process per GPU.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = forward + runner masked loss + backward (+ gradient all-reduce overlapped with it when N > 1) + global-norm
clip + AdamW + schedule + zero_grad, i.e. DistributedRunner.py:56-93 minus the Python collator (inputs are resident
in HBM before the timed region).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

V = 32100
BF16_PEAK_TFLOPS = 2500.0     # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def train_flops_per_sample(d, inner, F, H, NL, L, T, Vv, dk=64):
    """algorithmic FLOPs of one training sample (fwd x3), padding tokens counted."""
    enc = NL * L * 2 * (4 * d * inner + 2 * d * F) + NL * H * 4 * L * L * dk
    dec = NL * T * 2 * (6 * d * inner + 2 * d * F) + NL * L * 2 * (2 * d * inner) + NL * H * 4 * (T * T + T * L) * dk
    head = T * 2 * d * Vv
    return 3.0 * (enc + dec + head)


def synth_batch(B, L, T, device, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, V, (B, L), generator=g)
    lens = torch.randint(int(0.7 * L), L + 1, (B,), generator=g)
    lens[0] = L
    mask = (torch.arange(L)[None, :] < lens[:, None]).long()
    ids = ids * mask
    ww = torch.cumsum((torch.rand(B, L, generator=g) < 0.35).long(), 1) * mask
    ww = ww.clamp(max=511)
    labels = torch.randint(3, V, (B, T), generator=g)
    tl = torch.randint(max(2, T - 2), T + 1, (B,), generator=g)
    out_attn = (torch.arange(T)[None, :] < tl[:, None]).long()
    labels = labels * out_attn
    labels[torch.arange(B), tl - 1] = 1
    return [t.to(device) for t in (ids, ww, mask, labels, out_attn)]


def synth_item_trie(n_items, seed):
    """ML1M-like item ids: "<dataset> item_ <digits>" -> shared 4-piece prefix + 2-3 number pieces + </s>."""
    from openp5_amd.trie import Trie
    rnd = random.Random(seed)
    items = set()
    while len(items) < n_items:
        n = rnd.choice((2, 2, 3))
        items.add(tuple([0, 2000, 2001, 2002, 2003] + [rnd.randint(3000, 3999) for _ in range(n)] + [1]))
    return Trie(sorted(items))


def runner_loss(nll, out_attn):
    B, T = out_attn.shape
    m = (out_attn != 0).float()
    loss = nll.view(B, T) * m
    return (loss.sum(dim=1) / m.sum(dim=1).clamp(min=1)).mean()


def time_gemm_kernel(be, M, N, K, iters=50, wgrad=False):
    """average duration of a GEMM kernel measured with HIP events on the stream the kernel is launched on.
    wgrad=False: forward bf16 GEMM y = x W^T (both operands K-contiguous, 128x128 tile);
    wgrad=True : weight gradient dW[M,N] += dy^T x over K tokens (both operands K-strided, ring kernel, split-K atomics)."""
    import ctypes
    P = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    s = torch.cuda.current_stream()
    if wgrad:
        A = torch.randn(K, M, device=be.device).to(torch.bfloat16)
        Bm = torch.randn(K, N, device=be.device).to(torch.bfloat16)
        C = torch.zeros(M, N, device=be.device, dtype=torch.float32)
        call = lambda: be.lib.p5_op_gemm(1, P(A), P(Bm), P(C), None, M, N, K, M, N, N, 0, 1, 1, 4, 1, 0, 1.0, None, 0, 0.0, be.stream_ptr())  # noqa: E731
    else:
        A = torch.randn(M, K, device=be.device).to(torch.bfloat16)
        Bm = torch.randn(N, K, device=be.device).to(torch.bfloat16)
        C = torch.empty(M, N, device=be.device, dtype=torch.bfloat16)
        call = lambda: be.lib.p5_op_gemm(1, P(A), P(Bm), P(C), None, M, N, K, K, K, N, 0, 0, 0, 0, 0, 1, 1.0, None, 0, 0.0, be.stream_ptr())  # noqa: E731
    for _ in range(5):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(iters):
        call()
    e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def cpu_baseline(seconds_budget=20.0):
    """The CPU path timed on this box's host cores: the oracle restatement of the reference step (HF-equivalent
    fp32 T5-small, B=4, L=128, T=8: BASELINE.json configs[0]) -- forward + backward + clip + HF-AdamW."""
    from oracle import t5_oracle as O
    ncores = min(os.cpu_count() or 1, 32)     # more threads than this slows the small-matrix CPU path down
    torch.set_num_threads(ncores)
    cfg = O.T5Cfg.named("t5-small", dropout=0.0)
    P = {k: v.requires_grad_(True) for k, v in O.init_params(cfg, 2023).items()}
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    Vv = {k: torch.zeros_like(v) for k, v in P.items()}
    ids, ww, mask, labels, out_attn = synth_batch(4, 128, 8, "cpu", 1)
    times = []
    t_start = time.time()
    step = 0
    while True:
        t0 = time.time()
        nll = O.p5_forward_nll(P, cfg, ids, ww, mask, labels)
        loss = O.runner_loss(nll, out_attn)
        grads = torch.autograd.grad(loss, list(P.values()))
        _, coef = O.clip_coef(grads, 1.0)
        step += 1
        with torch.no_grad():
            for (k, p), g in zip(P.items(), grads):
                O.adamw_hf_step(p, g * coef, M[k], Vv[k], step, 1e-3)
        times.append(time.time() - t0)
        if (time.time() - t_start > seconds_budget and len(times) >= 2) or len(times) >= 12 or time.time() - t_start > 3 * seconds_budget:
            break
    rest = times[1:] if len(times) > 1 else times
    med = sorted(rest)[len(rest) // 2]
    return {"value": 4.0 / med, "unit": "samples/s", "cores": ncores, "kind": "port",
            "sample": f"{len(times)} train steps of oracle/t5_oracle.py (fp32 T5-small, B=4, L=128, T=8), median of all but the first"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--backbone", default="t5-small")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seq-len", type=int, default=128)
    ap.add_argument("--tgt-len", type=int, default=8)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-gen", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--gen-batches", type=int, default=10)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    from openp5_amd._lib import hip_backend
    from openp5_amd.model import P5ModelConfig, P5T5Native
    from openp5_amd.optim import FusedAdamW
    from openp5_amd.trie import prefix_allowed_tokens_fn

    be = hip_backend(device)
    cfg = P5ModelConfig.from_backbone(args.backbone, vocab_size=V, dropout_rate=0.1)
    model = P5T5Native(cfg, dtype=args.dtype, device=device, backend=be, seed=2023)
    model.ddp_world = world
    B, L, T = args.batch, args.seq_len, args.tgt_len
    total_steps = max(1000, args.steps + args.warmup)
    opt = FusedAdamW(model, lr=1e-3, eps=1e-6, weight_decay=0.01, max_grad_norm=1.0, warmup_steps=int(0.05 * total_steps), total_steps=total_steps)
    batch = synth_batch(B, L, T, device, 100 + rank)
    ids, ww, mask, labels, out_attn = batch
    model.train()
    model.set_dropout_seed(2023 + rank, 0)

    def step():
        out = model(input_ids=ids, whole_word_ids=ww, attention_mask=mask, labels=labels, alpha=2, return_dict=True)
        loss = runner_loss(out["loss"], out_attn)
        loss.backward()
        opt.step()
        model.zero_grad()
        return loss

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    samples_per_s = world * B * args.steps / dt
    final_loss = float(last)

    # ---- beam-10 constrained generation: items/s (B=20 users/GPU, K=10, ML1M-sized trie of 3416 items) ----
    gen = None
    if not args.no_gen:
        model.eval()
        trie = synth_item_trie(3416, 7)
        fn = prefix_allowed_tokens_fn(trie)
        gB, gK = 20, 10
        gids, gww, gmask, _, _ = synth_batch(gB, L, T, device, 500 + rank)
        for _ in range(2):
            model.generate(input_ids=gids, attention_mask=gmask, whole_word_ids=gww, max_length=30, prefix_allowed_tokens_fn=fn,
                           num_beams=gK, num_return_sequences=gK, output_scores=True, return_dict_in_generate=True)
        barrier()
        g0 = time.perf_counter()
        for _ in range(args.gen_batches):
            o = model.generate(input_ids=gids, attention_mask=gmask, whole_word_ids=gww, max_length=30, prefix_allowed_tokens_fn=fn,
                               num_beams=gK, num_return_sequences=gK, output_scores=True, return_dict_in_generate=True)
        barrier()
        gdt = time.perf_counter() - g0
        if world > 1:
            import torch.distributed as dist
            tmax = torch.tensor([gdt], device=device, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            gdt = float(tmax.item())
        gen = {"items_per_s": world * gB * gK * args.gen_batches / gdt, "ms_per_batch": gdt / args.gen_batches * 1e3,
               "users_per_batch": gB, "num_beams": gK, "max_length": 30, "trie_items": 3416,
               "decoded_len": int(o["sequences"].shape[1])}

    if rank == 0:
        c = cfg
        inner = c.num_heads * c.d_kv
        flops = train_flops_per_sample(c.d_model, inner, c.d_ff, c.num_heads, c.num_layers, L, T, V)
        # dominant kernel: forward/dgrad/wgrad bf16 GEMMs; time the FFN up-projection shape [B*L, d] x [d, F] live
        Mg, Ng, Kg = B * L, c.d_ff, c.d_model
        t_k = time_gemm_kernel(be, Mg, Ng, Kg)
        ach = 2.0 * Mg * Ng * Kg / t_k / 1e12
        t_w = time_gemm_kernel(be, Ng, Kg, Mg, wgrad=True)          # dW_i[F, d] over B*L tokens
        ach_w = 2.0 * Mg * Ng * Kg / t_w / 1e12
        line = {
            "metric": "train_samples_per_sec", "value": samples_per_s, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"ML-1M-shaped sequential prompts, {args.backbone} V={V}, B={B}/GPU L={L} T={T}, dropout 0.1, "
                                   "fwd+bwd+clip+AdamW (BASELINE.json configs[1])",
                       "global_batch": world * B, "seq_len": L, "tgt_len": T, "parallelism": f"dp{world}"},
            "final_loss": final_loss,
            "model_tflops": samples_per_s * flops / 1e12,
            "model_flops_frac_of_bf16_peak": samples_per_s * flops / 1e12 / (BF16_PEAK_TFLOPS * world),
            "beam10_items_per_sec": gen["items_per_s"] if gen else None,
            "generation": gen,
            # dominant kernel family = the bf16 MFMA GEMMs (p5_gemm_kernel forward/dgrad instantiations + the ring kernels are
            # ~70 % of the kernel time, profiles/r01_train_t5small_b64_kernel_stats.md); `roofline` times the largest forward
            # shape live, `roofline_wgrad` the same FLOPs as a weight gradient on the ring kernel (in the step that kernel
            # shares the GPU with the main stream, so its in-step launches are longer; profiles/README.md).  `traffic` is
            # the PMC-measured HBM bytes per launch of exactly this kernel+shape (2 x FETCH_SIZE + WRITE_SIZE with the gfx950
            # correction, profiles/r01_pmc_gemm.md) -- a recorded measurement, not collected inside this run.
            "roofline": {"bound": "mfma", "kernel": "p5_gemm_kernel<bf16,128,128,KC,KC,direct-to-LDS>", "shape": [Mg, Ng, Kg],
                         "achieved": ach, "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / BF16_PEAK_TFLOPS,
                         "traffic": 75.0e6 if (Mg, Ng, Kg) == (8192, 2048, 512) else None, "algorithmic_bytes": 2.0 * (Mg * Kg + Ng * Kg + Mg * Ng),
                         "avg_launch_us": t_k * 1e6},
            "roofline_wgrad": {"bound": "mfma", "kernel": "p5_gemm2_kernel<128,128,ring4,KS,KS>", "shape": [Ng, Kg, Mg], "achieved": ach_w,
                               "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_w / BF16_PEAK_TFLOPS,
                               "traffic": 112.0e6 if (Mg, Ng, Kg) == (8192, 2048, 512) else None,   # PMC, profiles/r01_pmc_gemm.md
                               "algorithmic_bytes": 2.0 * (Mg * Kg + Mg * Ng) + 4.0 * Ng * Kg, "avg_launch_us": t_w * 1e6},
        }
        if not args.no_cpu and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
