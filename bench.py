#!/usr/bin/env python
"""bench.py -- BASELINE.json metric on the MI355X: training samples/s (+ beam-10 items/s) of the OpenP5 src_t5 hot path,
T5-small on ML-1M-shaped synthetic data (B=64/GPU, sequential prompts L=128, T=8, bf16,
dropout 0.1, V=32100), one process per GPU.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = forward + runner masked-mean loss + backward (+ gradient all-reduce overlapped with it when N > 1) +
global-norm clip + AdamW + schedule + zero_grad, i.e. DistributedRunner.py:56-93 minus the Python collator (inputs are
resident in HBM before the timed region).  Prints ONE JSON line on rank 0.

Besides the headline (C2) the line carries, at N = 1 only (`--legs none` drops them):
  * `roofline` (dominant kernel of the step by time, profiles/README.md), `roofline_fwd`, `roofline_generation`;
    `traffic` and `mfma_busy` are measured by this run itself when rocprofv3 is on the box (separate `--pmc` subprocess passes over
    tools/gemm_one.py and tools/gen_bench.py, bench.pmc_live); otherwise they come from profiles/pmc_traffic.json with `traffic_stale: true`;
  * `cpu_baseline` (training, oracle port) and `cpu_baseline_generation` (oracle beam search), bounded samples;
  * `legs`: single-GPU measurements of the other BASELINE.json configs -- C3 per-GPU (T5-base, B=64), C4 (T5-base,
    beam 20, 20 users, V=32600), C5 per-GPU (T5-large, L=512) -- and `task_mix`, the real alternation of sequential
    (L~130) and straightforward (L~16) batches through datasets + sampler + collator + prefetch thread
    (SingleMultiDataTaskSampler.py:55-71).
"""
import argparse
import json
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

DIMS = {  # d_model, d_ff, heads, layers
    "t5-small": (512, 2048, 8, 6),
    "t5-base": (768, 3072, 12, 12),
    "t5-large": (1024, 4096, 16, 24),
}


def train_flops_per_sample(d, inner, F, H, NL, L, T, Vv, dk=64):
    """algorithmic FLOPs of one training sample (fwd x3), padding tokens counted."""
    enc = NL * L * 2 * (4 * d * inner + 2 * d * F) + NL * H * 4 * L * L * dk
    dec = NL * T * 2 * (6 * d * inner + 2 * d * F) + NL * L * 2 * (2 * d * inner) + NL * H * 4 * (T * T + T * L) * dk
    head = T * 2 * d * Vv
    return 3.0 * (enc + dec + head)


def gen_bytes_per_step(d, inner, F, NL, Vv, B, K, L, t, sz=2):
    """HBM bytes one decode step must move at least (weights once + shared cross-KV + self-KV so far)."""
    return NL * (6 * d * inner + 2 * d * F) * sz + d * Vv * sz + B * NL * 2 * L * inner * sz + B * K * NL * 2 * t * inner * sz


def gen_flops_per_user(d, inner, F, H, NL, L, K, S, Vv, dk=64):
    enc = NL * L * 2 * (4 * d * inner + 2 * d * F) + NL * H * 4 * L * L * dk
    return enc + NL * L * 2 * (2 * d * inner) + S * K * (NL * 2 * (6 * d * inner + 2 * d * F) + 2 * d * Vv)


def synth_batch(B, L, T, device, seed, vocab=V):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, vocab, (B, L), generator=g)
    lens = torch.randint(int(0.7 * L), L + 1, (B,), generator=g)
    lens[0] = L
    mask = (torch.arange(L)[None, :] < lens[:, None]).long()
    ids = ids * mask
    ww = torch.cumsum((torch.rand(B, L, generator=g) < 0.35).long(), 1) * mask
    ww = ww.clamp(max=511)
    labels = torch.randint(3, vocab, (B, T), generator=g)
    tl = torch.randint(max(2, T - 2), T + 1, (B,), generator=g)
    out_attn = (torch.arange(T)[None, :] < tl[:, None]).long()
    labels = labels * out_attn
    labels[torch.arange(B), tl - 1] = 1
    return [t.to(device) for t in (ids, ww, mask, labels, out_attn)]


def synth_items(n_items, seed, lo=3000, hi=3999, pieces=(2, 2, 3)):
    """ML1M-like item ids: "<dataset> item_ <digits>" -> decoder start + shared 4-piece prefix + 2-3 number pieces + </s>, sorted."""
    rnd = random.Random(seed)
    items = set()
    while len(items) < n_items:
        n = rnd.choice(pieces)
        items.add(tuple([0, 2000, 2001, 2002, 2003] + [rnd.randint(lo, hi) for _ in range(n)] + [1]))
    return sorted(items)


def synth_item_trie(n_items, seed, lo=3000, hi=3999, pieces=(2, 2, 3)):
    from openp5_amd.trie import Trie
    return Trie(synth_items(n_items, seed, lo, hi, pieces))


def _dist():
    import torch.distributed as dist
    return dist


def barrier(world):
    if world > 1:
        _dist().barrier()
    torch.cuda.synchronize()


def max_over_ranks(dt, world, device):
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        _dist().all_reduce(t, op=_dist().ReduceOp.MAX)
        dt = float(t.item())
    return dt


def build_model(backbone, dtype, device, be, world, rank, vocab=V, dropout=0.1, total_steps=1000):
    from openp5_amd.model import P5ModelConfig, P5T5Native
    from openp5_amd.optim import FusedAdamW
    cfg = P5ModelConfig.from_backbone(backbone, vocab_size=vocab, dropout_rate=dropout)
    model = P5T5Native(cfg, dtype=dtype, device=device, backend=be, seed=2023)
    model.ddp_world = world
    opt = FusedAdamW(model, lr=1e-3, eps=1e-6, weight_decay=0.01, max_grad_norm=1.0, warmup_steps=int(0.05 * total_steps), total_steps=total_steps)
    model.set_dropout_seed(2023 + rank, 0)
    return cfg, model, opt


def train_step(model, opt, batch):
    """One runner step (openp5_amd/runner.py train loop body == DistributedRunner.py:56-93 without barriers)."""
    from openp5_amd.runner import training_step
    return training_step(model, opt, batch, alpha=2)


def time_training(model, opt, batch, steps, warmup, world, device):
    model.train()
    last = None
    for _ in range(warmup):
        train_step(model, opt, batch)
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(steps):
        last = train_step(model, opt, batch)
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world, device)
    return dt, float(last.detach())


def profile_training(be, model, opt, batch, steps=3):
    """Per-kernel time of `steps` training steps measured IN THIS RUN by the library's own profiler (p5_profile_begin / p5_profile_end:
    two HIP events around every launch, on the launch stream): a list of {"kernel", "launches", "total_us", "flops"} per (kernel, grid),
    per step.  Durations include each launch's dispatch gap, so they are upper bounds of the rocprofv3 kernel durations."""
    import ctypes
    model.train()
    train_step(model, opt, batch)
    torch.cuda.synchronize()
    be.check(be.lib.p5_profile_begin(), "p5_profile_begin")
    for _ in range(steps):
        train_step(model, opt, batch)
    buf = ctypes.create_string_buffer(1 << 20)
    be.check(be.lib.p5_profile_end(buf, len(buf)), "p5_profile_end")
    rows = json.loads(buf.value.decode() or "[]")
    for r in rows:
        r["launches_per_step"] = r["launches"] / steps
        r["us_per_step"] = r["total_us"] / steps
        r["flops_per_step"] = r["flops"] / steps
    return rows


def kernel_classes(rows):
    """aggregate the profiler's (kernel, grid) rows by kernel name (+ tag): what a rocprofv3 --stats table shows per kernel."""
    cls = {}
    for r in rows:
        name = r["kernel"].split(" grid=")[0]
        c = cls.setdefault(name, {"kernel": name, "launches_per_step": 0.0, "us_per_step": 0.0, "flops_per_step": 0.0, "grids": []})
        c["launches_per_step"] += r["launches_per_step"]
        c["us_per_step"] += r["us_per_step"]
        c["flops_per_step"] += r["flops_per_step"]
        c["grids"].append({"grid": r["kernel"].split(" grid=")[1], "launches_per_step": r["launches_per_step"], "avg_us": r["total_us"] / max(1, r["launches"]),
                           "tflops": r["flops"] / max(r["total_us"], 1e-9) / 1e6})
    return sorted(cls.values(), key=lambda c: -c["us_per_step"])


def time_generation(model, gB, gK, L, trie, max_length, batches, world, device, seed, vocab=V, mode=None):
    """`batches` generate() calls of `gB` users, beam `gK`, bracketed like the training step (barrier + synchronize).  mode (bf16 models):
    "verified" = the bf16 search with extra beams proposes, one fp32 pass decides (lists and scores are the fp32 search's: csrc/p5_verify.h),
    "draft" = the plain bf16 search.  Returns (seconds, decoded length, engine timing of the search kernels, per-call median ms, stats)."""
    from openp5_amd.trie import prefix_allowed_tokens_fn
    model.eval()
    if mode is not None:
        model.generation_mode = mode
    fn = prefix_allowed_tokens_fn(trie)
    gids, gww, gmask, _, _ = synth_batch(gB, L, 8, device, seed, vocab)
    kw = dict(input_ids=gids, attention_mask=gmask, whole_word_ids=gww, max_length=max_length, prefix_allowed_tokens_fn=fn,
              num_beams=gK, num_return_sequences=gK, output_scores=True, return_dict_in_generate=True)
    for _ in range(3):
        o = model.generate(**kw)
    for k in model.verify_stats:
        model.verify_stats[k] = 0
    lanes = int(getattr(model, "gen_lanes", 1))
    for _ in model.map_lanes(lambda k: model.generate(**k), [kw] * (2 * lanes), lanes=lanes):      # (lane engines, workspaces, graphs: built outside the timed region)
        pass
    for k in model.verify_stats:
        model.verify_stats[k] = 0
    barrier(world)
    g0 = time.perf_counter()
    # `batches` evaluation batches through the model's generation lanes, as the runner's evaluation loop does (P5T5Native.map_lanes: up to
    # `gen_lanes` batches in flight, each on its own engines / workspaces / HIP stream over the one set of weights; results in order)
    for o in model.map_lanes(lambda k: model.generate(**k), [kw] * batches, lanes=lanes):
        pass
    barrier(world)
    gdt = max_over_ranks(time.perf_counter() - g0, world, device)
    stats = dict(model.verify_stats)
    # per-call wall times (each call ends with the read of the result lengths, i.e. is synchronous) and the device time of the search's
    # decode loop alone (engine-side HIP events around it), both on extra calls OUTSIDE the timed region
    per_call = []
    for _ in range(max(5, min(20, batches))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.generate(**kw)
        torch.cuda.synchronize()
        per_call.append((time.perf_counter() - t0) * 1e3)
    timing = None
    if hasattr(model, "time_generate"):
        model.time_generate(True)
        enc, dec, ff = [], [], 0
        for _ in range(3):
            model.generate(**kw)
            t = model.last_generate_timing()
            enc.append(t["encode_ms"])
            dec.append(t["decode_ms"])
            ff = int(t["forced_prefix_steps"])
        model.time_generate(False)
        timing = {"encode_ms": sorted(enc)[1], "decode_ms": sorted(dec)[1], "forced_prefix_steps": ff,
                  "source": "HIP events recorded by p5_generate around its decode loop, median of 3 calls; encode_ms = encoder pass (or the cast of the "
                            "verification pass's encoder output) + the forced-prefix pass + cross-attention K/V + beam state"}
    stats["lanes"] = lanes
    return gdt, int(o["sequences"].shape[1]), timing, sorted(per_call)[len(per_call) // 2], stats


def trained_generation_leg(be, device, backbone, dtype, world, rank, gB, gK, L, n_items, gen_batches, train_steps=1500, clusters=8, head=100):
    """The generation headline on TRAINED weights (round-5 verdict 1b): random-init weights score every item within ~1e-3 of every other,
    weights trained on noise likewise; a recommender's scores are peaked.  A fresh model of the benchmarked architecture is trained with the
    benchmarked training step on a learnable synthetic task -- the first input token names one of `clusters` user clusters, the target is an
    item id drawn from that cluster's own Zipf(1.2) popularity over `head` items (cluster-conditional popularity, what a sequential
    recommender learns first) -- then the verified beam search is timed on users of the same distribution.  Reports items/s, the verification
    statistics (flagged users), the loss reached, and how often the top-1 item is one of the user's cluster's ten most popular items."""
    cfg, model, opt = build_model(backbone, dtype, device, be, world, rank, total_steps=train_steps)
    items = synth_items(n_items, 7)
    T = max(len(it) for it in items) - 1
    g = torch.Generator().manual_seed(4242 + rank)
    perms = [torch.randperm(n_items, generator=g)[:head] for _ in range(clusters)]
    w = 1.0 / torch.arange(1, head + 1, dtype=torch.float64) ** 1.2
    item_tok = torch.zeros(n_items, T, dtype=torch.long)
    for i, it in enumerate(items):
        item_tok[i, :len(it) - 1] = torch.tensor(it[1:])

    def batch(B, seed):
        ids, ww, mask, _, _ = synth_batch(B, L, T, "cpu", seed)
        gg = torch.Generator().manual_seed(seed)
        cl = torch.randint(0, clusters, (B,), generator=gg)
        ids[:, 0] = 100 + cl
        rank_in_cluster = torch.multinomial(w, B, replacement=True, generator=gg)
        tgt = torch.stack([perms[int(c)][int(r)] for c, r in zip(cl, rank_in_cluster)])
        labels = item_tok[tgt]
        return [t.to(device) for t in (ids, ww, mask, labels, (labels != 0).long())], cl
    model.train()
    pool = [batch(64, 9000 + i)[0] for i in range(32)]
    loss = None
    for st in range(train_steps):
        loss = train_step(model, opt, pool[st % len(pool)])
    final_loss = float(loss.detach())
    (gids, gww, gmask, _, _), cl = batch(gB, 777 + rank)
    from openp5_amd.trie import Trie, prefix_allowed_tokens_fn
    fn = prefix_allowed_tokens_fn(Trie(items))
    model.eval()
    model.generation_mode = "verified"
    kw = dict(input_ids=gids, attention_mask=gmask, whole_word_ids=gww, max_length=30, prefix_allowed_tokens_fn=fn, num_beams=gK, num_return_sequences=gK,
              output_scores=True, return_dict_in_generate=True)
    lanes = int(getattr(model, "gen_lanes", 1))
    for _ in model.map_lanes(lambda k: model.generate(**k), [kw] * (2 * lanes), lanes=lanes):
        pass
    for k in model.verify_stats:
        model.verify_stats[k] = 0
    barrier(world)
    t0 = time.perf_counter()
    for o in model.map_lanes(lambda k: model.generate(**k), [kw] * gen_batches, lanes=lanes):
        pass
    barrier(world)
    gdt = max_over_ranks(time.perf_counter() - t0, world, device)
    vst = dict(model.verify_stats)
    seq = o["sequences"].view(gB, gK, -1)[:, 0].cpu()
    sc = o["sequences_scores"].view(gB, gK).cpu()
    index = {it[1:]: i for i, it in enumerate(items)}
    hits = 0
    for b in range(gB):
        toks = tuple(t for t in seq[b].tolist()[1:] if t != 0)
        hits += int(index.get(toks, -1) in set(perms[int(cl[b])][:10].tolist()))
    return {"items_per_s": world * gB * gK * gen_batches / gdt, "ms_per_batch": gdt / gen_batches * 1e3, "lanes": lanes, "verify_stats": vst,
            "fallback_users": vst.get("fallback_users", 0), "escalated_users": vst.get("escalated_users", 0), "users": vst.get("users", 0),
            "train_steps": train_steps, "final_train_loss": final_loss, "top1_in_cluster_top10": hits / gB,
            "score_gap_top1_top10": float((sc[:, 0] - sc[:, -1]).mean()),
            "note": f"fresh {backbone} trained for {train_steps} benchmark steps (B=64, dropout 0.1, lr 1e-3 linear warm-up) on cluster-conditional Zipf(1.2) item "
                    f"popularity ({clusters} clusters x {head} items of the {n_items}-item trie), then verified beam-{gK} generation for {gB} users per batch"}


def time_gemm_kernel(be, M, N, K, iters=50, wgrad=False):
    """average duration of a GEMM kernel measured with HIP events on the stream the kernel is launched on.
    wgrad=False: forward bf16 GEMM y = x W^T (both operands K-contiguous);
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


def time_wgrad_group(be, M, d, F, inner, iters=30, layers=2):
    """average duration of the dominant kernel of the step, as the step launches it: the weight gradients of TWO encoder layers
    (per layer dW_o[d,F], dW_i[F,d], dW_attn_o[d,inner], dW_qkv[3 inner,d], each reduced over all M tokens) as ONE launch of the
    wave-specialised persistent kernel (p5_gemm5_kernel<KS>, 192 tiles of 256x128, no split-K; the epilogue stores, as on the first
    micro-batch of a step), HIP events on the launch stream.  Returns (seconds, flops, bytes)."""
    import ctypes
    from openp5_amd._abi import P5GemmProblem
    shapes = [(d, F), (F, d), (d, inner), (3 * inner, d)] * layers
    arr = (P5GemmProblem * len(shapes))()
    keep = []
    flops = byts = 0.0
    for i, (n_out, k_in) in enumerate(shapes):
        A = torch.randn(M, n_out, device=be.device).to(torch.bfloat16)
        Bm = torch.randn(M, k_in, device=be.device).to(torch.bfloat16)
        C = torch.zeros(n_out, k_in, device=be.device, dtype=torch.float32)
        keep += [A, Bm, C]
        q = arr[i]
        q.A, q.B, q.C, q.aux = A.data_ptr(), Bm.data_ptr(), C.data_ptr(), None
        q.M, q.N, q.K, q.lda, q.ldb, q.ldc, q.ldaux = n_out, k_in, M, n_out, k_in, k_in, 0
        q.epi, q.c_f32, q.splitk, q.alpha = 0, 1, 1, 1.0
        q.rowss, q.rowss_eps, q.ssq_out = None, 0.0, None
        flops += 2.0 * M * n_out * k_in
        byts += 2.0 * M * (n_out + k_in) + 4.0 * n_out * k_in      # both operands once (bf16) + the fp32 gradient written once
    s = torch.cuda.current_stream()
    call = lambda: be.lib.p5_op_gemm_group(1, 1, len(shapes), arr, None, 0, 0.0, be.stream_ptr())  # noqa: E731  (tile config 1 = 256x128, as wgrad_flush picks)
    for _ in range(5):
        be.check(call(), "gemm_group")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(iters):
        call()
    e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3, flops, byts


def _rocprof_pmc_pass(counters, cmd, timeout_s=150):
    """One `rocprofv3 --pmc <counters> --kernel-trace -- <cmd>` pass (kernel trace only next to --pmc, as MI355X_MICROARCH.md and gpurun
    require) -> {kernel_name: {counter: (sum over dispatches, dispatches)}}, or None when rocprofv3 is absent / the pass fails."""
    import glob, shutil, sqlite3, subprocess, tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    out = tempfile.mkdtemp(prefix="p5pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", PYTHONDONTWRITEBYTECODE="1")
    try:
        r = subprocess.run([exe, "--pmc", *counters, "--kernel-trace", "-d", out, "-o", "p", "--"] + cmd, cwd="/tmp", env=env, timeout=timeout_s,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dbs = glob.glob(os.path.join(out, "**", "*_results.db"), recursive=True)
        if r.returncode != 0 or not dbs:
            return None
        cur = sqlite3.connect(dbs[0]).cursor()
        tabs = [t[0] for t in cur.execute("select name from sqlite_master where type='table'")]
        ev = [t for t in tabs if "pmc_event" in t][0]; info = [t for t in tabs if "info_pmc" in t][0]
        disp = [t for t in tabs if "kernel_dispatch" in t][0]; sym = [t for t in tabs if "info_kernel_symbol" in t][0]
        res = {}
        for k, n, v, c in cur.execute(f"select s.kernel_name, i.name, sum(e.value), count(*) from {ev} e join {info} i on e.pmc_id=i.id join {disp} d on "
                                      f"e.event_id=d.event_id join {sym} s on d.kernel_id=s.id group by s.kernel_name, i.name"):
            res.setdefault(k, {})[n] = (float(v), int(c))
        return res
    except Exception:
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def pmc_live(shape, want_decode=True):
    """HBM-side traffic and matrix-core occupancy of the roofline kernels measured IN THIS RUN (round-5 verdict item 5): separate --pmc passes
    over tools/gemm_one.py (the forward / data-gradient GEMM at `shape`) -- FETCH_SIZE, WRITE_SIZE (they do not fit one pass; FETCH_SIZE
    doubled on gfx950, MI355X_MICROARCH.md HBM section) and the SQ activity counters -- and over tools/gen_bench.py (plain bf16 search) for
    the bytes of one decode step.  Returns {} when rocprofv3 is not on the box."""
    py = sys.executable
    gemm = [py, os.path.join(ROOT, "tools", "gemm_one.py")] + [str(int(x)) for x in shape]
    pick = lambda res, pat: next((v for k, v in (res or {}).items() if pat in k), None)      # noqa: E731
    out = {}
    f = pick(_rocprof_pmc_pass(["FETCH_SIZE"], gemm), "p5_gemm5_kernelILb0E")
    w = pick(_rocprof_pmc_pass(["WRITE_SIZE"], gemm), "p5_gemm5_kernelILb0E")
    if f and w and "FETCH_SIZE" in f and "WRITE_SIZE" in w:
        rd = f["FETCH_SIZE"][0] / f["FETCH_SIZE"][1] * 1024.0 * 2.0
        wr = w["WRITE_SIZE"][0] / w["WRITE_SIZE"][1] * 1024.0
        out["gemm"] = {"traffic_bytes": rd + wr, "read_bytes": rd, "write_bytes": wr, "launches": f["FETCH_SIZE"][1],
                       "source": "this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel trace only) over tools/gemm_one.py, per launch; "
                                 "FETCH_SIZE x2 (gfx950); fabric-side counters (Infinity-Cache hits included): an upper bound of HBM traffic"}
    q = pick(_rocprof_pmc_pass(["SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"], gemm), "p5_gemm5_kernelILb0E")
    if q and "SQ_BUSY_CYCLES" in q and "SQ_VALU_MFMA_BUSY_CYCLES" in q:
        busy, mf = q["SQ_BUSY_CYCLES"][0], q["SQ_VALU_MFMA_BUSY_CYCLES"][0]
        wc = q.get("SQ_WAVE_CYCLES", (0.0, 1))[0]
        out["sq"] = {"mfma_busy": mf / max(1.0, 32.0 * busy), "wait_frac_of_wave_cycles": q.get("SQ_WAIT_ANY", (0.0, 1))[0] / max(1.0, wc),
                     "issue_frac_of_wave_cycles": q.get("SQ_ACTIVE_INST_ANY", (0.0, 1))[0] / max(1.0, wc),
                     "source": "this run: rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY over tools/gemm_one.py; "
                               "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 8 x SQ_BUSY_CYCLES) (profiles/pmc_summary.py)"}
    if want_decode:
        step_kernels = ("p5_skinny_gemm_kernel", "p5_dec_self_attn2_kernel", "p5_dec_cross_attn3_kernel", "p5_head_lse_kernel", "p5_dec_score2_kernel",
                        "p5_beam_step_kernel", "p5_rmsnorm_f32in_kernel")
        gen = [py, os.path.join(ROOT, "tools", "gen_bench.py"), "20", "5", "10"]
        tot = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            os.environ["P5_GEN_MODE"] = "draft"
            res = _rocprof_pmc_pass([ctr], gen)
            os.environ.pop("P5_GEN_MODE", None)
            if not res:
                tot = None
                break
            v = sum(c[ctr][0] for k, c in res.items() if ctr in c and any(sk in k for sk in step_kernels))
            steps = sum(c[ctr][1] for k, c in res.items() if ctr in c and "p5_beam_step_kernel" in k)
            tot[ctr] = (v, steps)
        if tot and tot["FETCH_SIZE"][1] > 0 and tot["FETCH_SIZE"][1] == tot["WRITE_SIZE"][1]:
            st = tot["FETCH_SIZE"][1]
            rd, wr = tot["FETCH_SIZE"][0] * 1024.0 * 2.0 / st, tot["WRITE_SIZE"][0] * 1024.0 / st
            out["decode"] = {"traffic_bytes_per_step": rd + wr, "read_bytes_per_step": rd, "write_bytes_per_step": wr, "decode_steps": st,
                             "source": "this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/gen_bench.py (plain bf16 search, 20 users, beam 10), "
                                       "summed over every kernel of the decode steps, per p5_beam_step_kernel dispatch; FETCH_SIZE x2 (gfx950)"}
    return out


def in_step_us(kernel_key):
    """average in-step duration (us) of a kernel from profiles/in_step.json (written by profiles/summarize_rocpd.py --in-step from the
    rocprofv3 kernel trace of `bench.py` itself, where the kernel shares the GPU with the other stream) -- None when not collected."""
    path = os.path.join(ROOT, "profiles", "in_step.json")
    try:
        with open(path) as f:
            return json.load(f).get(kernel_key)
    except Exception:
        return None


def decode_step_traffic():
    """bytes per decode step from profiles/pmc_decode_step.json (written on the GPU box by tools/r5_final_run.sh), or None"""
    path = os.path.join(ROOT, "profiles", "pmc_decode_step.json")
    try:
        return float(json.load(open(path))["traffic_bytes_per_step"])
    except Exception:
        return None


def pmc_traffic(kernel, shape):
    """HBM bytes per launch of `kernel` at `shape` from profiles/pmc_traffic.json (profiles/collect_pmc.sh: separate FETCH_SIZE /
    WRITE_SIZE passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) -- None when not collected."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            tab = json.load(f)
        ent = tab.get(f"{kernel}|{'x'.join(str(int(s)) for s in shape)}")
        return float(ent["traffic_bytes"]) if ent else None
    except Exception:
        return None


def cpu_baseline_train_hf(seconds_budget=15.0):
    """The reference's own CPU path where it can be built: STOCK HuggingFace T5ForConditionalGeneration (installed transformers,
    eager attention, fp32) fed `shared(ids) + whole_word(ww)` as P5_T5.forward does (P5_T5.py:94-100), CE(reduction="none") + the
    runner's masked mean (DistributedRunner.py:72-77), clip_grad_norm_ + HF-AdamW semantics + step.  None if transformers is missing."""
    try:
        from oracle import hf_ref, t5_oracle as O
        cfg = O.T5Cfg.named("t5-small", dropout=0.0)
        params = O.init_params(cfg, 2023)
        m, wwe = hf_ref.build_hf(cfg, params)
    except Exception:
        return None
    ncores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(ncores)
    m.train()
    plist = [p for p in list(m.parameters()) + list(wwe.parameters()) if p.requires_grad]
    Mm = [torch.zeros_like(p) for p in plist]
    Vv = [torch.zeros_like(p) for p in plist]
    ids, ww, mask, labels, out_attn = synth_batch(4, 128, 8, "cpu", 1)
    times, t_start, step = [], time.time(), 0
    while True:
        t0 = time.time()
        nll, _ = hf_ref.hf_forward_nll(m, wwe, ids, ww, mask, labels)
        loss = O.runner_loss(nll, out_attn)
        grads = torch.autograd.grad(loss, plist, allow_unused=True)
        grads = [g if g is not None else torch.zeros_like(p) for g, p in zip(grads, plist)]
        _, coef = O.clip_coef(grads, 1.0)
        step += 1
        with torch.no_grad():
            for p, g, mm, vv in zip(plist, grads, Mm, Vv):
                O.adamw_hf_step(p, g * coef, mm, vv, step, 1e-3)
        times.append(time.time() - t0)
        if (time.time() - t_start > seconds_budget and len(times) >= 2) or len(times) >= 12 or time.time() - t_start > 3 * seconds_budget:
            break
    rest = times[1:] if len(times) > 1 else times
    med = sorted(rest)[len(rest) // 2]
    import transformers
    return {"value": 4.0 / med, "unit": "samples/s", "cores": ncores, "kind": "reference",
            "sample": f"{len(times)} train steps of stock HF T5ForConditionalGeneration (transformers {transformers.__version__}, eager, fp32 T5-small, B=4, L=128, T=8) "
                      "driven as P5_T5.forward + the runner's loss / clip / AdamW do, median of all but the first"}


def cpu_baseline_train(seconds_budget=15.0):
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


def cpu_baseline_generation(seconds_budget=15.0):
    """items/s of the CPU path: the oracle's restatement of HF beam search + per-row Python trie callbacks
    (DistributedRunner.py:361-371 semantics) on the host cores: 4 users x beam 10 over the same 3416-item trie."""
    from oracle import t5_oracle as O
    ncores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(ncores)
    cfg = O.T5Cfg.named("t5-small", dropout=0.0)
    P = O.init_params(cfg, 2023)
    trie = synth_item_trie(3416, 7)
    ids, ww, mask, _, _ = synth_batch(4, 128, 8, "cpu", 500)
    times = []
    t_start = time.time()
    with torch.no_grad():
        while True:
            t0 = time.time()
            O.beam_search(P, cfg, ids, ww, mask, lambda b, s: trie.get(s.tolist()), 10, 30)
            times.append(time.time() - t0)
            if (time.time() - t_start > seconds_budget and len(times) >= 2) or len(times) >= 6 or time.time() - t_start > 3 * seconds_budget:
                break
    rest = times[1:] if len(times) > 1 else times
    med = sorted(rest)[len(rest) // 2]
    return {"value": 4 * 10 / med, "unit": "items/s", "cores": ncores, "kind": "port",
            "sample": f"{len(times)} calls of oracle beam_search (fp32 T5-small, 4 users x beam 10, L=128, 3416-item trie, max_length 30), "
                      "median of all but the first"}


def task_mix_leg(be, device, steps=60, warmup=10):
    """The real task mix: ML1M-style prompts (max_his 20) from a synthetic 400-user dataset through MultiTaskDataset ->
    SingleMultiDataTaskSampler (alternating sequential / straightforward batches of 64) -> Collator in the prefetch thread ->
    pinned H2D -> training step.  Collation is INSIDE the timed region here."""
    import tempfile
    from torch.utils.data import ConcatDataset, DataLoader
    from openp5_amd.collator import Collator
    from openp5_amd.data import MultiTaskDataset
    from openp5_amd.runner import Prefetcher, build_arg_parser
    from openp5_amd.sampler import SingleMultiDataTaskSampler
    from openp5_amd.synth import write_dataset, write_prompt_file
    from openp5_amd.tokenizer import build_offline_tokenizer
    tmp = tempfile.mkdtemp(prefix="p5bench_")
    write_dataset(os.path.join(tmp, "data"), "ML1M", n_users=400, n_items=3416, n_inter=400 * 60)
    prompt = write_prompt_file(os.path.join(tmp, "prompt.txt"))
    args = build_arg_parser().parse_args(["--data_path", os.path.join(tmp, "data"), "--datasets", "ML1M", "--tasks", "sequential,straightforward",
                                          "--item_indexing", "sequential", "--prompt_file", prompt, "--sample_prompt", "1", "--sample_num", "1,1",
                                          "--max_his", "20", "--distributed", "0", "--batch_size", "64"])
    args.rank = 0
    random.seed(0)
    tok = build_offline_tokenizer()
    train = ConcatDataset([MultiTaskDataset(args, "ML1M", "train")])
    loader = DataLoader(train, sampler=SingleMultiDataTaskSampler(train, 64, args.seed), batch_size=64, collate_fn=Collator(tok))
    _, model, opt = build_model("t5-small", "bf16", device, be, 1, 0)
    model.train()
    n, lens, t0 = 0, [], None
    for i, b in enumerate(Prefetcher(loader, pin=True)):
        if i == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if i >= warmup + steps:
            break
        b = [t.to(device, non_blocking=True) if torch.is_tensor(t) else t for t in b]
        train_step(model, opt, (b[0], b[2], b[1], b[3], b[4]))
        if i >= warmup:
            n += b[0].shape[0]
            lens.append(int(b[0].shape[1]))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"samples_per_s": n / dt, "ms_per_step": dt / max(1, len(lens)) * 1e3, "steps": len(lens), "batch": 64,
            "L_sequential_mean": sum(x for x in lens if x > 40) / max(1, sum(1 for x in lens if x > 40)),
            "L_straightforward_mean": sum(x for x in lens if x <= 40) / max(1, sum(1 for x in lens if x <= 40)),
            "note": "collator + H2D inside the timed region; alternating sequential/straightforward batches (SingleMultiDataTaskSampler.py:55-71)"}


def config_legs(be, device, dtype):
    """Single-GPU measurements of BASELINE.json configs[2..4]."""
    legs = {}
    # C3 per GPU: T5-base, B=64, L=128, T=8
    for key, backbone, B, L, T, steps, warm in (("c3_t5base_b64_per_gpu", "t5-base", 64, 128, 8, 8, 3),
                                                ("c5_t5large_b64_l512_per_gpu", "t5-large", 64, 512, 10, 4, 2)):
        try:
            cfg, model, opt = build_model(backbone, dtype, device, be, 1, 0)
            batch = synth_batch(B, L, T, device, 100)
            dt, loss = time_training(model, opt, batch, steps, warm, 1, device)
            d, F, H, NL = DIMS[backbone]
            fl = train_flops_per_sample(d, H * 64, F, H, NL, L, T, V)
            sps = B * steps / dt
            legs[key] = {"samples_per_s": sps, "ms_per_step": dt / steps * 1e3, "steps": steps, "batch": B, "seq_len": L, "tgt_len": T,
                         "model_tflops": sps * fl / 1e12, "frac_of_bf16_peak": sps * fl / 1e12 / BF16_PEAK_TFLOPS, "final_loss": loss}
            if backbone == "t5-base":
                # C4: Beauty collaborative indexing, T5-base, beam 20, eval batch 20, vocabulary grown by 500 <CIk> tokens
                pass
            del model, opt
            torch.cuda.empty_cache()
        except Exception as ex:      # a leg must never take the headline down
            legs[key] = {"error": repr(ex)[:300]}
    try:
        # C2 generation in the fp32 parity mode: the mode whose ranked lists are identical to the fp32 oracle's for every user
        # (tests/test_gpu_dataset.py); the bf16 headline number is the fast mode, whose lists may differ where the oracle's own
        # decision margins are below the bf16 score tolerance
        cfg, model, opt = build_model("t5-small", "fp32", device, be, 1, 0)
        gdt, dec_len, timing, med, _ = time_generation(model, 20, 10, 128, synth_item_trie(3416, 7), 30, 20, 1, device, 500)
        legs["c2_generation_fp32_parity_mode"] = {"items_per_s": 20 * 10 * 20 / gdt, "ms_per_batch": gdt / 20 * 1e3, "ms_per_batch_median_call": med,
                                                  "batches": 20, "users_per_batch": 20, "num_beams": 10, "decoded_len": dec_len, "dtype": "f32", "timing_ms": timing}
        # the fp32 parity engine's TRAINING step at C2 (the reference trains in fp32; exact-fp32 MFMA chain, no folded norms, ungrouped weight gradients)
        batch = synth_batch(64, 128, 8, device, 100)
        tdt, tloss = time_training(model, opt, batch, 5, 2, 1, device)
        legs["c2_training_fp32_parity_mode"] = {"samples_per_s": 64 * 5 / tdt, "ms_per_step": tdt / 5 * 1e3, "steps": 5, "batch": 64, "dtype": "f32", "final_loss": tloss}
        del opt
        del model
        torch.cuda.empty_cache()
    except Exception as ex:
        legs["c2_generation_fp32_parity_mode"] = {"error": repr(ex)[:300]}
    try:
        Vc = V + 500
        cfg, model, _ = build_model("t5-base", dtype, device, be, 1, 0, vocab=Vc)
        trie = synth_item_trie(12101, 11, lo=V, hi=Vc - 1, pieces=(2, 3, 3, 4))       # Beauty: 12,101 items, <CIk> token paths
        gB, gK, L = 20, 20, 128
        gdt, dec_len, timing, _, vst = time_generation(model, gB, gK, L, trie, 30, 5, 1, device, 700, vocab=Vc, mode="verified")
        gdt_d, _, timing_d, _, _ = time_generation(model, gB, gK, L, trie, 30, 5, 1, device, 700, vocab=Vc, mode="draft")
        d, F, H, NL = DIMS["t5-base"]
        S = dec_len - 1
        ms = gdt / 5 * 1e3
        legs["c4_t5base_beam20_b20"] = {"items_per_s": gB * gK * 5 / gdt, "ms_per_batch": ms, "users_per_batch": gB, "num_beams": gK, "vocab": Vc,
                                        "trie_items": 12101, "decoded_len": dec_len, "timing_ms": timing, "mode": "verified", "verify_stats": vst,
                                        "plain_bf16": {"items_per_s": gB * gK * 5 / gdt_d, "ms_per_batch": gdt_d / 5 * 1e3, "timing_ms": timing_d},
                                        "hbm_bytes_per_step_min": gen_bytes_per_step(d, H * 64, F, NL, Vc, gB, gK, L, S // 2 + 1),
                                        "gflop_per_batch": gB * gen_flops_per_user(d, H * 64, F, H, NL, L, gK, S, Vc) / 1e9}
        del model
        torch.cuda.empty_cache()
    except Exception as ex:
        legs["c4_t5base_beam20_b20"] = {"error": repr(ex)[:300]}
    return legs


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
    ap.add_argument("--legs", default="all", help="all | none | comma list of: configs,task_mix")
    ap.add_argument("--gen-batches", type=int, default=20)
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc subprocess passes (roofline.traffic / mfma_busy then come from profiles/*.json, marked stale)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        _dist().init_process_group("nccl", device_id=device)

    from openp5_amd._lib import hip_backend
    be = hip_backend(device)
    B, L, T = args.batch, args.seq_len, args.tgt_len
    cfg, model, opt = build_model(args.backbone, args.dtype, device, be, world, rank, total_steps=max(1000, args.steps + args.warmup))
    batch = synth_batch(B, L, T, device, 100 + rank)
    dt, final_loss = time_training(model, opt, batch, args.steps, args.warmup, world, device)
    samples_per_s = world * B * args.steps / dt
    prof_rows = profile_training(be, model, opt, batch) if (rank == 0 and world == 1) else None      # outside the timed region
    staged = None
    if world == 1 and args.legs != "none":
        # the data-parallel code path on ONE GPU: stage-by-stage backward (16 engine calls, gradient ranges reported per stage, the same
        # two-layer weight-gradient groups), no collective -- what staging itself costs against the single-call backward timed above
        model.staged_backward = True
        sdt, _ = time_training(model, opt, batch, args.steps, 2, 1, device)
        model.staged_backward = False
        staged = {"ms_per_step": sdt / args.steps * 1e3, "samples_per_s": B * args.steps / sdt, "steps": args.steps,
                  "note": "p5_backward_stage x 16 + p5_backward_final_range per stage, world size 1, no all-reduce"}
    comm_wait_ms = None
    if world > 1:
        # device time the main stream waits for the gradient exchange after the last backward stage (extra steps outside the timed region)
        model.ddp_timing, model.ddp_wait_ms = True, []
        for _ in range(5):
            train_step(model, opt, batch)
        torch.cuda.synchronize()
        w = sorted(a.elapsed_time(b) for a, b in model.ddp_wait_ms)
        comm_wait_ms = w[len(w) // 2] if w else None
        model.ddp_timing = False

    # ---- beam-10 constrained generation: items/s (B=20 users/GPU, K=10, ML1M-sized trie of 3416 items) ----
    # HEADLINE = the bf16 model's default "verified" mode: ranked lists and scores are the fp32 search's (tests/test_gpu_dataset.py holds it to
    # the fp32 criteria against the oracle); the plain bf16 search ("draft": lists may differ at near-ties) is reported next to it.
    gen = gen_draft = None
    if not args.no_gen:
        gB, gK = 20, 10
        trie = synth_item_trie(3416, 7)
        # random-init weights of the architecture (as every generation number of rounds 1-4's fp32 leg and tools/gen_bench.py): the model the
        # training loop above has just pushed through 25 steps on NOISE scores every item within ~1e-3 of every other (loss ~ ln V), a regime in
        # which no lower-precision proposer can know the fp32 top-K -- that number is reported too (`generation_after_noise_training`)
        trained = model
        _, model, _ = build_model(args.backbone, args.dtype, device, be, world, rank)
        gdt, dec_len, timing, med, vst = time_generation(model, gB, gK, L, trie, 30, args.gen_batches, world, device, 500 + rank, mode="verified")
        gen = {"mode": "verified (bf16 search with 6 extra beams proposes, one teacher-forced fp32 pass decides; csrc/p5_verify.h)",
               "items_per_s": world * gB * gK * args.gen_batches / gdt, "ms_per_batch": gdt / args.gen_batches * 1e3, "ms_per_batch_median_call": med,
               "lanes": vst.get("lanes"), "timing_note": "ms_per_batch = timed region / batches with `lanes` batches in flight (throughput); ms_per_batch_median_call = one "
               "generate() call by itself (latency)", "users_per_batch": gB, "num_beams": gK, "max_length": 30, "trie_items": 3416, "decoded_len": dec_len, "draft_timing_ms": timing,
               "verify_stats": vst, "weights": "random-init (near-flat item scores; `generation.trained_model` is the same call on trained weights)",
               "fallback_users": vst.get("fallback_users", 0), "escalated_users": vst.get("escalated_users", 0), "users": vst.get("users", 0)}
        gdt, dec_len, timing, med, _ = time_generation(model, gB, gK, L, trie, 30, args.gen_batches, world, device, 500 + rank, mode="draft")
        gen_draft = {"mode": "draft (plain bf16 search)", "items_per_s": world * gB * gK * args.gen_batches / gdt, "ms_per_batch": gdt / args.gen_batches * 1e3,
                     "ms_per_batch_median_call": med, "users_per_batch": gB, "num_beams": gK, "decoded_len": dec_len, "timing_ms": timing}
        model.generation_mode = "verified"
        # (the secondary generation legs -- the noise-trained model, the trained model -- are single-GPU extras: a multi-rank run keeps to the
        #  training metric and the two headline generation numbers, so that no rank-local failure inside an extra can strand the others at a barrier)
        if world == 1:
          gdt, _, _, med, vst2 = time_generation(trained, gB, gK, L, trie, 30, 5, world, device, 500 + rank, mode="verified")
          gen["after_noise_training"] = {"items_per_s": world * gB * gK * 5 / gdt, "ms_per_batch": gdt / 5 * 1e3, "verify_stats": vst2,
                                       "note": "the same call on the model the timed training steps left behind (25 steps on random labels): near-uniform item scores, "
                                               "so the fp32 top-K is not among the bf16 draft's beams for some users; they get a wider draft, then the fp32 search"}
        model = trained
        if world == 1:
            try:
                gen["trained_model"] = trained_generation_leg(be, device, args.backbone, args.dtype, world, rank, gB, gK, L, 3416, args.gen_batches)
            except Exception as ex:
                gen["trained_model"] = {"error": repr(ex)[:300]}

    if rank == 0:
        c = cfg
        inner = c.num_heads * c.d_kv
        flops = train_flops_per_sample(c.d_model, inner, c.d_ff, c.num_heads, c.num_layers, L, T, V)
        Mg, Ng, Kg = B * L, c.d_ff, c.d_model
        t_k = time_gemm_kernel(be, Mg, Ng, Kg)
        ach = 2.0 * Mg * Ng * Kg / t_k / 1e12
        t_w, fl_w, by_w = time_wgrad_group(be, Mg, c.d_model, c.d_ff, inner)
        ach_w = fl_w / t_w / 1e12
        k_fwd = "p5_gemm5_kernel<KC> (256x128 tiles, 4 loader + 4 compute waves, ring of 3 K-steps)"
        k_wg = "p5_gemm5_kernel<KS> (256x128 tiles, 4 loader + 4 compute waves): the 8 weight gradients of two encoder layers in one launch"
        ddp = None
        if world > 1:
            half = str(getattr(model, "ddp_bucket_dtype", "fp32")).replace("torch.", "") in ("bf16", "bfloat16")
            ddp = {"world_size": _dist().get_world_size(), "backend": _dist().get_backend(), "collective": "all_reduce(SUM) of contiguous gradient-arena "
                   "buckets, one per backward stage, issued behind the stage on the side stream",
                   "allreduce_bytes_per_step_per_rank": int(model._n) * (2 if half else 4), "bucket_dtype": "bf16" if half else "fp32",
                   "comm_wait_ms_after_backward": comm_wait_ms,
                   "note": "comm_wait = device time between the end of the backward and the completion of the last bucket on the main stream (median of 5 extra steps)"}
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
            # users the verification pass could not settle from the bf16 draft (re-run on the plain fp32 search) / users, for the headline call
            # (random-init weights) and for the same call on a trained model -- the headline's regime is the friendly one only if both are small
            "beam10_fallback_users_over_users": ([gen["fallback_users"], gen["users"]] if gen else None),
            "beam10_items_per_sec_trained_model": (gen.get("trained_model", {}).get("items_per_s") if gen else None),
            "beam10_fallback_users_over_users_trained_model": ([gen["trained_model"].get("fallback_users"), gen["trained_model"].get("users")]
                                                                if gen and "error" not in gen.get("trained_model", {"error": 1}) else None),
            "generation": gen,
            "generation_plain_bf16": gen_draft,
            "distributed": ddp,
            # `roofline` = the kernel with the largest share of the step's time IN THIS RUN (library profiler, HIP events around every
            # launch of 3 extra steps outside the timed region): aggregate algorithmic FLOPs of its launches / their summed in-step
            # durations.  `alone` = its largest shape timed by itself (operands warm) with the PMC HBM traffic of that launch.
            # `roofline_wgrad`: the grouped weight-gradient launch (two encoder layers, round 3's `roofline`), `step_kernels`: the
            # whole per-kernel table of the step.
        }
        live = {}
        if world == 1 and not args.no_pmc:
            try:
                live = pmc_live((Mg, Ng, Kg), want_decode=gen_draft is not None)
            except Exception as ex:      # (never lose the bench line to a profiler hiccup)
                live = {"error": repr(ex)[:200]}
        cls = kernel_classes(prof_rows) if prof_rows else []
        gemm_cls = [c for c in cls if c["flops_per_step"] > 0]
        fwd_alone = {"shape": [Mg, Ng, Kg], "avg_launch_us": t_k * 1e6, "achieved": ach, "frac": ach / BF16_PEAK_TFLOPS,
                     "traffic": pmc_traffic("fwd_wide", (Mg, Ng, Kg)), "algorithmic_bytes": 2.0 * (Mg * Kg + Ng * Kg + Mg * Ng), "traffic_stale": True,
                     "traffic_source": "STALE: profiles/pmc_traffic.json, collected by an earlier run (profiles/collect_pmc.sh: separate FETCH_SIZE / WRITE_SIZE "
                                       "passes, FETCH_SIZE x2 on gfx950) -- rocprofv3 was not available to this run or --no-pmc was given"}
        if live.get("gemm"):
            fwd_alone.update({"traffic": live["gemm"]["traffic_bytes"], "traffic_read": live["gemm"]["read_bytes"], "traffic_write": live["gemm"]["write_bytes"],
                              "traffic_stale": False, "traffic_source": live["gemm"]["source"]})
        if live.get("sq"):
            fwd_alone.update({"mfma_busy": live["sq"]["mfma_busy"], "wait_frac_of_wave_cycles": live["sq"]["wait_frac_of_wave_cycles"],
                              "issue_frac_of_wave_cycles": live["sq"]["issue_frac_of_wave_cycles"], "mfma_busy_source": live["sq"]["source"]})
        wg_alone = {"shape": [[c.d_model, c.d_ff, Mg], [c.d_ff, c.d_model, Mg], [c.d_model, inner, Mg], [3 * inner, c.d_model, Mg]] * 2,
                    "avg_launch_us": t_w * 1e6, "achieved": ach_w, "frac": ach_w / BF16_PEAK_TFLOPS, "flops_per_launch": fl_w,
                    "traffic": pmc_traffic("wgrad_group2", (Mg, c.d_model, c.d_ff)), "algorithmic_bytes": by_w, "traffic_stale": True,
                    "traffic_source": "STALE: profiles/pmc_traffic.json (an earlier run of profiles/collect_pmc.sh)"}
        if cls:
            top = cls[0]
            tf = top["flops_per_step"] / max(top["us_per_step"], 1e-9) / 1e6
            is_ks = "[KS" in top["kernel"] or " KS]" in top["kernel"]      # (the tag, not the template parameter's name)
            line["roofline"] = {"bound": "mfma", "kernel": top["kernel"], "achieved": tf, "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / BF16_PEAK_TFLOPS,
                                "launches_per_step": top["launches_per_step"], "us_per_step": top["us_per_step"], "flops_per_step": top["flops_per_step"],
                                "share_of_kernel_time": top["us_per_step"] / max(1e-9, sum(c["us_per_step"] for c in cls)),
                                "by_grid": top["grids"], "alone": wg_alone if is_ks else fwd_alone, "traffic": (wg_alone if is_ks else fwd_alone)["traffic"],
                                "traffic_source": (wg_alone if is_ks else fwd_alone)["traffic_source"], "traffic_stale": (wg_alone if is_ks else fwd_alone).get("traffic_stale", True),
                                "traffic_of": "one launch of the kernel's largest shape in the step (`alone`), per launch like `alone.achieved`",
                                "mfma_busy": (None if is_ks else fwd_alone.get("mfma_busy")), "mfma_busy_source": (None if is_ks else fwd_alone.get("mfma_busy_source")),
                                "source": "p5_profile_begin/end in this run: HIP events around every launch of 3 steps (durations include the dispatch gap)"}
            # which number to read: `frac` is computed from event-BRACKETED durations (each includes its launch's dispatch gap: an upper bound of
            # the kernel time, so `frac` is a lower bound).  The smallest bracket of the step (a trivial kernel) bounds that gap from above;
            # `frac_excl_dispatch` subtracts it per launch and is what a rocprofv3 kernel trace of the same step shows within a few per cent
            # (profiles/r05_train_in_step.json holds the rocprofv3 durations of this build).
            floor_us = min(r["total_us"] / max(1, r["launches"]) for r in prof_rows)
            us_excl = max(1e-9, top["us_per_step"] - floor_us * top["launches_per_step"])
            line["roofline"]["bracket_floor_us"] = floor_us
            line["roofline"]["frac_excl_dispatch"] = top["flops_per_step"] / us_excl / 1e6 / BF16_PEAK_TFLOPS
            ks = [c for c in cls if "p5_gemm5_kernel" in c["kernel"] and "[KS" in c["kernel"]]
            if ks and not is_ks:
                k0 = ks[0]
                tfk = k0["flops_per_step"] / max(k0["us_per_step"], 1e-9) / 1e6
                line["roofline_wgrad"] = {"bound": "mfma", "kernel": k0["kernel"], "achieved": tfk, "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfk / BF16_PEAK_TFLOPS,
                                          "launches_per_step": k0["launches_per_step"], "us_per_step": k0["us_per_step"], "alone": wg_alone, "traffic": wg_alone["traffic"]}
            line["gemm_family"] = {"tflops": sum(c["flops_per_step"] for c in gemm_cls) / max(1e-9, sum(c["us_per_step"] for c in gemm_cls)) / 1e6,
                                   "us_per_step": sum(c["us_per_step"] for c in gemm_cls), "launches_per_step": sum(c["launches_per_step"] for c in gemm_cls)}
            line["step_kernels"] = [{"kernel": c["kernel"], "launches_per_step": c["launches_per_step"], "us_per_step": round(c["us_per_step"], 2),
                                     "tflops": (round(c["flops_per_step"] / max(c["us_per_step"], 1e-9) / 1e6, 1) if c["flops_per_step"] > 0 else None),
                                     "by_grid": ([{"grid": g["grid"], "launches_per_step": g["launches_per_step"], "avg_us": round(g["avg_us"], 2), "tflops": round(g["tflops"], 1)}
                                                  for g in c["grids"]] if c["flops_per_step"] > 0 and len(c["grids"]) > 1 else None)} for c in cls]
            line["step_launches"] = sum(c["launches_per_step"] for c in cls)
        else:
            line["roofline"] = dict(bound="mfma", kernel=k_wg, peak=BF16_PEAK_TFLOPS, unit="TFLOP/s", **wg_alone)
        line["roofline_fwd"] = dict(bound="mfma", kernel=k_fwd, peak=BF16_PEAK_TFLOPS, unit="TFLOP/s", **fwd_alone)
        if gen_draft:
            # the decode step of the plain bf16 search (K = 10): the steps the forced-prefix pass did not cover
            timing = gen_draft["timing_ms"] or {}
            S = gen_draft["decoded_len"] - 1 - int(timing.get("forced_prefix_steps", 0))
            timed = timing.get("decode_ms")
            step_ms = timed / max(1, S) if timed else gen_draft["ms_per_batch"] / max(1, S)
            byts = gen_bytes_per_step(c.d_model, inner, c.d_ff, c.num_decoder_layers, V, 20, 10, L, (gen_draft["decoded_len"] - 1) // 2 + 1)
            gbs = byts / (step_ms * 1e-3) / 1e9
            line["roofline_generation"] = {"bound": "hbm", "kernel": "decode step (all launches of one step), plain bf16 search", "achieved": gbs, "peak": HBM_PEAK_GBS,
                                           "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes": byts, "ms_per_step": step_ms,
                                           "steps": S, "forced_prefix_steps": int(timing.get("forced_prefix_steps", 0)),
                                           "traffic": live["decode"]["traffic_bytes_per_step"] if live.get("decode") else decode_step_traffic(),
                                           "traffic_stale": not live.get("decode"),
                                           "traffic_source": live["decode"]["source"] if live.get("decode") else
                                           "STALE: profiles/pmc_decode_step.json from an earlier run (FETCH_SIZE x2 + WRITE_SIZE over every kernel of the "
                                           "decode steps of tools/gen_bench.py, separate --pmc passes, per step; fabric-side counters: L2-miss traffic)",
                                           "time_source": "device time of the decode loop (engine HIP events) / decode steps" if timed else "whole generate() / steps",
                                           "note": "bytes = decoder weights + tied head once + shared cross-KV + self-KV (SURVEY 8(d)); the steps every item id shares "
                                                   "run as ONE teacher-forced pass before the loop (p5_decode.h) and are not decode steps"}
        legs_on = [] if args.legs == "none" else (["configs", "task_mix"] if args.legs == "all" else args.legs.split(","))
        if world == 1:
            del model, opt
            torch.cuda.empty_cache()
            legs = {}
            if staged:
                legs["staged_backward_1gpu"] = staged
            if "task_mix" in legs_on:
                try:
                    legs["task_mix"] = task_mix_leg(be, device)
                except Exception as ex:
                    legs["task_mix"] = {"error": repr(ex)[:300]}
            if "configs" in legs_on:
                legs.update(config_legs(be, device, args.dtype))
            if legs:
                line["legs"] = legs
            if not args.no_cpu:
                line["cpu_baseline"] = cpu_baseline_train_hf() or cpu_baseline_train()      # stock HF when transformers is importable, else the oracle port
                line["cpu_baseline_generation"] = cpu_baseline_generation()
        print(json.dumps(line), flush=True)
    if world > 1:
        barrier(world)      # rank 0 times its stand-alone kernels after the last collective: the others wait here, then all tear down together
        _dist().destroy_process_group()


if __name__ == "__main__":
    main()
