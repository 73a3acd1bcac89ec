"""-m gpu: the data-parallel step on real device buffers.  The round's GPU box has ONE MI355X, so two ranks share cuda:0 and
talk over gloo (RCCL needs one device per rank); this exercises the staged backward + side-stream bucket all-reduce +
fused optimizer on the HIP path.  The RCCL path itself is the same code with backend="nccl" (bench.py, main.py)."""
import os
import random

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from openp5_amd._lib import hip_backend
    from openp5_amd.model import P5ModelConfig, P5T5Native
    from openp5_amd.optim import FusedAdamW
    from openp5_amd.runner import masked_mean_loss
    be = hip_backend(torch.device("cuda:0"))
    cfg = P5ModelConfig(vocab_size=600, d_model=128, d_ff=256, num_layers=2, num_decoder_layers=2, num_heads=2, dropout_rate=0.0)
    model = P5T5Native(cfg, dtype="fp32", backend=be, seed=3)
    model.ddp_world = world
    opt = FusedAdamW(model, lr=1e-2, max_grad_norm=1.0)
    g = torch.Generator().manual_seed(100 + rank)
    ids = torch.randint(3, 600, (4, 12), generator=g).cuda()
    labels = torch.randint(3, 600, (4, 5), generator=g).cuda()
    model.eval()
    nll = model(input_ids=ids, whole_word_ids=torch.zeros_like(ids), attention_mask=torch.ones_like(ids), labels=labels)["loss"]
    masked_mean_loss(nll, torch.ones_like(labels)).backward()
    opt.step()
    torch.cuda.synchronize()
    torch.save({"flat": model._flat.cpu(), "grads": model._grads.cpu(), "ids": ids.cpu(), "labels": labels.cpu()}, os.path.join(tmp, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_gpu_gloo(hip, tmp_path):
    world, port = 2, 29500 + random.randint(0, 2000)
    try:
        mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    except Exception as e:   # gloo without device-tensor support on this build
        if "gloo" in str(e).lower() or "not supported" in str(e).lower() or "backend" in str(e).lower():
            pytest.skip(f"gloo cannot all-reduce HIP tensors here: {e}")
        raise
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["flat"], r1["flat"]), "ranks diverged"
    from openp5_amd.model import P5ModelConfig, P5T5Native
    from openp5_amd.optim import FusedAdamW
    from openp5_amd.runner import masked_mean_loss
    cfg = P5ModelConfig(vocab_size=600, d_model=128, d_ff=256, num_layers=2, num_decoder_layers=2, num_heads=2, dropout_rate=0.0)
    gs = []
    for r in (r0, r1):
        m = P5T5Native(cfg, dtype="fp32", backend=hip, seed=3)
        m.eval()
        ids, labels = r["ids"].cuda(), r["labels"].cuda()
        nll = m(input_ids=ids, whole_word_ids=torch.zeros_like(ids), attention_mask=torch.ones_like(ids), labels=labels)["loss"]
        masked_mean_loss(nll, torch.ones_like(labels)).backward()
        gs.append(m._grads.clone())
    assert torch.allclose(r0["grads"].cuda(), gs[0] + gs[1], atol=1e-5, rtol=1e-4)
    m = P5T5Native(cfg, dtype="fp32", backend=hip, seed=3)
    opt = FusedAdamW(m, lr=1e-2, max_grad_norm=1.0)
    m._grads.copy_(0.5 * (gs[0] + gs[1]))
    opt.step()
    assert torch.allclose(m._flat, r0["flat"].cuda(), atol=1e-5, rtol=1e-4)


def test_world2_runner_train_and_test_on_device(hip, tmp_path):
    """a16/e2 on the device: `runner.train()` + `runner.test()` with two ranks sharing the MI355X over gloo -- sharded training,
    DistributedSampler evaluation with constrained beam search on the HIP path, metric all-reduce (DistributedRunner.py:186,389-395).
    Same assertions as the CPU-emulator test: bit-identical parameters on both ranks, metrics equal to a single-process evaluation
    of the union of the shards."""
    from tests.test_runner_emu import run_world2_runner_check
    try:
        run_world2_runner_check(hip, tmp_path, "hip")
    except Exception as e:   # gloo without device-tensor support on this build
        if "gloo" in str(e).lower() and "support" in str(e).lower():
            pytest.skip(f"gloo cannot all-reduce HIP tensors here: {e}")
        raise


def _nccl_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)      # "nccl" IS RCCL on ROCm
    from openp5_amd._lib import hip_backend
    from openp5_amd.model import P5ModelConfig, P5T5Native
    from openp5_amd.optim import FusedAdamW
    from openp5_amd.runner import training_step
    be = hip_backend(dev)
    cfg = P5ModelConfig(vocab_size=600, d_model=128, d_ff=256, num_layers=2, num_decoder_layers=2, num_heads=2, dropout_rate=0.1)
    model = P5T5Native(cfg, dtype="bf16", device=dev, backend=be, seed=3)
    model.ddp_world = world
    model.set_dropout_seed(2023 + rank, 0)
    opt = FusedAdamW(model, lr=1e-2, max_grad_norm=1.0)
    model.train()
    g = torch.Generator().manual_seed(100 + rank)
    for _ in range(3):                                          # three real steps: staged backward + bucket all-reduce on the side stream
        ids = torch.randint(3, 600, (8, 24), generator=g).to(dev)
        labels = torch.randint(3, 600, (8, 6), generator=g).to(dev)
        training_step(model, opt, (ids, torch.zeros_like(ids), torch.ones_like(ids), labels, torch.ones_like(labels)))
    torch.cuda.synchronize()
    torch.save({"flat": model._flat.cpu()}, os.path.join(tmp, f"n{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL path needs one GPU per rank (this box has fewer than 2)")
def test_two_ranks_rccl(tmp_path):
    """backend="nccl" (RCCL over xGMI), one process per GPU: after three dropout-on bf16 training steps with rank-specific data the
    ranks hold bit-identical parameters (gradients all-reduced per backward stage, deterministic clip factor)."""
    world, port = 2, 27500 + random.randint(0, 1500)
    mp.spawn(_nccl_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    n0, n1 = torch.load(tmp_path / "n0.pt"), torch.load(tmp_path / "n1.pt")
    assert torch.equal(n0["flat"], n1["flat"]), "ranks diverged over RCCL"


def test_staged_backward_ranges_and_gradients(hip):
    """the data-parallel code path on one GPU: stage-by-stage backward == whole backward bit for bit, its final ranges tile the arena;
    T5-small at the benchmark shape (two-layer weight-gradient groups inside the staged backward) and a ragged toy"""
    from oracle import t5_oracle as O
    from tests import cases
    cases.staged_backward_case(hip, O.T5Cfg.named("t5-small", dropout=0.1), 64, 128, 8, dropout=0.1)
    cases.staged_backward_case(hip, O.T5Cfg.named("tiny"), 3, 10, 5)
