import os, sys, random, torch, torch.distributed as dist, torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def worker(rank, world, port, tmp, side, sync):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from openp5_amd._lib import hip_backend
    from openp5_amd.model import P5ModelConfig, P5T5Native
    from openp5_amd.runner import masked_mean_loss
    P5T5Native.use_side_stream = side
    be = hip_backend(torch.device("cuda:0"))
    cfg = P5ModelConfig(vocab_size=600, d_model=128, d_ff=256, num_layers=2, num_decoder_layers=2, num_heads=2, dropout_rate=0.0)
    model = P5T5Native(cfg, dtype="fp32", backend=be, seed=3)
    g = torch.Generator().manual_seed(100 + rank)
    ids = torch.randint(3, 600, (4, 12), generator=g).cuda(); labels = torch.randint(3, 600, (4, 5), generator=g).cuda()
    model.eval()
    # local grads first
    model.ddp_world = 1
    nll = model(input_ids=ids, whole_word_ids=torch.zeros_like(ids), attention_mask=torch.ones_like(ids), labels=labels)["loss"]
    masked_mean_loss(nll, torch.ones_like(labels)).backward(); torch.cuda.synchronize()
    local = model._grads.clone()
    ref = local.clone(); dist.all_reduce(ref); torch.cuda.synchronize()
    model.ddp_world = world
    nll = model(input_ids=ids, whole_word_ids=torch.zeros_like(ids), attention_mask=torch.ones_like(ids), labels=labels)["loss"]
    masked_mean_loss(nll, torch.ones_like(labels)).backward()
    if sync: torch.cuda.synchronize()
    got = model._grads.clone(); torch.cuda.synchronize()
    d = (got - ref).abs()
    bad = (d > 1e-5 * (1 + ref.abs())).nonzero().flatten()
    print(f"[side={side} sync={sync}] rank {rank}: max|got-ref| {d.max().item():.3e}  n_bad {bad.numel()} of {got.numel()}  first bad {bad[:5].tolist()}  |local-got| {(got-local).abs().max().item():.3e}", flush=True)
    if bad.numel():
        # which parameters?
        for name, (off, n, shape) in model._views.items():
            nb = ((bad >= off) & (bad < off + n)).sum().item()
            if nb: print(f"   rank {rank} bad in {name}: {nb}/{n}", flush=True)
    dist.barrier(); dist.destroy_process_group()
if __name__ == "__main__":
    for side in (True, False):
        for sync in (False, True):
            mp.spawn(worker, args=(2, 29500 + random.randint(0, 2000), "/tmp", side, sync), nprocs=2, join=True)
