"""Launcher with the reference's command line (/root/reference/src/src_t5/main.py:24-232): same four flag groups, same
dataset / sampler / collator / runner wiring.  One process per GPU: either started by torchrun (RANK / LOCAL_RANK /
WORLD_SIZE in the environment) or spawned here over the ids in --gpu, rendezvous on 127.0.0.1:--master_port, RCCL backend.

    python -m openp5_amd.main --datasets ML1M --tasks sequential,straightforward --item_indexing sequential \\
        --prompt_file ../prompt.txt --batch_size 64 --sample_prompt 1 --sample_num 3,3 --max_his 20 --epochs 10
"""
import argparse
import logging
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch.utils.data import ConcatDataset, DataLoader
from torch.utils.data.distributed import DistributedSampler

from .collator import Collator
from .data import MultiTaskDataset
from .model import P5ModelConfig, P5T5Native
from .runner import DistributedRunner
from .sampler import DistMultiDataTaskSampler, SingleMultiDataTaskSampler, parse_sampler_args
from .tokenizer import load_tokenizer
from .utils import utils
from .utils.initialization import random_initialization


def build_parser():
    from .runner import build_arg_parser
    return build_arg_parser()


def get_dataset(args):
    train_sets, valid_sets = [], []
    for name in args.datasets.split(","):
        train_sets.append(MultiTaskDataset(args, name, "train"))
        if args.valid_select > 0:
            valid_sets.append(MultiTaskDataset(args, name, "validation"))
    return ConcatDataset(train_sets), (ConcatDataset(valid_sets) if args.valid_select > 0 else None)


def get_loader(args, tokenizer, TrainSet, ValidSet, rank=0, world_size=1):
    if args.dist_sampler == 0:
        sampler = (DistMultiDataTaskSampler(TrainSet, args.batch_size, world_size, rank, args.seed, shuffle=True) if world_size > 1
                   else SingleMultiDataTaskSampler(TrainSet, args.batch_size, args.seed, shuffle=True))
    else:
        sampler = DistributedSampler(TrainSet, num_replicas=world_size, rank=rank) if world_size > 1 else None
    collator = Collator(tokenizer)
    train_loader = DataLoader(dataset=TrainSet, sampler=sampler, batch_size=args.batch_size, collate_fn=collator, shuffle=False)
    valid_loader = None
    if ValidSet is not None:
        vs = DistributedSampler(ValidSet, num_replicas=world_size, rank=rank) if world_size > 1 else None
        valid_loader = DataLoader(dataset=ValidSet, sampler=vs, batch_size=args.batch_size, collate_fn=collator, shuffle=False)
    return train_loader, valid_loader


def worker(local_rank, args, world_size, spawned):
    rank = int(os.environ.get("RANK", local_rank)) if not spawned else local_rank
    args.rank, args.world_size = rank, world_size
    args.distributed = 1 if world_size > 1 else 0
    utils.set_seed(args.seed)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", args.master_addr)
        os.environ.setdefault("MASTER_PORT", str(args.master_port))
        dist.init_process_group(backend="nccl", world_size=world_size, rank=rank, device_id=device)
    if rank == 0:
        utils.setup_logging(args)
        logging.info(vars(args))
    else:
        args.log_name = utils.log_name(args)
    utils.setup_model_path(args)
    tokenizer = load_tokenizer(args.backbone)
    TrainSet, ValidSet = get_dataset(args)
    train_loader, valid_loader = get_loader(args, tokenizer, TrainSet, ValidSet, rank, world_size)
    cfg = P5ModelConfig.from_backbone(args.backbone, dropout_rate=args.dropout if hasattr(args, "dropout") else 0.1)
    model = P5T5Native.from_pretrained(args.backbone, config=cfg, dtype=args.compute_dtype, device=device, seed=args.seed)
    if args.item_indexing == "collaborative":
        for ds in TrainSet.datasets:
            # first-occurrence order, duplicates dropped by add_tokens itself: the <CIk> token ids -- and so the rows of the
            # embedding table -- are the reference's (main.py:190-192), which keeps collaborative checkpoints interchangeable
            tokenizer.add_tokens(list(ds.new_token))
    model.resize_token_embeddings(len(tokenizer))
    if args.random_initialize == 1:
        random_initialization(model, tokenizer, args.backbone)
    if args.load:
        utils.load_model(model, args.model_path, args)
    runner = DistributedRunner(model, tokenizer, train_loader, valid_loader, device, args, rank)
    if args.train:
        runner.train()
    if world_size > 1:
        dist.barrier()
    # the reference always evaluates the SERIALISED checkpoint after training (main.py:143-147,221-225): with --valid_select 1
    # that is the best-validation model, and the state-dict round trip is exercised either way
    if rank == 0:
        logging.info(f"Load model from {args.model_path}")
    runner.test(args.model_path)
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    args, _ = build_parser().parse_known_args(argv)
    if "WORLD_SIZE" in os.environ:                       # started by torchrun: one process per GPU already
        worker(int(os.environ.get("LOCAL_RANK", 0)), args, int(os.environ["WORLD_SIZE"]), spawned=False)
        return
    gpus = [g for g in str(args.gpu).split(",") if g != ""]
    # the reference exports CUDA_VISIBLE_DEVICES = --gpu before any device is touched (main.py:226-228); the ROCm runtime reads
    # the same variable (and HIP_VISIBLE_DEVICES), so local rank r runs on the r-th id listed in --gpu
    if gpus and not torch.cuda.is_initialized():
        os.environ["CUDA_VISIBLE_DEVICES"] = ",".join(gpus if args.distributed else gpus[:1])
    n = min(len(gpus), torch.cuda.device_count()) if args.distributed else 1
    if n <= 1:
        worker(0, args, 1, spawned=True)
    else:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = args.master_addr, str(args.master_port)
        mp.spawn(worker, args=(args, n, True), nprocs=n, join=True)


if __name__ == "__main__":
    main()
