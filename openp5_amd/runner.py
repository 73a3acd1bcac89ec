"""Trainer / evaluator for the T5 path -- the same contract as the reference's `DistributedRunner`
(/root/reference/src/src_t5/runner/DistributedRunner.py:21-400 on top of SingleRunner.py:13-233), with its flags.

Differences, all deliberate:
  * world_size == 1 works (the reference's SingleRunner is broken as shipped);
  * gradients ARE averaged across ranks: the reference wraps the model in DDP but calls `.module(...)`, so its reducer
    never fires; here the backward all-reduces contiguous buckets of the flat gradient arena while it is still running;
  * clip + AdamW + schedule is the fused arena optimizer (same arithmetic: HF AdamW, wd on every parameter, linear warmup);
  * no per-step barriers / loss all-reduce (loss is all-reduced once per logging interval);
  * the trie constraint runs on the device (the model recognises our and the reference's prefix_allowed_tokens_fn).
"""
import logging
import math
import time

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from . import evaluate
from .collator import Collator, TestCollator
from .data import TestDataset
from .optim import FusedAdamW
from .trie import CompiledTrie, Trie, prefix_allowed_tokens_fn


def parse_runner_args(parser):
    parser.add_argument("--optim", type=str, default="AdamW", help="The name of the optimizer")
    parser.add_argument("--epochs", type=int, default=10)
    parser.add_argument("--lr", type=float, default=1e-3)
    parser.add_argument("--clip", type=float, default=1)
    parser.add_argument("--logging_step", type=int, default=100)
    parser.add_argument("--warmup_prop", type=float, default=0.05)
    parser.add_argument("--gradient_accumulation_steps", type=int, default=1)
    parser.add_argument("--weight_decay", type=float, default=0.01)
    parser.add_argument("--adam_eps", type=float, default=1e-6)
    parser.add_argument("--dropout", type=float, default=0.1)
    parser.add_argument("--alpha", type=float, default=2)
    parser.add_argument("--train", type=int, default=1, help="train or not")
    parser.add_argument("--backbone", type=str, default="t5-small", help="backbone model name")
    parser.add_argument("--metrics", type=str, default="hit@5,hit@10,ndcg@5,ndcg@10", help="Metrics used for evaluation")
    parser.add_argument("--load", type=int, default=0, help="load model from model path or not.")
    parser.add_argument("--random_initialize", type=int, default=1, help="Randomly initialize number-related tokens.")
    parser.add_argument("--test_epoch", type=int, default=1, help="test once for how many epochs, 0 for no test during training.")
    parser.add_argument("--valid_select", type=int, default=0, help="use validation loss to select models")
    parser.add_argument("--test_before_train", type=int, default=1, help="whether test before training")
    parser.add_argument("--test_filtered", type=int, default=0, help="whether filter out the items in the training data.")
    parser.add_argument("--test_filtered_batch", type=int, default=1, help="whether testing with filtered data in batch (1 = the reference's widened beam, "
                        "up to 64 beams; 2 = history excluded inside the search, any history length; 0 = per-user protocol).")
    parser.add_argument("--gen_lanes", type=int, default=3, help="evaluation batches in flight (P5T5Native.map_lanes): each lane has its own search / "
                        "verification engines, workspaces and HIP stream over the one set of weights, so one batch's latency-bound beam search overlaps "
                        "the next one's; 1 = one batch at a time")
    parser.add_argument("--id_metrics", type=int, default=1, help="compare generated token ids with the gold ids on the device "
                        "instead of decoding to strings (same Hit/NDCG; 0 = the reference's string path).")
    parser.add_argument("--compute_dtype", type=str, default="bf16", help="bf16 (fast) or fp32 (parity) engine arithmetic")
    parser.add_argument("--ddp_bucket_dtype", type=str, default="fp32", help="fp32 | bf16: dtype of the gradient buckets the ranks exchange "
                        "(bf16 halves the bytes on the xGMI links; the sums are identical on every rank either way).")
    parser.add_argument("--resume", type=int, default=0, help="continue from <model_path>.resume when it exists (weights + optimizer "
                        "moments + schedule position + epoch/step + dropout and data-order state; the reference saves weights only).")
    parser.add_argument("--save_steps", type=int, default=0, help="with --resume: also write the resume file every N optimizer steps "
                        "(0 = at epoch ends only).")
    return parser


def build_arg_parser(description="OpenP5 (MI355X-native T5 path)"):
    """The reference's four flag groups merged (main.py:24-232: global, dataset, sampler, runner)."""
    import argparse
    from .data import MultiTaskDataset
    from .sampler import parse_sampler_args
    from .utils import utils
    parser = argparse.ArgumentParser(description=description)
    utils.parse_global_args(parser)
    MultiTaskDataset.parse_dataset_args(parser)
    parse_sampler_args(parser)
    parse_runner_args(parser)
    return parser


def masked_mean_loss(nll, output_attention):
    """DistributedRunner.py:72-77."""
    B, T = output_attention.shape
    m = (output_attention != 0).float()
    loss = nll.view(B, T) * m
    return (loss.sum(dim=1) / m.sum(dim=1).clamp(min=1)).mean()


def training_step(model, optimizer, batch, alpha=2, micro=0, accum=1):
    """One pass of the reference loop body (DistributedRunner.py:63-87): forward, masked-mean loss, backward, clip + AdamW +
    scheduler (fused), zero_grad.  `batch` = (input_ids, whole_word_ids, attention_mask, labels, output_attention) on the
    device.  With the native model the loss and its gradient seed are computed by the engine
    (`P5T5Native.loss_and_backward`: masked mean folded behind the CE kernel, no torch autograd graph); any other model
    goes through the generic torch path.

    `accum` > 1 = --gradient_accumulation_steps: micro-batch `micro` (0-based, within a group of `accum`) adds its gradients
    to the arena (every gradient kernel accumulates; the arena clear at the start of a backward is skipped), ranks exchange
    gradients on the LAST micro-batch only, and the optimizer steps once per group on the group mean (1/accum folded into the
    AdamW gradient scale).  The reference only divides total_steps by this flag and still steps every batch
    (SingleRunner.py:182), which drives its schedule to lr = 0 after 1/accum of the run; this is the intended behaviour."""
    input_ids, whole_ids, attn, output_ids, output_attention = batch[:5]
    fused = getattr(model, "loss_and_backward", None)
    last = micro == accum - 1
    if fused is not None and hasattr(model, "begin_micro_batch"):
        model.begin_micro_batch(first=micro == 0, sync=last)
    if fused is not None:
        loss = fused(input_ids, whole_ids, attn, output_ids, output_attention)
    else:
        out = model(input_ids=input_ids, whole_word_ids=whole_ids, attention_mask=attn, labels=output_ids, alpha=alpha, return_dict=True)
        loss = masked_mean_loss(out["loss"], output_attention)
        (loss / accum if accum > 1 else loss).backward()
    if last:
        if accum > 1 and fused is not None:
            optimizer.step(grad_accum=accum)
        else:
            optimizer.step()          # clip_grad_norm_ + AdamW + scheduler.step, fused
        model.zero_grad()
    return loss.detach()


class Prefetcher:
    """Background collation: the DataLoader (tokenisation + whole-word ids, num_workers=0 as in main.py:61) runs in a thread
    and stays `depth` batches ahead, so host-side batch preparation overlaps the asynchronously issued GPU step instead of
    serialising with it (the reference collates in the training loop)."""

    def __init__(self, loader, depth=3, pin=False):
        import queue
        import threading
        self.q = queue.Queue(maxsize=depth)
        self.pin = pin
        self._done = object()
        self.err = None

        def work():
            try:
                for batch in loader:
                    if self.pin:
                        batch = [t.pin_memory() if torch.is_tensor(t) else t for t in batch]
                    self.q.put(batch)
            except BaseException as e:   # surfaced in the consumer
                self.err = e
            self.q.put(self._done)

        self.t = threading.Thread(target=work, daemon=True)
        self.t.start()

    def __iter__(self):
        while True:
            item = self.q.get()
            if item is self._done:
                if self.err is not None:
                    raise self.err
                return
            yield item


MAX_DEVICE_BEAMS = 64      # include/p5hip.h: p5_generate keeps <= 64 beams per batch item


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class DistributedRunner:
    parse_runner_args = staticmethod(parse_runner_args)

    def __init__(self, model, tokenizer, train_loader, valid_loader, device, args, rank=0):
        self.model, self.tokenizer = model, tokenizer
        self.train_loader, self.valid_loader = train_loader, valid_loader
        self.device, self.args, self.rank = device, args, rank
        self.world = _world()
        self.model.ddp_world = self.world          # gradient all-reduce inside the staged backward
        self.model.ddp_bucket_dtype = getattr(args, "ddp_bucket_dtype", "fp32")
        ds0 = self.train_loader.dataset.datasets[0] if train_loader is not None else None
        self.regenerate_candidate = ds0 is not None and "candidate_items" in ds0.info
        self.reconstruct_data = args.sample_prompt
        self.test_epoch, self.valid_select = args.test_epoch, args.valid_select
        self.test_before_train = args.test_before_train
        self.test_filtered, self.test_filtered_batch = args.test_filtered, args.test_filtered_batch
        self.id_metrics = int(getattr(args, "id_metrics", 1))
        self.gen_lanes = int(getattr(args, "gen_lanes", 3))
        self.metrics = args.metrics.split(",")
        self.generate_num = max(int(m.split("@")[1]) for m in self.metrics)
        self.get_testloader()
        if args.train:
            self.optimizer, self.scheduler = self.create_optimizer_and_scheduler()
        self.samples_per_sec = None
        self.items_per_sec = None

    # ------------------------------------------------------------------ setup
    def create_optimizer_and_scheduler(self):
        batch_per_epoch = len(self.train_loader)
        # one optimizer step per group of `accum` batches, the trailing (shorter) group of an epoch included
        total_steps = math.ceil(batch_per_epoch / max(1, self.args.gradient_accumulation_steps)) * self.args.epochs
        warmup_steps = int(total_steps * self.args.warmup_prop)
        if self.rank == 0:
            logging.info(f"Batch per epoch: {batch_per_epoch}; total steps: {total_steps}; warm up steps: {warmup_steps}")
        if self.args.optim.lower() != "adamw":
            raise NotImplementedError(self.args.optim)
        opt = FusedAdamW(self.model, lr=self.args.lr, eps=self.args.adam_eps, weight_decay=self.args.weight_decay,
                         max_grad_norm=self.args.clip, warmup_steps=warmup_steps, total_steps=total_steps)
        return opt, opt      # the fused optimizer advances its own linear-warmup schedule

    def get_testloader(self):
        self.testloaders = []
        collator = (TestCollator(self.tokenizer, solo_rows=(self.test_filtered_batch == 0)) if self.test_filtered > 0
                    else Collator(self.tokenizer))
        for dataset in self.args.datasets.split(","):
            for task in self.args.tasks.split(","):
                testdata = TestDataset(self.args, dataset, task)
                sampler = DistributedSampler(testdata, num_replicas=self.world, rank=self.rank) if self.world > 1 else None
                self.testloaders.append(DataLoader(dataset=testdata, sampler=sampler, batch_size=self.args.eval_batch_size,
                                                   collate_fn=collator, shuffle=False))

    def _to_dev(self, batch):
        return [t.to(self.device, non_blocking=True) if torch.is_tensor(t) else t for t in batch]

    # ------------------------------------------------------------------ resume state
    def _resume_path(self):
        return str(self.args.model_path) + ".resume"

    def _epoch_start_state(self):
        """Everything the data order of an epoch is derived from, captured BEFORE the epoch's prompt re-sampling and in-place
        cumulative shuffle (MultiTaskDataset.py:189-195): restoring it and replaying the epoch's setup reproduces the batch
        stream exactly, so a mid-epoch resume only has to skip the batches already consumed."""
        import random
        # (plain containers + tensors only, so that the resume file loads with torch.load(weights_only=True))
        pv, pk, pg = random.getstate()
        nn, nk, npos, nhas, ncached = np.random.get_state()
        state = {"py_random": [int(pv), [int(x) for x in pk], pg], "np_random": [str(nn), torch.from_numpy(nk.astype(np.int64)), int(npos), int(nhas), float(ncached)],
                 "torch_rng": torch.get_rng_state()}
        if self.train_loader is not None:
            state["task_data"] = [{t: list(v) for t, v in ds.task_data.items()} for ds in self.train_loader.dataset.datasets]
        return state

    def _restore_epoch_start(self, state):
        import random
        pv, pk, pg = state["py_random"]
        random.setstate((pv, tuple(pk), pg))
        nn, nk, npos, nhas, ncached = state["np_random"]
        np.random.set_state((nn, nk.numpy().astype(np.uint32), npos, nhas, ncached))
        torch.set_rng_state(state["torch_rng"])
        for ds, td in zip(self.train_loader.dataset.datasets, state.get("task_data", [])):
            ds.task_data = {t: list(v) for t, v in td.items()}

    def save_checkpoint(self, path, epoch, step_in_epoch, epoch_start, extra=None):
        """Weights (HF key layout, as utils.save_model) + optimizer moments / step / schedule position + epoch, step inside
        the epoch, dropout counter and the epoch-start data state.  Written by rank 0: parameters and optimizer state are
        identical on every rank (all-reduced gradients, deterministic clip factor), the Python / NumPy / torch RNG states that
        drive the data order are identical by construction (every rank seeds them alike and draws the same sequence; the
        samplers shard AFTER shuffling, DistMultiDataTaskSampler.py:30-33).  The dropout counter is the one piece of state a
        launcher may seed per rank (bench.py, tests/test_gpu_ddp.py do): it is gathered from all ranks and restored per rank."""
        rng = [int(x) for x in getattr(self.model, "_rng_cpu", [0, 0])]
        rng_ranks = [rng]
        if self.world > 1:
            parts = [None] * self.world          # (through the object channel: no device-tensor collective, works on every backend)
            dist.all_gather_object(parts, rng)
            rng_ranks = [[int(v) for v in p] for p in parts]
        if self.rank != 0:
            return
        ck = {"model": {k: v.detach().cpu() for k, v in self.model.state_dict().items()},
              "optimizer": {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in self.optimizer.state_dict().items()},
              "epoch": int(epoch), "step_in_epoch": int(step_in_epoch), "epoch_start": epoch_start,
              "dropout_rng": rng, "dropout_rng_ranks": rng_ranks, "world": self.world}
        ck.update(extra or {})
        tmp = str(path) + ".tmp"
        torch.save(ck, tmp)
        import os
        os.replace(tmp, path)

    def load_checkpoint(self, path):
        ck = torch.load(path, map_location="cpu", weights_only=True)     # tensors and plain containers only: nothing is unpickled
        self.model.load_state_dict(ck["model"], strict=False)
        self.optimizer.load_state_dict(ck["optimizer"])
        if hasattr(self.model, "set_dropout_seed"):
            ranks = ck.get("dropout_rng_ranks") or [ck["dropout_rng"]]
            self.model.set_dropout_seed(*(ranks[self.rank] if self.rank < len(ranks) and ck.get("world") == self.world else ck["dropout_rng"]))
        return ck

    # ------------------------------------------------------------------ training
    def train(self):
        import os
        self.model.zero_grad()
        train_losses, valid_losses, best_epoch = [], [], -1
        accum = max(1, int(self.args.gradient_accumulation_steps))
        resume = int(getattr(self.args, "resume", 0)) > 0
        save_steps = int(getattr(self.args, "save_steps", 0))
        start_epoch, skip_batches, pending_start = 0, 0, None
        carry_sum, carry_cnt = 0.0, 0          # loss of the batches of a resumed epoch that ran before the checkpoint was written
        if resume and os.path.exists(self._resume_path()):
            ck = self.load_checkpoint(self._resume_path())
            start_epoch, skip_batches, pending_start = ck["epoch"], ck["step_in_epoch"], ck["epoch_start"]
            carry_sum, carry_cnt = float(ck.get("loss_sum", 0.0)), int(ck.get("loss_cnt", 0))
            train_losses, valid_losses, best_epoch = ck.get("train_losses", []), ck.get("valid_losses", []), ck.get("best_epoch", -1)
            if self.rank == 0:
                logging.info(f"Resume from {self._resume_path()}: epoch {start_epoch + 1}, batch {skip_batches}, optimizer step {self.optimizer.t}")
        elif self.test_before_train > 0:
            self.test()
        for epoch in range(start_epoch, self.args.epochs):
            if self.rank == 0:
                logging.info(f"Start training for epoch {epoch + 1}")
            if pending_start is not None:
                self._restore_epoch_start(pending_start)
            epoch_start = self._epoch_start_state() if resume else None
            pending_start = None
            if self.regenerate_candidate or self.reconstruct_data:
                for ds in self.train_loader.dataset.datasets:
                    if self.regenerate_candidate and hasattr(ds, "generate_candidates"):
                        ds.generate_candidates()
                    ds.construct_sentence()
            if hasattr(self.train_loader.sampler, "set_epoch"):
                self.train_loader.sampler.set_epoch(epoch)
            self.model.train()
            losses, n_samples, n_batches = [], 0, 0
            n_total = len(self.train_loader)
            t0 = time.perf_counter()
            for batch in Prefetcher(self.train_loader, pin=torch.cuda.is_available()):
                n_batches += 1
                if n_batches <= skip_batches:         # already consumed before the checkpoint was written
                    continue
                input_ids, attn, whole_ids, output_ids, output_attention = self._to_dev(batch)[:5]
                # groups of `accum` batches; the LAST group of the epoch may be shorter -- its last batch is still the one that
                # exchanges gradients across ranks and steps (on the mean over the group's actual size), so every rank applies the
                # same update (a partial group stepped on rank-local sums would let the replicas drift apart for good)
                g0 = (n_batches - 1) // accum * accum
                gsize = min(accum, n_total - g0)
                micro = n_batches - 1 - g0
                loss = training_step(self.model, self.optimizer, (input_ids, whole_ids, attn, output_ids, output_attention), self.args.alpha,
                                     micro=micro, accum=gsize)
                losses.append(loss)
                n_samples += input_ids.shape[0]
                if resume and save_steps > 0 and micro == gsize - 1 and self.optimizer.t % save_steps == 0:
                    self.save_checkpoint(self._resume_path(), epoch, n_batches, epoch_start,
                                         {"train_losses": train_losses, "valid_losses": valid_losses, "best_epoch": best_epoch,
                                          "loss_sum": carry_sum + float(torch.stack(losses).sum()), "loss_cnt": carry_cnt + len(losses)})
            skip_batches = 0
            assert n_batches == n_total, (n_batches, n_total)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            self.samples_per_sec = self.world * n_samples / max(dt, 1e-9)
            epoch_loss = ((torch.stack(losses).sum() + carry_sum) / (len(losses) + carry_cnt) if losses
                          else torch.full((), carry_sum / max(1, carry_cnt), device=self.device))
            carry_sum, carry_cnt = 0.0, 0
            if self.world > 1:
                dist.all_reduce(epoch_loss, op=dist.ReduceOp.SUM)
                epoch_loss /= self.world
            if self.rank == 0:
                train_losses.append(float(epoch_loss))
                logging.info(f"The average training loss for epoch {epoch + 1} is {float(epoch_loss):.6f} ({self.samples_per_sec:.1f} samples/s)")
            if self.valid_select > 0:
                v = self.validate()
                if self.rank == 0:
                    valid_losses.append(v)
                    logging.info(f"The average valid loss for epoch {epoch + 1} is {v}")
                    if v == min(valid_losses):
                        best_epoch = epoch + 1
                        torch.save(self.model.state_dict(), self.args.model_path)
                        logging.info(f"Save the current model to {self.args.model_path}")
            if self.test_epoch > 0 and (epoch + 1) % self.test_epoch == 0:
                self.model.eval()
                self.test()
            if resume:
                # the next epoch starts from the data state as it is NOW (prompts re-sampled, lists permuted cumulatively)
                self.save_checkpoint(self._resume_path(), epoch + 1, 0, self._epoch_start_state(),
                                     {"train_losses": train_losses, "valid_losses": valid_losses, "best_epoch": best_epoch})
            if self.world > 1:
                dist.barrier()
        if self.rank == 0:
            if self.valid_select > 0:
                logging.info(f"The best validation at Epoch {best_epoch}")
            elif getattr(self.args, "model_path", None):
                torch.save(self.model.state_dict(), self.args.model_path)
                logging.info(f"Save the current model to {self.args.model_path}")
        return train_losses

    @torch.no_grad()
    def validate(self):
        self.model.eval()
        if self.args.valid_prompt_sample > 0:
            for ds in self.valid_loader.dataset.datasets:
                ds.construct_sentence()
        losses = []
        for batch in self.valid_loader:
            input_ids, attn, whole_ids, output_ids, output_attention = self._to_dev(batch)[:5]
            out = self.model(input_ids=input_ids, whole_word_ids=whole_ids, attention_mask=attn, labels=output_ids,
                             alpha=self.args.alpha, return_dict=True)
            losses.append(masked_mean_loss(out["loss"], output_attention))
        v = torch.stack(losses).mean()
        if self.world > 1:
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
            v /= self.world
        return float(v)

    # ------------------------------------------------------------------ evaluation
    def test(self, path=None):
        self.model.eval()
        if path:
            self.model.load_state_dict(torch.load(path, map_location="cpu"), strict=False)
        results = []
        for loader in self.testloaders:
            if self.test_filtered > 0:
                if self.test_filtered_batch > 0:
                    results.append(self.test_dataset_task_filtered_batch(loader))
                else:
                    results.append(self.test_dataset_task_filtered(loader))
            else:
                results.append(self.test_dataset_task(loader))
        return results

    def _item_sequences(self, ds, candidates):
        """[decoder start] + token ids of "<dataset> item_<id>" per candidate (DistributedRunner.py:344-350); tokenised once per
        item and cached -- the per-user filtered protocol (:286-297) would otherwise re-tokenise every item for every user."""
        cache = self.__dict__.setdefault("_item_tok_cache", {})
        out = []
        for c in candidates:
            key = (ds.dataset, c)
            seq = cache.get(key)
            if seq is None:
                seq = [0] + self.tokenizer.encode(f"{ds.dataset} item_{c}")
                cache[key] = seq
            out.append(seq)
        return out

    def _generate(self, batch, fn, num_beams, max_length):
        input_ids, attn, whole_ids, output_ids = batch[0], batch[1], batch[2], batch[3]
        pred = self.model.generate(input_ids=input_ids, attention_mask=attn, whole_word_ids=whole_ids, max_length=max_length,
                                   prefix_allowed_tokens_fn=fn, num_beams=num_beams, num_return_sequences=num_beams,
                                   output_scores=True, return_dict_in_generate=True)
        gold = self.tokenizer.batch_decode(output_ids, skip_special_tokens=True)
        gen = self.tokenizer.batch_decode(pred["sequences"], skip_special_tokens=True)
        return gold, gen, pred["sequences_scores"].detach().cpu().tolist()

    def _generate_ids(self, batch, num_beams, max_length, **kw):
        """ID-level evaluation of one batch: bool relevance [B, K] on the device (no decode to strings)."""
        pred = self.model.generate(input_ids=batch[0], attention_mask=batch[1], whole_word_ids=batch[2], max_length=max_length,
                                   num_beams=num_beams, num_return_sequences=num_beams, output_scores=True,
                                   return_dict_in_generate=True, **kw)
        return evaluate.rel_results_ids(pred["sequences"], pred["sequences_scores"], batch[3].to(pred["sequences"].device), num_beams)

    def _lanes_map(self, fn, batches):
        """fn(batch) for every batch in order, several batches in flight when the model offers generation lanes (P5T5Native.map_lanes)."""
        ml = getattr(self.model, "map_lanes", None)
        if ml is None or self.gen_lanes <= 1:
            return (fn(b) for b in batches)
        return ml(fn, batches, lanes=self.gen_lanes)

    def _dataset_trie(self, ds):
        """Item trie of a dataset, compiled once, with the item -> path index used for per-user exclusion."""
        cache = self.__dict__.setdefault("_trie_cache", {})
        ent = cache.get(ds.dataset)
        if ent is None:
            items = list(ds.all_items)
            seqs = self._item_sequences(ds, items)
            trie = Trie(seqs)
            ct = CompiledTrie.from_trie(trie)
            ct.index_items(seqs)
            ent = cache[ds.dataset] = (trie, ct, {it: i for i, it in enumerate(items)})
        return ent

    def _finish(self, metrics_res, test_total, testloader, t0):
        metrics_res = torch.as_tensor(metrics_res, dtype=torch.float64).to(self.device)
        total = torch.tensor(float(test_total), dtype=torch.float64, device=self.device)
        if self.world > 1:
            dist.all_reduce(metrics_res, op=dist.ReduceOp.SUM)
            dist.all_reduce(total, op=dist.ReduceOp.SUM)
        dt = time.perf_counter() - t0
        self.items_per_sec = float(total) * self.generate_num / max(dt, 1e-9)
        metrics_res = (metrics_res / total).cpu().numpy()
        if self.rank == 0:
            for name, val in zip(self.metrics, metrics_res):
                logging.info(f"{name}: {val}")
            logging.info(f"{testloader.dataset.dataset}/{testloader.dataset.task}: {self.items_per_sec:.1f} items/s")
        return dict(zip(self.metrics, metrics_res.tolist()))

    @torch.no_grad()
    def test_dataset_task(self, testloader):
        """DistributedRunner.py:339-399 (max_length 50 there)."""
        ds = testloader.dataset
        trie, ct, _ = self._dataset_trie(ds)
        fn = prefix_allowed_tokens_fn(trie)
        metrics_res, test_total, t0 = 0, 0, time.perf_counter()

        def one(batch):          # runs on a generation lane (its own stream): everything up to the batch's metric sums
            batch = self._to_dev(batch)
            if self.id_metrics:
                rel = self._generate_ids(batch, self.generate_num, 50, trie=ct)
                return evaluate.get_metrics_results_ids(rel, self.metrics), len(rel)
            gold, gen, scores = self._generate(batch, fn, self.generate_num, 50)
            rel = evaluate.rel_results(gen, gold, scores, self.generate_num)
            return evaluate.get_metrics_results(rel, self.metrics), len(rel)

        # (collation overlaps the previous generate(); up to --gen_lanes batches are in flight on the device, results come back in order)
        for m, n in self._lanes_map(one, Prefetcher(testloader, pin=self.device.type == "cuda")):
            if torch.is_tensor(m) and m.is_cuda:
                m.record_stream(torch.cuda.current_stream())      # (allocated on the lane's stream, read here)
            metrics_res = metrics_res + (m.to(self.device) if torch.is_tensor(m) else m)
            test_total += n
        return self._finish(metrics_res, test_total, testloader, t0)

    @torch.no_grad()
    def test_dataset_task_filtered(self, testloader):
        """DistributedRunner.py:271-337: one trie per user = all items minus the user's history."""
        ds = testloader.dataset
        _, ct, index = self._dataset_trie(ds)
        metrics_res, test_total, t0 = 0, 0, time.perf_counter()

        def one(batch):
            batch = self._to_dev(batch)
            # the reference rebuilds Trie(all_items - positive) per user (hence its eval_batch_size == 1); here the shared
            # device trie is used with one excluded-node bitmap per user, so any batch size works
            users = [ds.id2user[int(u)] for u in batch[5].detach().cpu().tolist()]
            excluded = ct.excluded_bitmap([[index[i] for i in ds.positive[u] if i in index] for u in users])
            if self.id_metrics:
                rel = self._generate_ids(batch, self.generate_num, 30, trie=ct, excluded=excluded)
                return evaluate.get_metrics_results_ids(rel, self.metrics), len(rel)
            else:
                input_ids, attn, whole_ids, output_ids = batch[0], batch[1], batch[2], batch[3]
                pred = self.model.generate(input_ids=input_ids, attention_mask=attn, whole_word_ids=whole_ids, max_length=30, trie=ct,
                                           excluded=excluded, num_beams=self.generate_num, num_return_sequences=self.generate_num,
                                           output_scores=True, return_dict_in_generate=True)
                gold = self.tokenizer.batch_decode(output_ids, skip_special_tokens=True)
                gen = self.tokenizer.batch_decode(pred["sequences"], skip_special_tokens=True)
                rel = evaluate.rel_results(gen, gold, pred["sequences_scores"].detach().cpu().tolist(), self.generate_num)
                return evaluate.get_metrics_results(rel, self.metrics), len(rel)

        for m, n in self._lanes_map(one, Prefetcher(testloader, pin=self.device.type == "cuda")):
            if torch.is_tensor(m) and m.is_cuda:
                m.record_stream(torch.cuda.current_stream())      # (allocated on the lane's stream, read here)
            metrics_res = metrics_res + (m.to(self.device) if torch.is_tensor(m) else m)
            test_total += n
        return self._finish(metrics_res, test_total, testloader, t0)

    @torch.no_grad()
    def test_dataset_task_filtered_batch(self, testloader):
        """DistributedRunner.py:204-269: widen the beam by max_positive, filter the history afterwards."""
        ds = testloader.dataset
        trie, ct, index = self._dataset_trie(ds)
        fn = prefix_allowed_tokens_fn(trie)
        width = self.generate_num + ds.max_positive
        if self.test_filtered_batch >= 2:
            # OPT-IN (--test_filtered_batch 2): exclude each user's history INSIDE the constrained search (shared device trie + per-user
            # excluded-node bitmap, batched) instead of widening the beam -- history never appears, the top generate_num are kept.  This is
            # the --test_filtered_batch 0 protocol's result, not the widened-beam protocol's: a beam of generate_num + max_history can
            # keep items an excluded search prunes, so the two may differ.
            return self.test_dataset_task_filtered(testloader)
        if width > MAX_DEVICE_BEAMS:
            # the literal protocol of the reference's DEFAULT flag (SingleRunner.py:39, DistributedRunner.py:235-236) needs
            # num_beams = generate_num + max history; never silently substitute another protocol for it
            raise ValueError(f"--test_filtered_batch 1 needs num_beams = {self.generate_num} + max history {ds.max_positive} = {width} > "
                             f"{MAX_DEVICE_BEAMS} (device beam limit).  Use --test_filtered_batch 2 (history excluded inside the search, batched: "
                             "the results of --test_filtered_batch 0) or --test_filtered_batch 0 (the released test protocol).")
        seq2idx = None
        if self.id_metrics:     # item token tuple (without the decoder start) -> item index: no batch_decode, no string sets
            seq2idx = self.__dict__.setdefault("_seq2idx_cache", {}).get(ds.dataset)
            if seq2idx is None:
                items = list(ds.all_items)
                seq2idx = {tuple(q[1:]): i for i, q in enumerate(self._item_sequences(ds, items))}
                self._seq2idx_cache[ds.dataset] = seq2idx
        metrics_res, test_total, t0 = 0, 0, time.perf_counter()
        for batch in Prefetcher(testloader, pin=self.device.type == "cuda"):      # collation overlaps the previous generate()
            batch = self._to_dev(batch)
            if self.id_metrics:
                pred = self.model.generate(input_ids=batch[0], attention_mask=batch[1], whole_word_ids=batch[2], max_length=30, trie=ct,
                                           num_beams=width, num_return_sequences=width, output_scores=True, return_dict_in_generate=True)
                B = batch[0].shape[0]
                seqs = pred["sequences"].detach().cpu().numpy().reshape(B, width, -1)
                sc = pred["sequences_scores"].detach().cpu().numpy().reshape(B, width)
                users = [ds.id2user[int(u)] for u in batch[5].detach().cpu().tolist()]
                pos = [{index[i] for i in ds.positive[u] if i in index} for u in users]
                rel = evaluate.rel_results_filtered_ids(seqs, sc, batch[3].detach().cpu().numpy(), pos, seq2idx, self.generate_num)
            else:
                gold, gen, scores = self._generate(batch, fn, width, 30)
                rel = evaluate.rel_results_filtered(ds.positive_text, ds.id2user, batch[5].detach().cpu().numpy(), width, gen, gold, scores,
                                                    self.generate_num)
            test_total += len(rel)
            metrics_res = metrics_res + evaluate.get_metrics_results(rel, self.metrics)
        return self._finish(metrics_res, test_total, testloader, t0)


SingleRunner = DistributedRunner
