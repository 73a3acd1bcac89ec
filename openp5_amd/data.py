"""Datasets of the T5 pipeline: prompt-filled {'input','output'} text samples for training / validation / test.

Same construction rules, flags and public attributes as /root/reference/src/src_t5/data/MultiTaskDataset.py (:19-352)
and TestDataset.py (:17-181): leave-one-out split (train = items[:-2], validation target = items[-2], test target =
items[-1]), history clipped to --max_his, `item_` prefix, per-task index ranges (`task_data`, `task_index`) and the
in-place cumulative `shuffle(seed)` the samplers rely on."""
import logging
import os
import random
import re

import torch
from torch.utils.data import Dataset

from .utils import indexing, utils
from .utils.prompt import check_task_prompt, get_info_from_prompt, load_prompt_template


def parse_dataset_args(parser):
    parser.add_argument("--data_path", type=str, default="../data", help="data directory")
    parser.add_argument("--item_indexing", type=str, default="sequential", help="item indexing method, including random, sequential and collaborative")
    parser.add_argument("--tasks", type=str, default="sequential,direct,straightforward", help="Downstream tasks, separate by comma")
    parser.add_argument("--datasets", type=str, default="Beauty", help="Dataset names, separate by comma")
    parser.add_argument("--prompt_file", type=str, default="../prompt_template.txt", help="the path of the prompt template file")
    parser.add_argument("--sequential_order", type=str, default="original", help="The rank of user history during")
    parser.add_argument("--collaborative_token_size", type=int, default=200, help="the number of tokens used for indexing")
    parser.add_argument("--collaborative_cluster", type=int, default=20, help="the number of clusters in each level for collaborative indexing.")
    parser.add_argument("--collaborative_last_token", type=str, default="sequential", help="how to assign the last token to items within the same clusters, random or sequential")
    parser.add_argument("--collaborative_float32", type=int, default=0, help="1 for use float32 during indexing, 0 for float64.")
    parser.add_argument("--max_his", type=int, default=-1, help="the max number of items in history sequence, -1 means no limit")
    parser.add_argument("--his_prefix", type=int, default=1, help="whether add prefix in history")
    parser.add_argument("--his_sep", type=str, default=" , ", help="The separator used for history")
    parser.add_argument("--skip_empty_his", type=int, default=1, help="whether include data with empty history.")
    parser.add_argument("--valid_prompt", type=str, default="seen:0", help="The prompt used for evaluation, seen/unseen: id")
    parser.add_argument("--valid_prompt_sample", type=int, default=1, help="use sampled prompt for validation every epoch.")
    parser.add_argument("--valid_sample_num", type=str, default="3,3", help="the number of sampled data for each task")
    parser.add_argument("--test_prompt", type=str, default="seen:0", help="The prompt used for evaluation, seen/unseen: id")
    parser.add_argument("--sample_prompt", type=int, default=0, help="sample prompt or not")
    parser.add_argument("--sample_num", type=str, default="2,2,2", help="the number of sampled data for each task")
    return parser


def _apply_indexing(args, dataset, user_sequence_dict, rank=0, distributed=False):
    """Rank 0 writes the cached index files first, the others read them (MultiTaskDataset.py:97-121)."""
    def run():
        if args.item_indexing == "sequential":
            return indexing.sequential_indexing(args.data_path, dataset, user_sequence_dict, args.sequential_order)
        if args.item_indexing == "random":
            return indexing.random_indexing(args.data_path, dataset, user_sequence_dict)
        if args.item_indexing == "collaborative":
            return indexing.collaborative_indexing(args.data_path, dataset, user_sequence_dict, args.collaborative_token_size,
                                                   args.collaborative_cluster, args.collaborative_last_token, args.collaborative_float32)
        raise NotImplementedError(args.item_indexing)
    if distributed:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            if rank == 0:
                run()
            dist.barrier()
    return run()


def _sample_fields(prefix, his_sep, max_his, dataset, user, target, history, want_history):
    s = {"dataset": dataset, "user_id": user, "target": ("item_" + target) if prefix > 0 else target}
    if want_history:
        if max_his > 0:
            history = history[-max_his:]
        s["history"] = his_sep.join(("item_" + h) if prefix > 0 else h for h in history)
    return s


class MultiTaskDataset(Dataset):
    parse_dataset_args = staticmethod(parse_dataset_args)

    def __init__(self, args, dataset, mode):
        super().__init__()
        self.args, self.dataset, self.mode = args, dataset, mode
        self.data_path = args.data_path
        self.tasks = args.tasks.split(",")
        if args.sample_prompt > 0:
            assert len(self.tasks) == len(args.sample_num.split(",")), "prompt sample number does not match task number"
        self.item_indexing = args.item_indexing
        self.rank = getattr(args, "rank", 0)
        self.prefix, self.skip_empty_his = args.his_prefix, args.skip_empty_his
        self.prompt = load_prompt_template(args.prompt_file, self.tasks)
        check_task_prompt(self.prompt, self.tasks)
        self.info = get_info_from_prompt(self.prompt)
        self.max_his, self.his_sep = args.max_his, args.his_sep
        self.user_sequence = utils.ReadLineFromFile(os.path.join(self.data_path, dataset, "user_sequence.txt"))
        self.user_sequence_dict = indexing.construct_user_sequence_dict(self.user_sequence)
        self.reindex_user_seq_dict, self.item_map = _apply_indexing(args, dataset, self.user_sequence_dict, self.rank, bool(getattr(args, "distributed", 0)))
        if self.item_indexing == "collaborative":
            self.new_token = []
            for idx in self.item_map.values():
                self.new_token += re.findall(r"\<.*?\>", idx)
        self.all_items = list(self.item_map.values())
        self.positive = self.get_positive()
        if mode == "train":
            self.data_samples = self.load_train()
        elif mode == "validation":
            self.data_samples = self.load_validation()
            self.valid_prompt = args.valid_prompt
        else:
            raise NotImplementedError(mode)
        self.get_prompt_info()
        self.construct_sentence()
        if self.rank == 0:
            logging.info(f"{dataset}/{mode}: {len(self.data_samples)} samples -> {len(self)} prompts")

    def get_positive(self):
        cut = {"train": -2, "validation": -1}.get(self.mode)
        return {u: set(items[:cut] if cut else items) for u, items in self.reindex_user_seq_dict.items()}

    def shuffle(self, seed):
        g = torch.Generator()
        g.manual_seed(seed)
        for task in self.task_data:          # in place and cumulative across epochs, one generator for all tasks
            order = torch.randperm(len(self.task_data[task]), generator=g).tolist()
            self.task_data[task] = [self.task_data[task][i] for i in order]

    def get_prompt_info(self):
        if self.mode == "train":
            nums = ([len(self.prompt[t]["seen"]) for t in self.tasks] if self.args.sample_prompt == 0
                    else [int(x) for x in self.args.sample_num.split(",")][:len(self.tasks)])
        else:
            nums = ([1] * len(self.tasks) if self.args.valid_prompt_sample == 0
                    else [int(x) for x in self.args.valid_sample_num.split(",")][:len(self.tasks)])
        self.task_prompt_num = nums
        self.task_index, self.task_data = [], {}
        start = 0
        for task, n in zip(self.tasks, nums):
            end = start + n * len(self.data_samples)
            self.task_index.append(end)
            self.task_data[task] = list(range(start, end))
            start = end

    def load_train(self):
        out = []
        want_his = "history" in self.info
        for user, seq in self.reindex_user_seq_dict.items():
            items = seq[:-2]
            for i in range(len(items)):
                if i == 0 and self.skip_empty_his > 0:
                    continue
                out.append(_sample_fields(self.prefix, self.his_sep, self.max_his, self.dataset, user, items[i], items[:i], want_his))
        return out

    def load_validation(self):
        want_his = "history" in self.info
        return [_sample_fields(self.prefix, self.his_sep, self.max_his, self.dataset, user, seq[-2], seq[:-2], want_his)
                for user, seq in self.reindex_user_seq_dict.items()]

    def __len__(self):
        return len(self.data["input"])

    def construct_sentence(self):
        self.data = {"input": [], "output": []}
        sampled = (self.mode == "train" and self.args.sample_prompt != 0) or (self.mode == "validation" and self.args.valid_prompt_sample != 0)
        if self.mode == "validation" and not sampled:
            seen, pid = self.valid_prompt.split(":")
        for t, task in enumerate(self.tasks):
            seen_prompts = self.prompt[task]["seen"]
            for dp in self.data_samples:
                if sampled:
                    picks = [str(random.randint(0, len(seen_prompts) - 1)) for _ in range(self.task_prompt_num[t])]
                    tpls = [seen_prompts[p] for p in picks]
                elif self.mode == "train":
                    tpls = list(seen_prompts.values())
                else:
                    tpls = [self.prompt[task][seen][pid]]
                for tpl in tpls:
                    self.data["input"].append(tpl["Input"].format(**dp))
                    self.data["output"].append(tpl["Output"].format(**dp))

    def __getitem__(self, idx):
        return {"input": self.data["input"][idx], "output": self.data["output"][idx]}


class TestDataset(Dataset):
    __test__ = False

    def __init__(self, args, dataset, task):
        super().__init__()
        self.args, self.dataset, self.task = args, dataset, task
        self.data_path, self.item_indexing = args.data_path, args.item_indexing
        self.prompt = load_prompt_template(args.prompt_file, [task])
        check_task_prompt(self.prompt, [task])
        self.info = get_info_from_prompt(self.prompt)
        self.max_his, self.his_sep, self.prefix = args.max_his, args.his_sep, args.his_prefix
        self.user_sequence = utils.ReadLineFromFile(os.path.join(self.data_path, dataset, "user_sequence.txt"))
        self.user_sequence_dict = indexing.construct_user_sequence_dict(self.user_sequence)
        self.reindex_user_seq_dict, self.item_map = _apply_indexing(args, dataset, self.user_sequence_dict)
        if self.item_indexing == "collaborative":
            self.new_token = []
            for idx in self.item_map.values():
                self.new_token += re.findall(r"\<.*?\>", idx)
        self.all_items = list(self.item_map.values())
        self.test_prompt, self.test_filtered = args.test_prompt, args.test_filtered
        if args.test_filtered > 0:
            self.user2id = {u: i for i, u in enumerate(self.reindex_user_seq_dict)}
            self.id2user = {i: u for u, i in self.user2id.items()}
            if args.test_filtered_batch > 0:
                self.positive_text, self.max_positive = self.get_positive_batch()
            self.positive = self.get_positive()
        want_his = "history" in self.info
        self.data_samples = [_sample_fields(self.prefix, self.his_sep, self.max_his, dataset, user, seq[-1], seq[:-1], want_his)
                             for user, seq in self.reindex_user_seq_dict.items()]
        self.construct_sentence()

    def get_positive(self):
        return {u: set(seq[:-1]) for u, seq in self.reindex_user_seq_dict.items()}

    def get_positive_batch(self):
        seen, pid = self.test_prompt.split(":")
        tpl = self.prompt[self.task][seen][pid]["Output"]
        positive, max_positive = {}, 0
        for u, seq in self.reindex_user_seq_dict.items():
            positive[u] = {tpl.format(dataset=self.dataset, target=("item_" + i) if self.prefix > 0 else i) for i in seq[:-1]}
            max_positive = max(max_positive, len(positive[u]))
        return positive, max_positive

    def __len__(self):
        return len(self.data_samples)

    def construct_sentence(self):
        seen, pid = self.test_prompt.split(":")
        tpl = self.prompt[self.task][seen][pid]
        self.data = {"input": [tpl["Input"].format(**dp) for dp in self.data_samples],
                     "output": [tpl["Output"].format(**dp) for dp in self.data_samples]}

    def __getitem__(self, idx):
        item = {"input": self.data["input"][idx], "output": self.data["output"][idx]}
        if self.test_filtered > 0:
            item["user_idx"] = self.user2id[self.data_samples[idx]["user_id"]]
        return item
