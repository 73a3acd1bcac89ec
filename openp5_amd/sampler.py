"""Task-homogeneous batch samplers (the data-parallel sharding rule of the training path).

Behaviour follows /root/reference/src/src_t5/processor/SingleMultiDataTaskSampler.py:27-80 and
DistMultiDataTaskSampler.py:10-71: every epoch each dataset's per-task index lists are permuted IN PLACE with
randperm(seed + epoch) (MultiTaskDataset.shuffle), rank r keeps task_data[task][r::world], and the stream is emitted
as consecutive groups of `batch_size` indices, round-robin over (dataset, task), wrapping short tasks, so every batch
is task-homogeneous and all ranks are in the same task at the same step.  One class serves both cases
(num_replicas=1, rank=0 == the single-process sampler)."""
import math

from torch.utils.data.sampler import Sampler


def parse_sampler_args(parser):
    parser.add_argument("--batch_size", type=int, default=32, help="batch size")
    parser.add_argument("--eval_batch_size", type=int, default=32, help="the batch size for evaluation")
    parser.add_argument("--dist_sampler", type=int, default=0, help="use DistributedSampler if 1, otherwise use our own sampler.")
    return parser


class DistMultiDataTaskSampler(Sampler):
    def __init__(self, dataset, batch_size, num_replicas=1, rank=0, seed=0, shuffle=True):
        self.dataset, self.batch_size = dataset, batch_size
        self.num_replicas, self.rank, self.seed, self.shuffle = num_replicas, rank, seed, shuffle
        self.epoch = 0
        self.dataset_task_size = [math.ceil(len(ds.task_data[t]) / num_replicas) for ds in dataset.datasets for t in ds.task_data]
        self.largest_task_size = max(self.dataset_task_size)

    parse_sampler_args = staticmethod(parse_sampler_args)

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.batch_size * math.ceil(self.largest_task_size / self.batch_size) * len(self.dataset_task_size)

    def __iter__(self):
        lists, bases = [], []
        starts = [0] + list(self.dataset.cumulative_sizes[:-1])
        for ds, base in zip(self.dataset.datasets, starts):
            if self.shuffle:
                ds.shuffle(self.seed + self.epoch)
            for task in ds.task_data:
                lists.append(ds.task_data[task][self.rank::self.num_replicas])
                bases.append(base)
        n_tasks = len(lists)
        cursors = [0] * n_tasks
        out = []
        rounds = math.ceil(self.largest_task_size / self.batch_size)
        for _ in range(rounds):
            for i in range(n_tasks):
                data, base = lists[i], bases[i]
                for _ in range(self.batch_size):
                    if cursors[i] >= len(data):
                        cursors[i] = 0          # wrap a task that is shorter than the longest one
                    out.append(data[cursors[i]] + base)
                    cursors[i] += 1
        return iter(out)


class SingleMultiDataTaskSampler(DistMultiDataTaskSampler):
    def __init__(self, dataset, batch_size, seed, shuffle=True):
        super().__init__(dataset, batch_size, 1, 0, seed, shuffle)
