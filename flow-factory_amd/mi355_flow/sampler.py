"""Data-parallel partitioner of a GRPO epoch (host side, no communication).

Mirror of the reference's `GroupContiguousSampler`
(reference src/flow_factory/data_utils/sampler.py:96-163): every rank derives the SAME
permutations from `seed + epoch`, rank r owns groups [r*M/W, (r+1)*M/W), each group's K repeats
stay contiguous on that rank, and the rank's M*K/W samples are cut into micro-batches.  This is
how the rollout shards across the 8 GPUs of a node with zero inter-rank traffic.
"""
from __future__ import annotations

from typing import Iterator, List

import torch


class GroupContiguousSampler:
    def __init__(self, dataset_size: int, batch_size: int, group_size: int, unique_sample_num: int, num_replicas: int,
                 rank: int, seed: int = 0):
        if unique_sample_num > dataset_size:
            raise ValueError(f"`unique_sample_num` ({unique_sample_num}) must be <= dataset size ({dataset_size}).")
        if unique_sample_num % num_replicas != 0:
            raise ValueError(f"unique_sample_num ({unique_sample_num}) must be divisible by num_replicas ({num_replicas}) "
                             f"for GroupContiguousSampler.")
        self.dataset_size, self.batch_size, self.k, self.m = dataset_size, batch_size, group_size, unique_sample_num
        self.num_replicas, self.rank, self.seed = num_replicas, rank, seed
        self.groups_per_rank = self.m // num_replicas
        per_rank = self.groups_per_rank * self.k
        if per_rank % batch_size != 0:
            raise ValueError(f"groups_per_rank * group_size ({per_rank}) must be divisible by batch_size ({batch_size})")
        self.num_batches_per_epoch = per_rank // batch_size
        self.epoch = 0

    def epoch_batches(self, epoch: int) -> List[List[int]]:
        g = torch.Generator()
        g.manual_seed(self.seed + epoch)
        picked = torch.randperm(self.dataset_size, generator=g)[: self.m].tolist()
        order = torch.randperm(self.m, generator=g).tolist()
        groups = [picked[i] for i in order]
        mine = groups[self.rank * self.groups_per_rank : (self.rank + 1) * self.groups_per_rank]
        flat = [idx for idx in mine for _ in range(self.k)]
        bs = self.batch_size
        return [flat[i * bs : (i + 1) * bs] for i in range(self.num_batches_per_epoch)]

    def __iter__(self) -> Iterator[List[int]]:
        while True:
            for b in self.epoch_batches(self.epoch):
                yield b
            self.epoch += 1

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch
