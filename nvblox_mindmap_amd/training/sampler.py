"""Rank-sharded weighted sampling (counterpart of ``DistributedSamplerWrapper(WeightedRandomSampler)`` from catalyst used at
mindmap/data_loading/dataset.py:566-583; catalyst is not available here).

Every rank draws the SAME weighted index sequence (same seed + epoch), pads it to a multiple of the world size and takes
its strided share, so the union over ranks is exactly the single-process sequence -- the property the reference's
tests/test_distributed_sampling.py:52-85 checks."""
from typing import Iterator, Optional, Sequence

import torch
from torch.utils.data import Sampler

from .distributed import get_rank, get_world_size


class DistributedWeightedSampler(Sampler[int]):
    def __init__(self, weights: Sequence[float], num_samples: int, replacement: bool = True, seed: int = 0,
                 num_replicas: Optional[int] = None, rank: Optional[int] = None):
        self.weights = torch.as_tensor(weights, dtype=torch.double)
        self.num_samples, self.replacement, self.seed = int(num_samples), replacement, int(seed)
        self.num_replicas = num_replicas if num_replicas is not None else get_world_size()
        self.rank = rank if rank is not None else get_rank()
        self.epoch = 0
        self.total = -(-self.num_samples // self.num_replicas) * self.num_replicas

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def global_indices(self) -> torch.Tensor:
        g = torch.Generator()
        g.manual_seed(self.seed + self.epoch)
        idx = torch.multinomial(self.weights, self.num_samples, self.replacement, generator=g)
        if self.total > idx.numel():  # pad by wrapping around, like DistributedSampler
            idx = torch.cat([idx, idx[: self.total - idx.numel()]])
        return idx

    def __iter__(self) -> Iterator[int]:
        return iter(self.global_indices()[self.rank:self.total:self.num_replicas].tolist())

    def __len__(self) -> int:
        return self.total // self.num_replicas
