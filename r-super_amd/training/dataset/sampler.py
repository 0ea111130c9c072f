"""ChunkedSampler -- the reference's epoch/chunk/rank sharding (training/dataset/dim3/sampler.py:7-146).

Semantics kept bit-for-bit (pinned by tests/golden/sampler.npz, generated from the imported reference):
  * a *cycle* = ceil(dataset_size / samples_per_epoch) epochs and covers the dataset once (:66);
  * the permutation is drawn once per cycle with torch.randperm(generator seeded seed + cycle) (:92-99), or is the identity
    without shuffling (:100-102);
  * epoch e uses the chunk [w * spe, (w + 1) * spe) of that permutation, w = e % cycle_length (:105-110);
  * a short last chunk is padded with random.choices from the rest of the permutation (the whole permutation if the rest is
    empty) -- Python's global RNG, exactly as the reference (:114-128);
  * rank r of world_size takes chunk[r::world_size] (:132); len() = ceil(samples_per_epoch / world_size) (:140-146).
"""
import math
import random

import torch
from torch.utils.data import Sampler


class ChunkedSampler(Sampler):
    def __init__(self, dataset_size: int, samples_per_epoch: int, shuffle: bool = True, seed: int = 0, rank: int = 0, world_size: int = 1):
        super().__init__()
        self.dataset_size, self.samples_per_epoch = dataset_size, samples_per_epoch
        self.shuffle, self.seed, self.rank, self.world_size = shuffle, seed, rank, world_size
        self.shuffled_indices = list(range(dataset_size))
        self.cycle_length = math.ceil(dataset_size / samples_per_epoch)
        self.epoch = 0
        self.cycle = -1                       # forces the first permutation draw

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def _permutation(self, cycle):
        if not self.shuffle:
            return list(range(self.dataset_size))
        g = torch.Generator()
        g.manual_seed(self.seed + cycle)
        return torch.randperm(self.dataset_size, generator=g).tolist()

    def epoch_chunk(self):
        """The samples_per_epoch indices of the current epoch before rank sharding."""
        cycle = self.epoch // self.cycle_length
        if cycle != self.cycle:
            self.cycle = cycle
            self.shuffled_indices = self._permutation(cycle)
        lo = (self.epoch % self.cycle_length) * self.samples_per_epoch
        hi = lo + self.samples_per_epoch
        chunk = self.shuffled_indices[lo:min(hi, self.dataset_size)]
        missing = self.samples_per_epoch - len(chunk)
        if missing > 0:
            pool = self.shuffled_indices[:lo] + self.shuffled_indices[hi:]
            chunk.extend(random.choices(pool if pool else self.shuffled_indices, k=missing))
        return chunk

    def __iter__(self):
        return iter(self.epoch_chunk()[self.rank::self.world_size])

    def __len__(self):
        return math.ceil(self.samples_per_epoch / self.world_size)
