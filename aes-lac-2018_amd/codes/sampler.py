"""Length-bucketed batch samplers (reference ``codes/sampler.py:9-30,100-135``).

Manifests are sorted by duration, so consecutive ids have similar lengths; a bin is ``batch_size``
consecutive ids.  ``DistributedBucketingSampler`` is the data-parallel partition of the hot path:
rank r consumes bins r, r+W, r+2W, ... of the (wrapped) bin list, so the global batch is W x batch_size.
The multi-task ``WeightedBucketingRandomSampler`` is out of scope.
"""
import math

import numpy as np
import torch
from torch.utils.data.sampler import Sampler


class BucketingSampler(Sampler):
    def __init__(self, data_source, batch_size=1):
        self.data_source = data_source
        ids = list(range(len(data_source)))
        self.bins = [ids[i:i + batch_size] for i in range(0, len(ids), batch_size)]

    def __iter__(self):
        for ids in self.bins:
            np.random.shuffle(ids)
            yield ids

    def __len__(self):
        return len(self.bins)

    def shuffle(self, epoch):
        np.random.seed(epoch)
        np.random.shuffle(self.bins)


class DistributedBucketingSampler(Sampler):
    def __init__(self, data_source, batch_size=1, num_replicas=None, rank=None):
        if num_replicas is None:
            num_replicas = torch.distributed.get_world_size()
        if rank is None:
            rank = torch.distributed.get_rank()
        self.data_source = data_source
        ids = list(range(len(data_source)))
        self.batch_size = batch_size
        self.bins = [ids[i:i + batch_size] for i in range(0, len(ids), batch_size)]
        self.num_replicas, self.rank = num_replicas, rank
        self.num_samples = int(math.ceil(len(self.bins) / float(num_replicas)))
        self.total_size = self.num_samples * num_replicas

    def __iter__(self):
        bins = self.bins + self.bins[:self.total_size - len(self.bins)]     # wrap so every rank gets as many
        return iter(bins[self.rank::self.num_replicas])

    def __len__(self):
        return self.num_samples

    def shuffle(self, epoch):
        g = torch.Generator()
        g.manual_seed(epoch)
        order = torch.randperm(len(self.bins), generator=g).tolist()
        self.bins = [self.bins[i] for i in order]
