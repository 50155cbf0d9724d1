"""Synthetic KITTI-shaped dataset + the reference's resumable samplers.

Neither box has KITTI / Cityscapes, and the reference's loaders need PIL/torchvision APIs
that no longer exist, so the trainer is fed by
``SyntheticTripletDataset``: it emits exactly the batch-dict contract
``Trainer.process_batch`` reads (reference: datasets/mono_dataset.py:189-279; SURVEY.md
section 3.4).  The samplers keep the reference's semantics (datasets/__init__.py:10-85):
permutation seeded by ``seed + epoch``, rank-strided partition, mid-epoch resume through
``set_start_iter``.
"""
import numpy as np
import torch
from torch.utils.data import Dataset, Sampler

from . import synthetic


class SyntheticTripletDataset(Dataset):
    """``device_augment`` (default): an item carries the three raw frames, the intrinsics, the
    affine metadata and the per-item augmentation DRAW (flip / jitter flags, ColorJitter factors
    and order: mono-vifi_amd/augment.py); the pixels are flipped, jittered and affine-warped on
    the device by ``Trainer.process_batch``.  False: the host builds every view (round-1 form)."""

    def __init__(self, height, width, length=39810, use_affine=True, seed=1234, device_augment=True):
        self.height, self.width, self.length = height, width, length
        self.use_affine, self.seed = use_affine, seed
        self.device_augment = bool(device_augment)
        self.epoch = 0

    def set_epoch(self, epoch):
        """The reference redraws flip / ColorJitter parameters on every access
        (datasets/mono_dataset.py:214-256); here the draw is a function of (seed, epoch, index), so
        an item is augmented differently in every epoch and identically on resume.  Called by
        `Trainer.run_epoch` before the loader's workers are created."""
        self.epoch = int(epoch)

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        b = synthetic.training_batch(self.seed * 1000003 + index, 1, self.height, self.width)
        item = {}
        if self.device_augment:
            from . import augment
            draw = augment.draw_params(np.random.default_rng([self.seed, self.epoch, index]), 1)
            b = {k: v for k, v in b.items()
                 if not (isinstance(k, tuple) and k[0] in ("color_aug", "color_affine", "color_affine_aug"))}
            b.update(draw)
        for k, v in b.items():
            t = torch.from_numpy(np.ascontiguousarray(v[0]))
            item[k] = t
        if not self.use_affine:
            for k in list(item):
                if isinstance(k, tuple) and k[0].startswith("color_affine"):
                    del item[k]
        return item


class CustomSampler(Sampler):
    """Single-process resumable sampler (reference: datasets/__init__.py:10-31)."""

    def __init__(self, dataset, seed=0):
        self.len = len(dataset)
        self.start_iter = 0
        self.epoch = 0
        self.seed = seed

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.seed + self.epoch)
        return iter(torch.randperm(self.len, generator=g).tolist()[self.start_iter:])

    def __len__(self):
        return self.len

    def set_start_iter(self, start_iter):
        self.start_iter = start_iter

    def set_epoch(self, epoch):
        self.epoch = epoch


class CustomDistributedSampler(Sampler):
    """Rank-strided resumable sampler (reference: datasets/__init__.py:34-85): every rank
    draws the same seeded permutation, truncates it to a multiple of the world size and
    keeps elements rank, rank+world, ... ; ``start_iter`` skips already-consumed samples."""

    def __init__(self, dataset, seed=0, num_replicas=None, rank=None):
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            num_replicas = dist.get_world_size()
            rank = dist.get_rank()
        self.dataset_len = len(dataset)
        self.num_replicas, self.rank, self.seed = num_replicas, rank, seed
        self.epoch = 0
        self.start_iter = 0
        self.total_size = self.dataset_len - (self.dataset_len % num_replicas)
        self.num_samples = self.total_size // num_replicas

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch + self.seed)
        idx = torch.randperm(self.dataset_len, generator=g).tolist()[:self.total_size]
        idx = idx[self.rank:self.total_size:self.num_replicas]
        assert len(idx) == self.num_samples
        return iter(idx[self.start_iter:])

    def __len__(self):
        return self.num_samples

    def set_start_iter(self, start_iter):
        self.start_iter = start_iter

    def set_epoch(self, epoch):
        self.epoch = epoch
