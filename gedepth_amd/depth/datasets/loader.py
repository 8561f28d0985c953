"""Batching for dict samples (mmcv ``collate`` semantics for this path: tensors stacked, ``img_metas`` kept as a
list) and a DistributedSampler-backed loader (depth/datasets/builder.py:93-149)."""
import torch
from torch.utils.data import DataLoader, DistributedSampler

from ...mmrt.runner import get_dist_info


def collate(samples):
    """Train samples: tensors stacked, ``img_metas`` a list of dicts.  Test-time-augmentation samples (every value a list
    over augmentations, pipelines/test_time_aug.py): a list over augmentations of stacked tensors / meta lists — the
    ``forward_test(imgs, img_metas)`` protocol of depther/base.py:64-90."""
    out = {}
    for k in samples[0]:
        vals = [s[k] for s in samples]
        if isinstance(vals[0], list):                     # TTA: transpose [sample][aug] -> [aug][sample]
            n_aug = len(vals[0])
            per_aug = [[v[a] for v in vals] for a in range(n_aug)]
            if k == 'img_metas':
                out[k] = per_aug
            else:
                out[k] = [torch.stack(a, 0) if torch.is_tensor(a[0]) else torch.as_tensor(a) for a in per_aug]
        elif k == 'img_metas':
            out[k] = vals
        elif torch.is_tensor(vals[0]):
            out[k] = torch.stack(vals, 0)
        else:
            out[k] = torch.as_tensor(vals)
    return out


def worker_init_fn(worker_id, num_workers, rank, seed):
    """depth/datasets/builder.py:152-157: every worker of every rank gets its own stream — the transforms draw from
    ``np.random`` (Resize ratio, RandomRotate, RandomFlip, RandomCrop, ColorAug), which torch does not re-seed in forked
    workers.  seed=None: derive from the worker's torch seed (distinct per worker and epoch-stable)."""
    import random

    import numpy as np
    worker_seed = (num_workers * rank + worker_id + seed) if seed is not None else (torch.initial_seed() + rank) % 2 ** 32
    np.random.seed(worker_seed % 2 ** 32)
    random.seed(worker_seed)
    torch.manual_seed(worker_seed)


def build_dataloader(dataset, samples_per_gpu, workers_per_gpu=0, dist=True, shuffle=True, seed=None, drop_last=False,
                     pin_memory=True, **kwargs):
    from functools import partial
    rank, world = get_dist_info()
    sampler = DistributedSampler(dataset, world, rank, shuffle=shuffle, seed=seed or 0) if dist else None
    init_fn = partial(worker_init_fn, num_workers=workers_per_gpu, rank=rank, seed=seed)
    return DataLoader(dataset, batch_size=samples_per_gpu, sampler=sampler, shuffle=(shuffle and sampler is None),
                      num_workers=workers_per_gpu, collate_fn=collate, pin_memory=pin_memory, drop_last=drop_last,
                      worker_init_fn=init_fn,
                      persistent_workers=workers_per_gpu > 0)      # the iter-based runner re-enters the loader every epoch


class SyntheticKITTI(torch.utils.data.Dataset):
    """Fixed-size synthetic KITTI-shaped dataset (one seeded sample per index)."""

    def __init__(self, length=64, height=352, width=1120, adaptive=True, seed=1234):
        self.length, self.h, self.w, self.adaptive, self.seed = length, height, width, adaptive, seed
        self._cache = {}                          # generating a sample costs ~0.25 s of host time: do it once per index

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        if i not in self._cache:
            from .synthetic import synthetic_batch
            b = synthetic_batch(1, self.h, self.w, seed=self.seed + i)
            out = dict(img=b['img'][0], img_metas=b['img_metas'][0], depth_gt=b['depth_gt'][0])
            if self.adaptive:
                out['pe_k_gt'] = b['pe_k_gt'][0]
            self._cache[i] = out
        return dict(self._cache[i])
