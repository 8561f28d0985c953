"""BaseDepther: call protocol, train_step and loss parsing (mirror of depth/models/depther/base.py:16-204).

MI355X-first difference: the reference does one ``all_reduce`` + ``.item()`` per logged scalar per
iteration (base.py:197-202 — 2-3 host syncs per step).  Here the scalars are stacked into one tensor, reduced
with ONE collective, and only brought to the host when a logger actually reads them (``DeferredLogVars``).
"""
from abc import ABCMeta, abstractmethod
from collections import OrderedDict

import torch
import torch.distributed as dist

from ....mmrt.bricks import BaseModule


class DeferredLogVars(OrderedDict):
    """name -> float mapping whose values live in one device tensor until first read."""

    def __init__(self, names, values):
        super().__init__((n, None) for n in names)
        self._names, self._tensor = list(names), values

    def _materialize(self):
        if self._tensor is not None:
            vals = self._tensor.detach().float().cpu().tolist()     # the single host sync
            self._tensor = None
            for n, v in zip(self._names, vals):
                super().__setitem__(n, v)

    def __getitem__(self, k):
        self._materialize()
        return super().__getitem__(k)

    def get(self, k, default=None):
        self._materialize()
        return super().get(k, default)

    def items(self):
        self._materialize()
        return super().items()

    def values(self):
        self._materialize()
        return super().values()

    def tensor(self):
        """Device tensor of the (already all-reduced) values, or None once materialised."""
        return self._tensor


class BaseDepther(BaseModule, metaclass=ABCMeta):

    def __init__(self, init_cfg=None):
        super().__init__(init_cfg)
        self.fp16_enabled = False

    @property
    def with_neck(self):
        return hasattr(self, 'neck') and self.neck is not None

    @property
    def with_auxiliary_head(self):
        return hasattr(self, 'auxiliary_head') and self.auxiliary_head is not None

    @property
    def with_decode_head(self):
        return hasattr(self, 'decode_head') and self.decode_head is not None

    @abstractmethod
    def extract_feat(self, imgs):
        pass

    @abstractmethod
    def encode_decode(self, img, img_metas):
        pass

    @abstractmethod
    def forward_train(self, imgs, img_metas, **kwargs):
        pass

    @abstractmethod
    def simple_test(self, img, img_meta, **kwargs):
        pass

    @abstractmethod
    def aug_test(self, imgs, img_metas, **kwargs):
        pass

    def forward_test(self, imgs, img_metas, **kwargs):
        for var, name in [(imgs, 'imgs'), (img_metas, 'img_metas')]:
            if not isinstance(var, list):
                raise TypeError(f'{name} must be a list, but got {type(var)}')
        num_augs = len(imgs)
        if num_augs != len(img_metas):
            raise ValueError(f'num of augmentations ({len(imgs)}) != num of image meta ({len(img_metas)})')
        for img_meta in img_metas:
            for key in ('ori_shape', 'img_shape', 'pad_shape'):
                vals = [m[key] for m in img_meta if key in m]
                assert all(v == vals[0] for v in vals)
        if num_augs == 1:
            return self.simple_test(imgs[0], img_metas[0], **kwargs)
        return self.aug_test(imgs, img_metas, **kwargs)

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.forward_test(img, img_metas, **kwargs)

    def train_step(self, data_batch, optimizer=None, **kwargs):
        losses = self(**data_batch)
        real_losses = {k: v for k, v in losses.items() if 'img' not in k}
        log_imgs = {k: v for k, v in losses.items() if 'img' in k}
        loss, log_vars = self._parse_losses(real_losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data_batch['img_metas']), log_imgs=log_imgs)

    def val_step(self, data_batch, **kwargs):
        return self(**data_batch, **kwargs)

    @staticmethod
    def _parse_losses(losses):
        log_vars = OrderedDict()
        for name, value in losses.items():
            if isinstance(value, torch.Tensor):
                log_vars[name] = value.mean()
            elif isinstance(value, list):
                log_vars[name] = sum(v.mean() for v in value)
            else:
                raise TypeError(f'{name} is not a tensor or list of tensors')
        loss = sum(v for k, v in log_vars.items() if 'loss' in k)
        log_vars['loss'] = loss
        stacked = torch.stack([v.detach().float() for v in log_vars.values()])
        if dist.is_available() and dist.is_initialized():
            stacked = stacked / dist.get_world_size()
            dist.all_reduce(stacked)                       # one message instead of one per scalar
        return loss, DeferredLogVars(log_vars.keys(), stacked)
