"""Tensor formatting (depth/datasets/pipelines/formating.py:62-143 ImageToTensor / DefaultFormatBundle, :146-220 Collect).
mmcv's DataContainer is not reproduced: samples are plain dicts of tensors plus an ``img_metas`` dict, batched by
``gedepth_amd.depth.datasets.loader.collate`` with the same result (stacked tensors, per-sample meta list)."""
import numpy as np
import torch

from ..builder import PIPELINES


def to_tensor(data):
    if isinstance(data, torch.Tensor):
        return data
    if isinstance(data, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(data))
    if isinstance(data, (int, float)):
        return torch.tensor(data)
    if isinstance(data, (list, tuple)):
        return torch.tensor(data)
    raise TypeError(f'type {type(data)} cannot be converted to tensor.')


@PIPELINES.register_module()
class ImageToTensor:

    def __init__(self, keys):
        self.keys = keys

    def __call__(self, results):
        for key in self.keys:
            img = results[key]
            if img.ndim < 3:
                img = np.expand_dims(img, -1)
            results[key] = to_tensor(img.transpose(2, 0, 1))
        return results

    def __repr__(self):
        return f'{self.__class__.__name__}(keys={self.keys})'


@PIPELINES.register_module()
class DefaultFormatBundle:
    """img -> (C, H, W) tensor; depth_gt -> (1, H, W) tensor."""

    def __call__(self, results):
        if 'img' in results:
            img = results['img']
            if img.ndim < 3:
                img = np.expand_dims(img, -1)
            results['img'] = to_tensor(img.transpose(2, 0, 1))
        if 'depth_gt' in results:
            results['depth_gt'] = to_tensor(results['depth_gt'][None, ...])
        return results

    def __repr__(self):
        return self.__class__.__name__


@PIPELINES.register_module()
class Collect:

    def __init__(self, keys, meta_keys=('filename', 'ori_filename', 'ori_shape', 'img_shape', 'pad_shape', 'scale_factor',
                                        'flip', 'flip_direction', 'img_norm_cfg')):
        self.keys, self.meta_keys = keys, meta_keys

    def __call__(self, results):
        data = {'img_metas': {k: results[k] for k in self.meta_keys}}
        for key in self.keys:
            v = results[key]
            data[key] = to_tensor(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else (
                torch.tensor(float(v)) if isinstance(v, (np.floating, float)) else v)
        return data

    def __repr__(self):
        return f'{self.__class__.__name__}(keys={self.keys}, meta_keys={self.meta_keys})'
