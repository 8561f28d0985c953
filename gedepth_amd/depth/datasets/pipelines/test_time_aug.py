"""Test-time augmentation wrapper (role of depth/datasets/pipelines/test_time_aug.py:9-134): the inner transforms run once
per (scale, flip, direction) combination with those three values preset in the sample, and the per-view dicts are merged
into one dict of lists."""
import itertools
import warnings

from ..builder import PIPELINES
from .compose import Compose


def _as_list(v):
    return v if isinstance(v, list) else [v]


@PIPELINES.register_module()
class MultiScaleFlipAug:

    def __init__(self, transforms, img_scale, img_ratios=None, flip=False, flip_direction='horizontal'):
        self.transforms = Compose(transforms)
        self.img_ratios = None if img_ratios is None else _as_list(img_ratios)
        if self.img_ratios is not None and not all(isinstance(r, float) for r in self.img_ratios):
            raise AssertionError('img_ratios must be floats')
        if img_scale is None:                                   # ratios of the input size, resolved per sample
            assert self.img_ratios is not None
            self.img_scale = None
        elif isinstance(img_scale, tuple) and self.img_ratios is not None:
            assert len(img_scale) == 2                          # one base scale times each ratio
            self.img_scale = [tuple(int(side * r) for side in img_scale) for r in self.img_ratios]
        else:                                                   # explicit scale(s)
            self.img_scale = _as_list(img_scale)
        assert self.img_scale is None or all(isinstance(s, tuple) for s in self.img_scale)
        self.flip = flip
        self.flip_direction = _as_list(flip_direction)
        if not flip and self.flip_direction != ['horizontal']:
            warnings.warn('flip_direction has no effect when flip is set to False')
        if flip and all(t['type'] != 'RandomFlip' for t in transforms):
            warnings.warn('flip has no effect when RandomFlip is not in transforms')

    def __call__(self, results):
        scales = self.img_scale
        if scales is None:
            h, w = results['img'].shape[:2]
            scales = [(int(w * r), int(h * r)) for r in self.img_ratios]
        views = []
        for scale, flip, direction in itertools.product(scales, (False, True) if self.flip else (False,), self.flip_direction):
            view = dict(results, scale=scale, flip=flip, flip_direction=direction)
            views.append(self.transforms(view))
        return {key: [v[key] for v in views] for key in views[0]}

    def __repr__(self):
        return (f'{type(self).__name__}(transforms={self.transforms}, img_scale={self.img_scale}, flip={self.flip}, '
                f'flip_direction={self.flip_direction})')
