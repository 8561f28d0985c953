"""Test-time augmentation wrapper (depth/datasets/pipelines/test_time_aug.py:9-134): runs the inner transforms once per
(scale, flip, direction) with those values preset in the sample, and returns a dict of lists."""
import warnings

from ..builder import PIPELINES
from .compose import Compose


@PIPELINES.register_module()
class MultiScaleFlipAug:

    def __init__(self, transforms, img_scale, img_ratios=None, flip=False, flip_direction='horizontal'):
        self.transforms = Compose(transforms)
        if img_ratios is not None:
            img_ratios = img_ratios if isinstance(img_ratios, list) else [img_ratios]
            assert all(isinstance(r, float) for r in img_ratios)
        if img_scale is None:
            self.img_scale = None
            assert img_ratios is not None
        elif isinstance(img_scale, tuple) and img_ratios is not None:
            assert len(img_scale) == 2
            self.img_scale = [(int(img_scale[0] * r), int(img_scale[1] * r)) for r in img_ratios]
        else:
            self.img_scale = img_scale if isinstance(img_scale, list) else [img_scale]
        assert self.img_scale is None or all(isinstance(s, tuple) for s in self.img_scale)
        self.flip, self.img_ratios = flip, img_ratios
        self.flip_direction = flip_direction if isinstance(flip_direction, list) else [flip_direction]
        if not self.flip and self.flip_direction != ['horizontal']:
            warnings.warn('flip_direction has no effect when flip is set to False')
        if self.flip and not any(t['type'] == 'RandomFlip' for t in transforms):
            warnings.warn('flip has no effect when RandomFlip is not in transforms')

    def __call__(self, results):
        if self.img_scale is None:
            h, w = results['img'].shape[:2]
            img_scale = [(int(w * r), int(h * r)) for r in self.img_ratios]
        else:
            img_scale = self.img_scale
        aug_data = []
        for scale in img_scale:
            for flip in ([False, True] if self.flip else [False]):
                for direction in self.flip_direction:
                    _results = results.copy()
                    _results['scale'], _results['flip'], _results['flip_direction'] = scale, flip, direction
                    aug_data.append(self.transforms(_results))
        return {key: [d[key] for d in aug_data] for key in aug_data[0]}

    def __repr__(self):
        return (f'{self.__class__.__name__}(transforms={self.transforms}, img_scale={self.img_scale}, flip={self.flip})'
                f'flip_direction={self.flip_direction}')
