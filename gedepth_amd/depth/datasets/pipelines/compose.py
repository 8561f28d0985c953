"""Pipeline composition (role of depth/datasets/pipelines/compose.py:8-49): config dicts are built through the PIPELINES
registry, callables are taken as they are; a stage that returns ``None`` drops the sample."""
from ....mmrt.registry import build_from_cfg
from ..builder import PIPELINES


def _stage(spec):
    if isinstance(spec, dict):
        return build_from_cfg(spec, PIPELINES)
    if callable(spec):
        return spec
    raise TypeError('transform must be callable or a dict')


class Compose:

    def __init__(self, transforms):
        self.transforms = [_stage(t) for t in transforms]

    def __call__(self, sample):
        for stage in self.transforms:
            sample = stage(sample)
            if sample is None:
                break
        return sample

    def __repr__(self):
        body = ''.join(f'\n    {stage}' for stage in self.transforms)
        return f'{type(self).__name__}({body}\n)'
