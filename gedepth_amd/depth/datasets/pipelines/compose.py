"""Pipeline composition (depth/datasets/pipelines/compose.py:8-49): a list of transform configs / callables applied in
order; a transform returning ``None`` aborts the sample."""
from ..builder import PIPELINES
from ....mmrt.registry import build_from_cfg


class Compose:

    def __init__(self, transforms):
        self.transforms = []
        for t in transforms:
            if isinstance(t, dict):
                self.transforms.append(build_from_cfg(t, PIPELINES))
            elif callable(t):
                self.transforms.append(t)
            else:
                raise TypeError('transform must be callable or a dict')

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
            if data is None:
                return None
        return data

    def __repr__(self):
        return self.__class__.__name__ + '(' + ''.join(f'\n    {t}' for t in self.transforms) + '\n)'
