"""Dataset / pipeline registries and builders (depth/datasets/builder.py:23-90)."""
from ...mmrt.registry import Registry, build_from_cfg

DATASETS = Registry('dataset')
PIPELINES = Registry('pipeline')


def build_dataset(cfg, default_args=None):
    return build_from_cfg(cfg, DATASETS, default_args)
