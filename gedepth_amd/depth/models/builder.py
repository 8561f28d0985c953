"""Model registries and ``build_depther`` — mirror of depth/models/builder.py:8-44.

All five model registries alias one ``MODELS`` registry, as in the reference.
"""
import warnings

from ...mmrt.bricks import ATTENTION as MMRT_ATTENTION
from ...mmrt.bricks import MODELS as MMRT_MODELS
from ...mmrt.registry import Registry

MODELS = Registry('models', parent=MMRT_MODELS)
ATTENTION = Registry('attention', parent=MMRT_ATTENTION)

BACKBONES = MODELS
NECKS = MODELS
HEADS = MODELS
LOSSES = MODELS
DEPTHER = MODELS


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_depther(cfg, train_cfg=None, test_cfg=None):
    """Same contract as the reference: train/test cfg may come from either place, never both."""
    if train_cfg is not None or test_cfg is not None:
        warnings.warn('train_cfg and test_cfg is deprecated, please specify them in model', UserWarning)
    assert cfg.get('train_cfg') is None or train_cfg is None, \
        'train_cfg specified in both outer field and model field '
    assert cfg.get('test_cfg') is None or test_cfg is None, \
        'test_cfg specified in both outer field and model field '
    return DEPTHER.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))
