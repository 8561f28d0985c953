from .backbones import *  # noqa: F401,F403
from .builder import (BACKBONES, DEPTHER, HEADS, LOSSES, MODELS, NECKS, build_backbone, build_depther, build_head,
                      build_loss, build_neck)
from .decode_heads import *  # noqa: F401,F403
from .depther import *  # noqa: F401,F403
from .losses import *  # noqa: F401,F403
from .necks import *  # noqa: F401,F403

__all__ = ['BACKBONES', 'HEADS', 'NECKS', 'LOSSES', 'DEPTHER', 'MODELS', 'build_backbone', 'build_head',
           'build_neck', 'build_loss', 'build_depther']
