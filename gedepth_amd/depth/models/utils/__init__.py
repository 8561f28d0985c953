from .ckpt_convert import swin_convert
from .embed import PatchEmbedSwin
from .layout import to_channels_last

__all__ = ['swin_convert', 'PatchEmbedSwin', 'to_channels_last']
