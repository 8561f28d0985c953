from .ckpt_convert import swin_convert
from .embed import PatchEmbedSwin

__all__ = ['swin_convert', 'PatchEmbedSwin']
