from .base import BaseDepther, DeferredLogVars
from .encoder_decoder import DepthEncoderDecoder

__all__ = ['BaseDepther', 'DeferredLogVars', 'DepthEncoderDecoder']
