from .depthformer_swin import DepthFormerSwin

__all__ = ['DepthFormerSwin']
