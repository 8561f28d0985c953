from .decode_head import DepthBaseDecodeHead
from .densedepth_head import DenseDepthHead

__all__ = ['DepthBaseDecodeHead', 'DenseDepthHead']
