"""gedepth_amd — MI355X-native implementation of the GEDepth training hot path.

``gedepth_amd.depth``  host-side mirror of the reference's ``depth`` package (registry / config surface)
``gedepth_amd.mmrt``   mmcv-free runtime: configs, registries, bricks, runner, hooks, DDP
``gedepth_amd.hip``    ctypes binding of the C-ABI kernel library (include/gedepth_hip.h)
``gedepth_amd.kernels`` autograd wrappers of the HIP kernels
"""
from .utils_version import __version__

__all__ = ['__version__']
