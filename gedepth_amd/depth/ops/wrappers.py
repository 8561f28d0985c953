"""``resize`` — the F.interpolate wrapper of depth/ops/wrappers.py:7-26, on the HIP bilinear kernel."""
import torch.nn.functional as F

from ...kernels import bilinear_resize


def resize(input, size=None, scale_factor=None, mode='nearest', align_corners=None, warning=False):
    if mode == 'bilinear' and size is not None and input.is_cuda:
        return bilinear_resize(input, size, bool(align_corners))
    if mode == 'bilinear' and input.is_cuda:
        size = [int(input.shape[2] * scale_factor), int(input.shape[3] * scale_factor)]
        return bilinear_resize(input, size, bool(align_corners))
    # non-bilinear modes (bicubic bias-table resize at checkpoint load time) are host-side plumbing
    return F.interpolate(input, size, scale_factor, mode, align_corners)
