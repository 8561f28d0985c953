"""Ground-attention and slope-logit necks.

``LightPEMASKNeck`` mirrors depth/models/necks/pemask_neck.py:29-64 and ``DynamicPENeckSOFT`` mirrors
depth/models/necks/dynamicpe_neck.py:490-539 (keys ``conv0..conv4``, ``convfinal``).  The reference
hard-codes the Swin-L widths 1536/768/384/192/64; ``in_channels`` defaults to that list so the
reference configs build unchanged, and lets Swin-T models (BASELINE configs #1/#2) be expressed.
"""
import torch
import torch.nn as nn

from ....mmrt.bricks import BaseModule, conv_bias_act, xavier_init
from ...ops import resize
from ..builder import NECKS

_SWIN_L = (1536, 768, 384, 192, 64)


class _PETrunk(BaseModule):
    """sum_i up_ac(conv3x3_i(x_i)) at the finest resolution, then a final 3x3 conv."""

    def __init__(self, out_channels, in_channels=None):
        super().__init__()
        in_channels = list(in_channels) if in_channels is not None else list(_SWIN_L)
        assert len(in_channels) == 5, 'coarse-to-fine channel list of the five feature levels'
        self.convfinal = nn.Conv2d(64, out_channels, kernel_size=3, padding=1, stride=1)
        for i, c in enumerate(in_channels):
            setattr(self, f'conv{i}', nn.Conv2d(c, 64, kernel_size=3, padding=1, stride=1))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                xavier_init(m, distribution='uniform')
        self._is_init = True

    @staticmethod
    def _conv(conv, x):
        return conv_bias_act(conv, x) if x.is_cuda else conv(x)

    def trunk(self, inputs):
        xs = inputs[::-1]                       # coarse -> fine
        ts = [self._conv(getattr(self, f'conv{i}'), xs[i]) for i in range(5)]
        if ts[4].is_cuda:                       # four up-samplings + four adds in one pass over the fine map (kernels.upsum)
            from ....kernels import upsum
            return upsum(ts[4], ts[:4], align_corners=True)
        size = ts[4].shape[2:]
        acc = None
        for i in range(5):
            t = ts[i] if i == 4 else resize(ts[i], size=size, mode='bilinear', align_corners=True)
            acc = t if acc is None else acc + t
        return acc


@NECKS.register_module()
class LightPEMASKNeck(_PETrunk):

    def __init__(self, in_channels=None):
        super().__init__(1, in_channels)
        self.sigmoid = nn.Sigmoid()

    def forward(self, inputs):
        x = self.trunk(inputs)
        return self.sigmoid(self._conv(self.convfinal, x)), x


@NECKS.register_module()
class DynamicPENeckSOFT(_PETrunk):

    def __init__(self, in_channels=None):
        super().__init__(11, in_channels)

    def forward(self, inputs):
        return self._conv(self.convfinal, self.trunk(inputs))
