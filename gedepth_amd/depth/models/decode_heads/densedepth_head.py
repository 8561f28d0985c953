"""DenseDepth up-convolution decoder (mirror of depth/models/decode_heads/densedepth_head.py:14-131).

Keys: ``conv_list.0.conv.*`` (1x1), ``conv_list.{1..4}.{convA,convB}.conv.*``, ``conv_depth.*``.
"""
import torch
import torch.nn as nn

from ....mmrt.bricks import ConvModule
from ...ops import resize
from ..builder import HEADS
from .decode_head import DepthBaseDecodeHead


class UpSample(nn.Sequential):
    """bilinear(align_corners=True) up -> concat skip -> conv3x3+act -> conv3x3+act."""

    def __init__(self, skip_input, output_features, conv_cfg=None, norm_cfg=None, act_cfg=None):
        super().__init__()
        kw = dict(kernel_size=3, stride=1, padding=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.convA = ConvModule(skip_input, output_features, **kw)
        self.convB = ConvModule(output_features, output_features, **kw)

    def forward(self, x, concat_with):
        if x.is_cuda:                       # up-sampling written straight into the concat buffer (kernels.upcat, csrc/decoder.hip)
            from ....kernels import upcat
            return self.convB(self.convA(upcat(x, concat_with.to(x.dtype), align_corners=True)))
        up = resize(x, size=concat_with.shape[2:], mode='bilinear', align_corners=True)
        return self.convB(self.convA(torch.cat([up, concat_with.to(up.dtype)], dim=1)))


@HEADS.register_module()
class DenseDepthHead(DepthBaseDecodeHead):

    def __init__(self, up_sample_channels, fpn=False, conv_dim=256, **kwargs):
        super().__init__(**kwargs)
        if fpn:
            raise NotImplementedError('the FPN variant is not used by the GEDepth configs')
        self.fpn = fpn
        self.up_sample_channels = up_sample_channels[::-1]
        self.in_channels = self.in_channels[::-1]
        self.conv_list = nn.ModuleList()
        prev = 0
        for index, (cin, cup) in enumerate(zip(self.in_channels, self.up_sample_channels)):
            if index == 0:
                self.conv_list.append(ConvModule(cin, cup, kernel_size=1, stride=1, padding=0, act_cfg=None))
            else:
                self.conv_list.append(UpSample(cin + prev, cup, norm_cfg=self.norm_cfg, act_cfg=self.act_cfg))
            prev = cup

    def forward(self, inputs, img_metas, pe_mask, depth_mask_y):
        feats = inputs[::-1]
        x = self.conv_list[0](feats[0])
        for index in range(1, len(feats)):
            x = self.conv_list[index](x, feats[index])
        return self.depth_pred(x, pe_mask, depth_mask_y)
