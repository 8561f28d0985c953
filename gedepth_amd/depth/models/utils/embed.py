"""4-channel patch embedding (mirror of depth/models/utils/embed.py:201-302).

Keys: ``projection.{weight,bias}`` (Conv2d k4 s4) and ``norm.{weight,bias}``.
"""
import torch.nn.functional as F

from ....mmrt.bricks import BaseModule, build_conv_layer, build_norm_layer


class PatchEmbedSwin(BaseModule):

    def __init__(self, in_channels=3, embed_dims=768, conv_type=None, kernel_size=16, stride=16, padding=0,
                 dilation=1, pad_to_patch_size=True, norm_cfg=None, init_cfg=None):
        super().__init__()
        self.embed_dims, self.init_cfg = embed_dims, init_cfg
        stride = kernel_size if stride is None else stride
        self.pad_to_patch_size = pad_to_patch_size
        self.patch_size = kernel_size if isinstance(kernel_size, tuple) else (kernel_size, kernel_size)
        self.projection = build_conv_layer(dict(type=conv_type or 'Conv2d'), in_channels=in_channels,
                                           out_channels=embed_dims, kernel_size=kernel_size, stride=stride,
                                           padding=padding, dilation=dilation)
        self.norm = build_norm_layer(norm_cfg, embed_dims)[1] if norm_cfg is not None else None
        if self.norm is not None and hasattr(self.norm, 'autocast_out'):
            self.norm.autocast_out = False          # its output is the fp32 residual stream of stage 0 (as under plain autocast)
        self.DH = self.DW = None

    def forward(self, x):
        """(B, Cin, H, W) -> tokens (B, DH*DW, C), (DH, DW)."""
        H, W = x.shape[2], x.shape[3]
        ph, pw = self.patch_size
        if self.pad_to_patch_size and (H % ph or W % pw):
            x = F.pad(x, (0, (pw - W % pw) % pw, 0, (ph - H % ph) % ph))
        x = self.projection(x)
        self.DH, self.DW = x.shape[2], x.shape[3]
        x = x.flatten(2).transpose(1, 2)
        if self.norm is not None:
            x = self.norm(x)
        return x, (self.DH, self.DW)
