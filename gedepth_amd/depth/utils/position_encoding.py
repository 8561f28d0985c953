"""SinePositionalEncoding (depth/utils/position_encoding.py:10-89), cached.

The reference recomputes the embedding from an all-false mask on every forward; it depends only
on (h, w), so it is computed once per shape/device and kept resident.
"""
import math

import torch

from ...mmrt.bricks import POSITIONAL_ENCODING, BaseModule


@POSITIONAL_ENCODING.register_module()
class SinePositionalEncoding(BaseModule):

    def __init__(self, num_feats, temperature=10000, normalize=False, scale=2 * math.pi, eps=1e-6, offset=0.,
                 init_cfg=None):
        super().__init__(init_cfg)
        if normalize:
            assert isinstance(scale, (float, int))
        self.num_feats, self.temperature, self.normalize = num_feats, temperature, normalize
        self.scale, self.eps, self.offset = scale, eps, offset
        self._cache = {}

    def grid(self, h, w, device):
        """(1, 2*num_feats, h, w) embedding of an unmasked h x w map."""
        key = (h, w, str(device))
        if key not in self._cache:
            y = torch.arange(1, h + 1, dtype=torch.float32, device=device).view(h, 1).expand(h, w)
            x = torch.arange(1, w + 1, dtype=torch.float32, device=device).view(1, w).expand(h, w)
            if self.normalize:
                y = (y + self.offset) / (h + self.eps) * self.scale
                x = (x + self.offset) / (w + self.eps) * self.scale
            dim_t = torch.arange(self.num_feats, dtype=torch.float32, device=device)
            dim_t = self.temperature ** (2 * torch.div(dim_t, 2, rounding_mode='floor') / self.num_feats)
            px, py = x[:, :, None] / dim_t, y[:, :, None] / dim_t
            px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).view(h, w, -1)
            py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).view(h, w, -1)
            self._cache[key] = torch.cat((py, px), dim=2).permute(2, 0, 1).unsqueeze(0).contiguous()
        return self._cache[key]

    def forward(self, mask):
        """mask (B,h,w) — must be all-false on this path (it always is: necks/hahi.py:259,297)."""
        B, h, w = mask.shape
        return self.grid(h, w, mask.device).expand(B, -1, -1, -1)
