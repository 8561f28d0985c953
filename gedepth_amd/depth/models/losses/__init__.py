"""Losses built by the GEDepth decode head: SigLoss (depth/models/losses/sigloss.py:8-69),
CrossEntropyLoss(ignore_index=255) (losses/celoss.py:354-413) and BinaryCrossEntropyLoss
(losses/bceloss.py; constructed by every head, decode_head.py:361-364, never evaluated on this path)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ....kernels import silog_loss
from ..builder import LOSSES


@LOSSES.register_module()
class SigLoss(nn.Module):
    """sqrt(var(g) + 0.15 mean(g)^2), g = log(pred+eps) - log(gt+eps) over gt > 0 (unbiased variance)."""

    def __init__(self, loss_name='loss_sig', valid_mask=True, loss_weight=1.0, max_depth=None, warm_up=False,
                 warm_iter=100):
        super().__init__()
        if warm_up:
            raise NotImplementedError('SigLoss warm-up is not used by the GEDepth configs')
        if not valid_mask or max_depth is not None:
            raise NotImplementedError('GEDepth configs use valid_mask=True, max_depth=None')
        self._loss_name, self.valid_mask, self.loss_weight, self.max_depth = loss_name, valid_mask, loss_weight, max_depth
        self.eps = 0.001

    def forward(self, depth_pred, depth_gt, **kwargs):
        return silog_loss(depth_pred, depth_gt, self.eps, self.loss_weight)

    @property
    def loss_name(self):
        return self._loss_name


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):

    def __init__(self, loss_weight=1.0):
        super().__init__()
        self.loss_weight = loss_weight
        self.loss = nn.CrossEntropyLoss(ignore_index=255)

    def forward(self, input, target):
        return self.loss_weight * self.loss(input.float(), target)


@LOSSES.register_module()
class BinaryCrossEntropyLoss(nn.Module):

    def __init__(self, loss_weight=1.0):
        super().__init__()
        self.loss_weight = loss_weight

    def forward(self, input, target):
        return self.loss_weight * F.binary_cross_entropy(input.float(), target.float())
