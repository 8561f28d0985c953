"""Base decode head: depth fusion formula and loss plumbing.

Interface mirror of ``DepthBaseDecodeHead`` (depth/models/decode_heads/decode_head.py:268-648) for the
regression path the GEDepth configs use (``classify=False, scale_up=False, depth2norm=False``).
``depth_pred`` (reference :489-508) is one fused kernel: ReLU, the two align_corners=True down-resizes of
``pe_mask`` and ``y`` and ``d*(1-y)+pe+min_depth`` (gedepth_amd/csrc/ground.hip: ge_depth_fuse_*).
``log_images`` (reference :628-648, a device->host sync per iteration whose consumer hook is deleted by
the configs) is opt-in.
"""
from abc import ABCMeta, abstractmethod

import torch
import torch.nn as nn

from .... import kernels
from ....kernels import depth_fuse
from ....mmrt.bricks import BaseModule
from ...ops import resize
from ..builder import build_loss


class DepthBaseDecodeHead(BaseModule, metaclass=ABCMeta):

    def __init__(self, in_channels, channels=96, conv_cfg=None, act_cfg=dict(type='ReLU'),
                 loss_decode=dict(type='SigLoss', valid_mask=True, loss_weight=10),
                 loss_pe=dict(type='BinaryCrossEntropyLoss', loss_weight=10),
                 loss_dynamic_pe=dict(type='CrossEntropyLoss', loss_weight=0.08), loss_surface_norm=None,
                 sampler=None, align_corners=False, min_depth=1e-3, max_depth=None, norm_cfg=None, classify=False,
                 n_bins=256, bins_strategy='UD', norm_strategy='linear', scale_up=False, depth2norm=False,
                 log_images=False):
        super().__init__()
        if classify or scale_up or depth2norm:
            raise NotImplementedError('classification / scale-up / depth2norm heads are outside the GEDepth path')
        self.in_channels, self.channels, self.conv_cfg, self.act_cfg = in_channels, channels, conv_cfg, act_cfg
        self.loss_decode = build_loss(loss_decode)
        self.loss_pe = build_loss(loss_pe)
        self.loss_dynamic_pe = build_loss(loss_dynamic_pe)
        self.align_corners, self.min_depth, self.max_depth, self.norm_cfg = align_corners, min_depth, max_depth, norm_cfg
        self.classify, self.n_bins, self.scale_up = classify, n_bins, scale_up
        self.conv_depth = nn.Conv2d(channels, 1, kernel_size=3, padding=1, stride=1)
        self.fp16_enabled = False
        self.with_log_images = log_images

    def extra_repr(self):
        return f'align_corners={self.align_corners}'

    @abstractmethod
    def forward(self, inputs, img_metas, pe_mask, depth_mask_y):
        pass

    def forward_train(self, img, inputs, img_metas, depth_gt, train_cfg, pe_mask, y, pe_offset, **kwargs):
        depth_pred, _ = self.forward(inputs, img_metas, pe_mask, y)
        if pe_offset is not None:
            losses = self.losses_dynamic_pe(depth_pred, depth_gt, pe_offset, kwargs['pe_k_gt'], None, None)
        else:
            losses = self.losses(depth_pred, depth_gt)
        if self.with_log_images:
            losses.update(**self.log_images(img[0], depth_pred[0], depth_gt[0], img_metas[0]))
        return losses

    def forward_test(self, img, inputs, img_metas, test_cfg, pe_mask, y, **kwargs):
        return self.forward(inputs, img_metas, pe_mask, y)[0]

    def depth_pred(self, feat, pe, depth_y):
        if feat.is_cuda and kernels.conv3x3_c1_ok(self.conv_depth, feat):   # 64 -> 1: streaming HIP kernel, fp32 result directly
            c = kernels.conv3x3_c1(self.conv_depth, feat, out_fp32=True)
        else:
            c = self.conv_depth(feat).float()
        if pe is None:
            return torch.relu(c) + self.min_depth, depth_y
        if not self.align_corners:
            raise NotImplementedError('GEDepth configs resize with align_corners=True (configs/_base_/models)')
        return depth_fuse(c, pe, depth_y, self.min_depth)

    def losses_dynamic_pe(self, depth_pred, depth_gt, dynamic_pe, pe_k_gt, attn_pred, attn_gt):
        loss = dict()
        depth_pred = resize(depth_pred.float(), size=depth_gt.shape[2:], mode='bilinear', align_corners=self.align_corners)
        loss['loss_dynamic_pe'] = self.loss_dynamic_pe(dynamic_pe, pe_k_gt.long())
        loss['loss_depth'] = self.loss_decode(depth_pred, depth_gt)
        return loss

    def losses(self, depth_pred, depth_gt, **unused):
        depth_pred = resize(depth_pred.float(), size=depth_gt.shape[2:], mode='bilinear', align_corners=self.align_corners)
        return dict(loss_depth=self.loss_decode(depth_pred, depth_gt))

    def log_images(self, img, depth_pred, depth_gt, img_meta):
        """Opt-in (costs a host sync): RGB / normalised pred / normalised gt of the first sample (reference decode_head.py:628-648).
        ``img_rgb`` is (3, H, W) uint8: ``img[:3] * std + mean`` in float32, clipped to [0, 255], truncated.  The reference de-normalises
        with ``to_bgr=to_rgb`` and then reverses the channels once more: for ``to_rgb=True`` (every shipped config) that is the tensor's
        own channel order, for ``to_rgb=False`` the reverse."""
        import numpy as np
        cfg = img_meta['img_norm_cfg']
        show = img.detach()[0:3].permute(1, 2, 0).float().cpu().numpy()
        show = show * np.asarray(cfg['std'], np.float32).reshape(1, -1) + np.asarray(cfg['mean'], np.float32).reshape(1, -1)
        show = np.clip(show, 0, 255).astype(np.uint8).transpose(2, 0, 1)
        if not cfg.get('to_rgb', True):
            show = show[::-1]
        return {'img_rgb': np.ascontiguousarray(show), 'img_depth_pred': (depth_pred / depth_pred.max()).detach().cpu(),
                'img_depth_gt': (depth_gt / depth_gt.max()).detach().cpu()}
