"""DepthEncoderDecoder: backbone -> HAHI neck -> PE necks -> ground embedding -> decode head -> losses.

Interface mirror of depth/models/depther/encoder_decoder.py:21-274 (constructor kwargs, call protocol,
returned keys).  The ground-embedding math of ``dynamic_pe`` (:79-102) and the vanilla branch (:111-123),
including the align_corners=False up-sampling of ``y`` and of the 11 slope logits, is one fused HIP kernel
per direction (gedepth_amd/csrc/ground.hip); nothing in the constructor touches a device (reference :68).
"""
import torch

from ....kernels import ground_embed_adaptive, ground_embed_vanilla
from ...core import add_prefix
from ...ops import resize
from .. import builder
from ..builder import DEPTHER
from .base import BaseDepther


@DEPTHER.register_module()
class DepthEncoderDecoder(BaseDepther):

    def __init__(self, backbone, decode_head, neck=None, pe_mask_neck=None, dynamic_pe_neck=None, train_cfg=None,
                 test_cfg=None, pretrained=None, init_cfg=None, depth_scale=200):
        super().__init__(init_cfg)
        if pretrained is not None:
            assert backbone.get('pretrained') is None, 'both backbone and depther set pretrained weight'
            backbone['pretrained'] = pretrained
        self.backbone = builder.build_backbone(backbone)
        self.decode_head = builder.build_head(decode_head)
        self.align_corners = self.decode_head.align_corners
        self.depth_scale = depth_scale
        self.pe_mask_neck_FLAGS = pe_mask_neck is not None
        self.dynamic_pe_neck_FLAGS = dynamic_pe_neck is not None
        if neck is not None:
            self.neck = builder.build_neck(neck)
        if pe_mask_neck is not None:
            self.pe_mask_neck = builder.build_neck(pe_mask_neck)
        if dynamic_pe_neck is not None:
            self.dynamic_pe_neck = builder.build_neck(dynamic_pe_neck)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        assert self.with_decode_head
        self.last_valid_mask = None      # u8 (B,H,W) "pe_offset_mask" of the latest adaptive forward

    # ------------------------------------------------------------------ ground embedding
    @staticmethod
    def _height(kwargs):
        if 'height' not in kwargs:
            return None
        h = kwargs['height'][0] if 'test' in kwargs else kwargs['height']       # reference :88-92
        return h.reshape(-1).float()

    def dynamic_pe(self, x, y_lr, img, img_metas, **kwargs):
        """-> (pe_mask, slope logits at image resolution, y at image resolution)."""
        logits_lr = self.dynamic_pe_neck(x)
        pe_mask, logits_hr, y_hr, valid = ground_embed_adaptive(logits_lr, y_lr, img, self._height(kwargs),
                                                                self.depth_scale)
        self.last_valid_mask = valid
        return pe_mask, logits_hr, y_hr

    def extract_feat(self, img, img_metas, **kwargs):
        img = img.float().contiguous()
        x = self.backbone(img)
        if not self.with_neck:
            return x, None, None, None
        x = self.neck(x)
        if not self.pe_mask_neck_FLAGS:
            return x, None, None, None
        y_lr, _ = self.pe_mask_neck(x)
        if self.dynamic_pe_neck_FLAGS:
            pe_mask, logits_hr, y = self.dynamic_pe(x, y_lr, img, img_metas, **kwargs)
            return x, y, pe_mask, logits_hr
        pe_mask, y = ground_embed_vanilla(y_lr, img, 200.0)       # hard-coded 200 (reference :122)
        return x, y, pe_mask, None

    # ------------------------------------------------------------------------- train / test
    def encode_decode(self, img, img_metas, rescale=True, **kwargs):
        x, y, pe_mask, _ = self.extract_feat(img, img_metas, **kwargs)
        out = self.decode_head.forward_test(img, x, img_metas, self.test_cfg, pe_mask, y, **kwargs)
        out = torch.clamp(out, min=self.decode_head.min_depth, max=self.decode_head.max_depth)   # clamp THEN resize
        if rescale:
            out = resize(out, size=img.shape[2:], mode='bilinear', align_corners=self.align_corners)
        return out

    def forward_dummy(self, img):
        return self.encode_decode(img, None)

    def forward_train(self, img, img_metas, depth_gt, **kwargs):
        x, y, pe_mask, pe_offset = self.extract_feat(img, img_metas, **kwargs)
        loss_decode = self.decode_head.forward_train(img, x, img_metas, depth_gt, self.train_cfg, pe_mask, y, pe_offset,
                                                     **kwargs)
        return dict(add_prefix(loss_decode, 'decode'))

    def whole_inference(self, img, img_meta, rescale, **kwargs):
        return self.encode_decode(img, img_meta, rescale, **kwargs)

    def inference(self, img, img_meta, rescale, **kwargs):
        assert self.test_cfg.mode in ['slide', 'whole']
        ori_shape = img_meta[0]['ori_shape']
        assert all(m['ori_shape'] == ori_shape for m in img_meta)
        if self.test_cfg.mode == 'slide':
            raise NotImplementedError
        output = self.whole_inference(img, img_meta, rescale, **kwargs)
        if img_meta[0]['flip']:
            direction = img_meta[0]['flip_direction']
            assert direction in ['horizontal', 'vertical']
            output = output.flip(dims=(3,) if direction == 'horizontal' else (2,))
        return output

    def simple_test(self, img, img_meta, rescale=True, **kwargs):
        return list(self.inference(img, img_meta, rescale, **kwargs).cpu().numpy())

    def aug_test(self, imgs, img_metas, rescale=True, **kwargs):
        """Flip-TTA average (reference :249-274); only rescale=True is supported."""
        assert rescale
        depth_pred = None
        for i in range(len(imgs)):
            kw = dict(kwargs)
            if 'pe_ori_point' in kwargs:
                kw['pe_ori_point_test'] = kwargs['pe_ori_point'][i]
            if 'pe_k_gt' in kwargs:
                kw['pe_k_gt_test'] = kwargs['pe_k_gt'][i]
            cur = self.inference(imgs[i], img_metas[i], rescale, **kw)
            depth_pred = cur if depth_pred is None else depth_pred + cur
        depth_pred = depth_pred / len(imgs)
        return list(depth_pred.cpu().numpy())
