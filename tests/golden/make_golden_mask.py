"""Generate tests/golden/dynamic_pe_mask.npz: the 0/1 ``pe_offset_mask`` of the reference's
DepthEncoderDecoder.dynamic_pe (depther/encoder_decoder.py:95-100) — the "integer pixel mask" of the north star —
by calling the reference method itself (imported through the stand-in of make_golden.py).  Build-container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_mask.py

The method does not return the mask; it returns ``pe_mask = (pe_offset * pe_offset_mask) * y``.  Calling it with
``y == 1`` gives ``pe_offset * pe_offset_mask``, from which the mask is read off exactly: mask = 1 <=> 0 < offset <= 200
<=> the product is > 0 (a masked pixel gives exactly 0, an unmasked one its positive offset).

Cases: 'a' = the inputs of dynamic_pe.npz (2 x 24 x 40, default height 1.65 and per-sample DDAD heights);
'b' = 2 x 88 x 280 with per-pixel noise on the ground depth, so that ~50 k pixels straddle both mask thresholds."""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_golden as MG  # noqa: E402


class _Neck(nn.Module):
    def __init__(self, t):
        super().__init__()
        self.t = t

    def forward(self, x):
        return self.t


def main():
    MG.install_shim()
    import depth.models  # noqa: F401
    from depth.models import build_depther
    from gedepth_amd.mmrt.config import Config, ConfigDict
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    ch = [64, 192, 384, 768, 1536]
    model = dict(
        type='DepthEncoderDecoder', pretrained=None,
        backbone=dict(type='DepthFormerSwin', pretrain_img_size=224, patch_size=4, window_size=7, mlp_ratio=4,
                      strides=(4, 2, 2, 2), out_indices=(0, 1, 2, 3), qkv_bias=True, qk_scale=None, patch_norm=True,
                      drop_rate=0., attn_drop_rate=0., drop_path_rate=0.0, use_abs_pos_embed=False,
                      act_cfg=dict(type='GELU'), norm_cfg=dict(type='LN', requires_grad=True), pretrain_style='official',
                      conv_norm_cfg=dict(type='BN', requires_grad=True), depth=50, num_stages=0, USEPE=True,
                      embed_dims=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48]),
        neck=dict(type='HAHIHeteroNeck', positional_encoding=dict(type='SinePositionalEncoding', num_feats=256),
                  in_channels=ch, out_channels=ch, embedding_dim=512, scales=[1, 1, 1, 1, 1]),
        pe_mask_neck=dict(type='LightPEMASKNeck'), dynamic_pe_neck=dict(type='DynamicPENeckSOFT'),
        decode_head=dict(type='DenseDepthHead', act_cfg=dict(type='LeakyReLU', inplace=True), in_channels=ch,
                         up_sample_channels=ch, channels=64, align_corners=True, min_depth=1e-3, max_depth=80,
                         loss_decode=dict(type='SigLoss', valid_mask=True, loss_weight=1.0)),
        train_cfg=dict(), test_cfg=dict(mode='whole'))
    with torch.enable_grad():
        dm = build_depther(ConfigDict(Config(dict(model=model)).model))
    arrays = {}
    for tag, (H, W, seed, gseed) in {'a': (24, 40, 5, 13), 'b': (88, 280, 6, 14)}.items():
        g = MG.gen(gseed)
        img, _, _ = MG.synth_img(2, H, W, seed=seed)
        img[:, 4] += 0.05 * torch.randn(2, H, W, generator=g)
        logits_lr = 2.0 * torch.randn(2, 11, H // 2, W // 2, generator=g)
        _ = torch.rand(2, 1, H, W, generator=g)            # keeps case 'a' on the stream of dynamic_pe.npz
        dm.dynamic_pe_neck = _Neck(logits_lr)
        ones = torch.ones(2, 1, H, W)
        heights = torch.tensor([1.56, 1.53])
        prod, _ = dm.dynamic_pe(None, ones, img, None)
        prod_h, _ = dm.dynamic_pe(None, ones, img, None, height=heights)
        assert torch.isfinite(prod).all() and torch.isfinite(prod_h).all()
        arrays.update({f'{tag}_pe_raw': img[:, 4], f'{tag}_logits_lr': logits_lr, f'{tag}_heights': heights,
                       f'{tag}_mask': (prod > 0).to(torch.uint8), f'{tag}_mask_h': (prod_h > 0).to(torch.uint8),
                       f'{tag}_offset_masked': prod, f'{tag}_offset_masked_h': prod_h})
        print(tag, 'mask ones:', int((prod > 0).sum()), 'of', prod.numel(), '; with heights:', int((prod_h > 0).sum()))
    if os.path.isfile(os.path.join(HERE, 'dynamic_pe.npz')):
        old = np.load(os.path.join(HERE, 'dynamic_pe.npz'))
        assert np.array_equal(old['img'][:, 4], arrays['a_pe_raw'].numpy()) and np.array_equal(old['logits_lr'], arrays['a_logits_lr'].numpy())
    MG.save('dynamic_pe_mask', **arrays)


if __name__ == '__main__':
    main()
