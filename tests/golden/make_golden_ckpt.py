#!/usr/bin/env python
"""Fixture for released-checkpoint interop: what the REFERENCE's ``DepthFormerSwin.init_weights`` makes
of an official-format Swin checkpoint (key renames, unfold-order fix of the patch-merging weights, bicubic resize of a
5x5-window relative-position table to 7x7, zero-padded 4th input channel of the patch embedding).

Runs only in the build container (imports /root/reference through the mmcv stand-in of make_golden.py):
    python tests/golden/make_golden_ckpt.py        ->  tests/golden/swin_official_ckpt.npz
The fake checkpoint itself is regenerated in the test from ``fake_official_swin(seed)`` below (name-keyed values).
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

ARCH = dict(embed_dims=96, depths=[2, 2, 2, 2], num_heads=[3, 6, 12, 24])


def fake_official_swin(seed=7, window=5):
    """An official-layout (microsoft/Swin-Transformer) Swin-T-like state dict with a 5x5 window, random values.  The
    ``relative_position_index`` buffers are left out: with a different window the reference's 4th-channel padding loop
    (depthformer_swin.py:1113-1123) would index them as 4-d tensors and fail."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    C = ARCH['embed_dims']
    sd = {'patch_embed.proj.weight': r(C, 3, 4, 4), 'patch_embed.proj.bias': r(C), 'patch_embed.norm.weight': r(C),
          'patch_embed.norm.bias': r(C)}
    for s, (d, nh) in enumerate(zip(ARCH['depths'], ARCH['num_heads'])):
        c = C * 2 ** s
        for b in range(d):
            p = f'layers.{s}.blocks.{b}.'
            sd.update({p + 'norm1.weight': r(c), p + 'norm1.bias': r(c), p + 'attn.qkv.weight': r(3 * c, c),
                       p + 'attn.qkv.bias': r(3 * c), p + 'attn.proj.weight': r(c, c), p + 'attn.proj.bias': r(c),
                       p + 'attn.relative_position_bias_table': r((2 * window - 1) ** 2, nh),
                       p + 'norm2.weight': r(c), p + 'norm2.bias': r(c), p + 'mlp.fc1.weight': r(4 * c, c),
                       p + 'mlp.fc1.bias': r(4 * c), p + 'mlp.fc2.weight': r(c, 4 * c), p + 'mlp.fc2.bias': r(c)})
        if s < 3:
            sd.update({f'layers.{s}.downsample.reduction.weight': r(2 * c, 4 * c), f'layers.{s}.downsample.norm.weight': r(4 * c),
                       f'layers.{s}.downsample.norm.bias': r(4 * c)})
    sd.update({'norm.weight': r(8 * C), 'norm.bias': r(8 * C), 'head.weight': r(10, 8 * C), 'head.bias': r(10)})
    return sd


def main():
    import make_golden as G
    G.install_shim()
    import mmcv.runner as mr
    mr._load_checkpoint = lambda path, logger=None, map_location='cpu': torch.load(path, map_location=map_location, weights_only=False)
    import depth.utils as du
    import logging
    du.get_root_logger = lambda *a, **k: logging.getLogger('ref')
    from depth.models.backbones.depthformer_swin import DepthFormerSwin
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'swin_official.pth')
        torch.save({'model': fake_official_swin()}, path)
        torch.manual_seed(0)
        m = DepthFormerSwin(pretrain_img_size=224, patch_size=4, window_size=7, mlp_ratio=4, strides=(4, 2, 2, 2),
                            out_indices=(0, 1, 2, 3), qkv_bias=True, qk_scale=None, patch_norm=True, drop_rate=0.,
                            attn_drop_rate=0., drop_path_rate=0.0, use_abs_pos_embed=False, act_cfg=dict(type='GELU'),
                            norm_cfg=dict(type='LN', requires_grad=True), pretrain_style='official', pretrained=path,
                            conv_norm_cfg=dict(type='BN', requires_grad=True), depth=50, num_stages=0, USEPE=True, **ARCH)
        m.init_weights()
    sd = m.state_dict()
    keep = [k for k in sd if k.startswith(('patch_embed.', 'stages.0.blocks.1.', 'stages.0.downsample.', 'norm3.'))
            or k.endswith('stages.2.blocks.0.attn.w_msa.relative_position_bias_table')]
    arrays = {'key::' + k: sd[k].detach().cpu().numpy() for k in keep}
    arrays['all_keys'] = np.array(sorted(sd.keys()))
    np.savez_compressed(os.path.join(HERE, 'swin_official_ckpt.npz'), **arrays)
    print(f'wrote swin_official_ckpt.npz with {len(keep)} tensors; keys total {len(sd)}')


if __name__ == '__main__':
    main()
