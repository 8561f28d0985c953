"""DepthFormerSwin backbone on the fused HIP window-attention op.

Interface mirror of depth/models/backbones/depthformer_swin.py (class names, constructor kwargs and
state-dict keys: ``patch_embed.*``, ``stages.{s}.blocks.{b}.{norm1,attn.w_msa.*,norm2,ffn.layers.*}``,
``stages.{s}.downsample.{norm,reduction}``, ``norm{i}``, ``conv1``, ``bn1``), re-designed for MI355X:
tokens stay in one un-padded (B, H*W, C) layout for the whole stage, and pad / roll / window
partition / mask / softmax / reverse (reference :285-393, six full-tensor copies per block plus a
~5x materialised attention tensor) are folded into one kernel's addressing
(gedepth_amd/csrc/window_attn*.hip).
"""
import warnings
from copy import deepcopy

import torch
import torch.nn as nn
import torch.nn.functional as F

from ....kernels import window_attention
from ....mmrt.bricks import (FFN, BaseModule, Linear, ModuleList, build_conv_layer, build_dropout, build_norm_layer,
                             constant_init, trunc_normal_init)
from ...ops import resize
from ..builder import BACKBONES
from ..utils import PatchEmbedSwin, swin_convert


class PatchMerging(BaseModule):
    """2x2 patch merge: channel-major ``nn.Unfold`` ordering (reference :98-122), LN(4C), Linear(4C->2C)."""

    def __init__(self, in_channels, out_channels, stride=2, bias=False, norm_cfg=dict(type='LN'), init_cfg=None):
        super().__init__(init_cfg)
        assert stride == 2
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride
        sample_dim = stride ** 2 * in_channels
        self.norm = build_norm_layer(norm_cfg, sample_dim)[1] if norm_cfg is not None else None
        self.reduction = Linear(sample_dim, out_channels, bias=bias)

    def forward(self, x, hw_shape):
        B, L, C = x.shape
        H, W = hw_shape
        assert L == H * W, 'input feature has wrong size'
        x = x.view(B, H, W, C)
        if H % 2 or W % 2:
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        Ho, Wo = (H + 1) // 2, (W + 1) // 2
        # unfold(2,2) of the NCHW map orders features as (c, kh, kw): one gather instead of permute+unfold
        x = x.view(B, Ho, 2, Wo, 2, C).permute(0, 1, 3, 5, 2, 4).reshape(B, Ho * Wo, 4 * C)
        if self.norm is not None:
            x = self.norm(x)
        return self.reduction(x), (Ho, Wo)


class WindowMSA(BaseModule):
    """Parameter holder of the window attention (keys ``relative_position_bias_table``,
    ``relative_position_index``, ``qkv``, ``proj``); reference :128-230."""

    def __init__(self, embed_dims, num_heads, window_size, qkv_bias=True, qk_scale=None, attn_drop_rate=0.,
                 proj_drop_rate=0., init_cfg=None):
        super().__init__()
        assert tuple(window_size) == (7, 7), 'the HIP kernel is specialised for 7x7 windows'
        assert embed_dims // num_heads == 32, 'the HIP kernel is specialised for head_dim 32'
        assert attn_drop_rate == 0., 'attention dropout is 0 in every GEDepth config'
        self.embed_dims, self.window_size, self.num_heads = embed_dims, tuple(window_size), num_heads
        self.scale = qk_scale or (embed_dims // num_heads) ** -0.5
        self.init_cfg = init_cfg
        Wh, Ww = self.window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * Wh - 1) * (2 * Ww - 1), num_heads))
        i = torch.arange(Wh * Ww) // Ww
        j = torch.arange(Wh * Ww) % Ww
        index = (i[:, None] - i[None, :] + Wh - 1) * (2 * Ww - 1) + (j[:, None] - j[None, :] + Ww - 1)
        self.register_buffer('relative_position_index', index.contiguous())
        self.qkv = Linear(embed_dims, embed_dims * 3, bias=qkv_bias)
        self.proj = Linear(embed_dims, embed_dims)
        self.proj_drop = nn.Dropout(proj_drop_rate)

    def init_weights(self):
        trunc_normal_init(self.relative_position_bias_table, std=0.02)

    def forward(self, x, hw_shape, shift, variant=0):
        """x: (B, H*W, C) un-padded tokens (already LayerNorm-ed)."""
        H, W = hw_shape
        qkv = self.qkv(x)
        bias = self.qkv.bias if self.qkv.bias is not None else qkv.new_zeros(qkv.shape[-1], dtype=torch.float32)
        out = window_attention(qkv, bias, self.relative_position_bias_table, H, W, self.num_heads, shift,
                               self.scale, variant)
        return self.proj_drop(self.proj(out))


class ShiftWindowMSA(BaseModule):
    """Shifted-window MSA; sub-module name ``w_msa`` as in the reference (:233-393)."""

    def __init__(self, embed_dims, num_heads, window_size, shift_size=0, qkv_bias=True, qk_scale=None,
                 attn_drop_rate=0, proj_drop_rate=0, dropout_layer=dict(type='DropPath', drop_prob=0.), init_cfg=None):
        super().__init__(init_cfg)
        self.window_size, self.shift_size = window_size, shift_size
        assert 0 <= self.shift_size < self.window_size
        self.w_msa = WindowMSA(embed_dims, num_heads, (window_size, window_size), qkv_bias, qk_scale,
                               attn_drop_rate, proj_drop_rate)
        self.drop = build_dropout(dropout_layer)
        self.kernel_variant = 0

    def forward(self, query, hw_shape, identity=None):
        """``drop(attention(query))``; with ``identity`` the residual add is folded in: ``identity + drop(...)``."""
        B, L, C = query.shape
        assert L == hw_shape[0] * hw_shape[1], 'input feature has wrong size'
        out = self.w_msa(query, hw_shape, self.shift_size, self.kernel_variant)
        if identity is None:
            return self.drop(out)
        return self.drop.residual(identity, out) if hasattr(self.drop, 'residual') else identity + self.drop(out)


def _norm_with_skip(norm, x):
    """(norm(x), x for the skip connection): the fused pair when the norm layer offers it (mmrt.bricks.LayerNorm.forward_with_skip)."""
    f = getattr(norm, 'forward_with_skip', None)
    return f(x) if f is not None else (norm(x), x)


class SwinBlock(BaseModule):
    """x + DropPath(attn(LN x)); x + DropPath(FFN(LN x))  (reference :396-472)."""

    def __init__(self, embed_dims, num_heads, feedforward_channels, window_size=7, shift=False, qkv_bias=True,
                 qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0., act_cfg=dict(type='GELU'),
                 norm_cfg=dict(type='LN'), init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg
        self.norm1 = build_norm_layer(norm_cfg, embed_dims)[1]
        self.attn = ShiftWindowMSA(embed_dims, num_heads, window_size, window_size // 2 if shift else 0, qkv_bias,
                                   qk_scale, attn_drop_rate, drop_rate,
                                   dropout_layer=dict(type='DropPath', drop_prob=drop_path_rate))
        self.norm2 = build_norm_layer(norm_cfg, embed_dims)[1]
        self.ffn = FFN(embed_dims=embed_dims, feedforward_channels=feedforward_channels, num_fcs=2, ffn_drop=drop_rate,
                       dropout_layer=dict(type='DropPath', drop_prob=drop_path_rate), act_cfg=act_cfg,
                       add_identity=True, init_cfg=None)

    def forward(self, x, hw_shape):
        y, x = _norm_with_skip(self.norm1, x)
        x = self.attn(y, hw_shape, identity=x)
        y, x = _norm_with_skip(self.norm2, x)
        return self.ffn(y, identity=x)


class SwinBlockSequence(BaseModule):
    """One Swin stage (reference :475-551)."""

    def __init__(self, embed_dims, num_heads, feedforward_channels, depth, window_size=7, qkv_bias=True,
                 qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0., downsample=None,
                 act_cfg=dict(type='GELU'), norm_cfg=dict(type='LN'), init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg
        rates = drop_path_rate if isinstance(drop_path_rate, list) else [deepcopy(drop_path_rate)] * depth
        self.blocks = ModuleList([
            SwinBlock(embed_dims, num_heads, feedforward_channels, window_size, shift=bool(i % 2), qkv_bias=qkv_bias,
                      qk_scale=qk_scale, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate, drop_path_rate=rates[i],
                      act_cfg=act_cfg, norm_cfg=norm_cfg) for i in range(depth)])
        self.downsample = downsample

    def forward(self, x, hw_shape):
        for block in self.blocks:
            x = block(x, hw_shape)
        if self.downsample:
            x_down, down_hw = self.downsample(x, hw_shape)
            return x_down, down_hw, x, hw_shape
        return x, hw_shape, x, hw_shape


@BACKBONES.register_module()
class DepthFormerSwin(BaseModule):
    """Swin encoder + conv stem of DepthFormer (reference :753-1184) for the GEDepth configs
    (``num_stages=0``: no ResNet branch; ``USEPE=True``: 4-channel patch embedding RGB + ground depth)."""

    def __init__(self, pretrain_img_size=224, in_channels=3, embed_dims=96, patch_size=4, window_size=7, mlp_ratio=4,
                 depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), strides=(4, 2, 2, 2), out_indices=(0, 1, 2, 3),
                 qkv_bias=True, qk_scale=None, patch_norm=True, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1,
                 use_abs_pos_embed=False, act_cfg=dict(type='GELU'), norm_cfg=dict(type='LN'),
                 pretrain_style='official', pretrained=None, init_cfg=None, conv_cfg=None, conv_norm_cfg=None,
                 depth=None, num_stages=None, with_cp=False, conv_strides=(1, 2, 2, 2), conv_dilations=(1, 1, 1, 1),
                 style='pytorch', conv_pretrained=None, USEPE=False, USE_PARAM_PE=False):
        super().__init__()
        if with_cp:        # activation checkpointing would re-run DropPath with a fresh row of the one-draw bank (mmrt/bricks.py): refuse, do not mis-train
            raise NotImplementedError('with_cp=True (activation checkpointing) is not supported: 288 GB of HBM make it unnecessary on MI355X')
        if num_stages not in (0, None):
            raise NotImplementedError('the ResNet branch (num_stages>0) is outside the GEDepth hot path')
        if USE_PARAM_PE:
            raise NotImplementedError('USE_PARAM_PE is an unused experiment of the reference')
        if not (isinstance(pretrained, str) or pretrained is None):
            raise TypeError('pretrained must be a str or None')
        assert pretrain_style in ['official', 'mmcls']
        assert strides[0] == patch_size, 'Use non-overlapping patch embed.'
        self.conv_cfg, self.conv_norm_cfg, self.USEPE = conv_cfg, conv_norm_cfg, USEPE
        self.out_indices, self.use_abs_pos_embed = out_indices, use_abs_pos_embed
        self.pretrain_style, self.pretrained, self.init_cfg = pretrain_style, pretrained, init_cfg
        self.num_stages = 0
        if isinstance(pretrain_img_size, int):
            pretrain_img_size = (pretrain_img_size, pretrain_img_size)

        self.patch_embed = PatchEmbedSwin(in_channels=4 if USEPE else in_channels, embed_dims=embed_dims,
                                          conv_type='Conv2d', kernel_size=patch_size, stride=strides[0],
                                          pad_to_patch_size=True, norm_cfg=norm_cfg if patch_norm else None)
        if use_abs_pos_embed:
            n = (pretrain_img_size[0] // patch_size) * (pretrain_img_size[1] // patch_size)
            self.absolute_pos_embed = nn.Parameter(torch.zeros((1, n, embed_dims)))
        self.drop_after_pos = nn.Dropout(p=drop_rate)

        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.stages = ModuleList()
        ch = embed_dims
        for i, d in enumerate(depths):
            down = None
            if i < len(depths) - 1:
                down = PatchMerging(ch, 2 * ch, stride=strides[i + 1], norm_cfg=norm_cfg if patch_norm else None)
            self.stages.append(SwinBlockSequence(ch, num_heads[i], mlp_ratio * ch, d, window_size, qkv_bias, qk_scale,
                                                 drop_rate, attn_drop_rate, dpr[:d], down, act_cfg, norm_cfg))
            dpr = dpr[d:]
            if down:
                ch = down.out_channels
        self.num_features = [int(embed_dims * 2 ** i) for i in range(len(depths))]
        for i in out_indices:
            self.add_module(f'norm{i}', build_norm_layer(norm_cfg, self.num_features[i])[1])

        # conv stem on RGB (reference :1031-1043): conv 7x7/2 -> BN -> ReLU, named conv1 / bn1
        self.conv1 = build_conv_layer(conv_cfg, 3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self._stem_norm_name, stem_norm = build_norm_layer(conv_norm_cfg, 64, postfix=1)
        self.add_module(self._stem_norm_name, stem_norm)

    # ------------------------------------------------------------------ weights
    def init_weights(self):
        if self.pretrained is None:
            super().init_weights()
            if self.use_abs_pos_embed:
                trunc_normal_init(self.absolute_pos_embed, std=0.02)
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    trunc_normal_init(m.weight, std=.02)
                    if m.bias is not None:
                        constant_init(m.bias, 0)
                elif isinstance(m, nn.LayerNorm):
                    constant_init(m.bias, 0)
                    constant_init(m.weight, 1.0)
            return
        self.load_pretrained(self.pretrained)

    def load_pretrained(self, path):
        """Official Swin checkpoint -> this module (reference :1059-1125): key/weight reorder, bicubic
        resize of mismatched bias tables, zero-padded 4th input channel of the patch embedding."""
        ckpt = torch.load(path, map_location='cpu', weights_only=False)
        state = ckpt.get('state_dict', ckpt.get('model', ckpt)) if isinstance(ckpt, dict) else ckpt
        if self.pretrain_style == 'official':
            state = swin_convert(state)
        if next(iter(state)).startswith('module.'):
            state = {k[7:]: v for k, v in state.items()}
        own = self.state_dict()
        for k in [k for k in state if 'relative_position_bias_table' in k]:
            if k not in own:
                continue
            src, dst = state[k], own[k]
            if src.shape[1] != dst.shape[1]:
                warnings.warn(f'Error in loading {k}, pass')
            elif src.shape[0] != dst.shape[0]:
                s1, s2 = int(src.shape[0] ** 0.5), int(dst.shape[0] ** 0.5)
                t = resize(src.permute(1, 0).reshape(1, -1, s1, s1), size=(s2, s2), mode='bicubic')
                state[k] = t.view(dst.shape[1], dst.shape[0]).permute(1, 0).contiguous()
        if self.USEPE:
            for k, v in list(state.items()):
                if k in own and own[k].shape != v.shape and own[k].dim() == 4:
                    padded = torch.zeros(own[k].shape)
                    padded[:, :padded.shape[1] - 1] = v
                    state[k] = padded
        self.load_state_dict(state, strict=False)

    # ------------------------------------------------------------------ forward
    channels_last = False       # set by depth.models.utils.to_channels_last: maps leave / enter the backbone as (B, H, W, C) storage

    def conv_stem(self, x):
        bn = getattr(self, self._stem_norm_name)
        if self.channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        y = self.conv1(x)
        if (y.is_cuda and type(bn) is nn.BatchNorm2d and bn.training and bn.affine and bn.track_running_stats
                and bn.momentum is not None):
            from ....kernels import bn_act
            return bn_act(y, bn, 0.0)                         # training-mode BN + ReLU in two streaming passes
        return F.relu(bn(y), inplace=True)

    def forward(self, x_ori):
        outs = [self.conv_stem(x_ori[:, 0:3] if self.USEPE else x_ori)]
        x4 = x_ori[:, 0:4] if self.USEPE else x_ori
        if self.channels_last:                        # the k4 s4 projection then emits (B, H/4, W/4, C): tokens without a transpose
            x4 = x4.contiguous(memory_format=torch.channels_last)
        x, hw_shape = self.patch_embed(x4)
        if self.use_abs_pos_embed:
            x = x + self.absolute_pos_embed
        x = self.drop_after_pos(x)
        for i, stage in enumerate(self.stages):
            x, hw_shape, out, out_hw = stage(x, hw_shape)
            if i in self.out_indices:
                out = getattr(self, f'norm{i}')(out)
                out = out.view(-1, *out_hw, self.num_features[i]).permute(0, 3, 1, 2)     # channels-last storage of a (B,C,H,W) map
                outs.append(out if self.channels_last else out.contiguous())
        return outs
