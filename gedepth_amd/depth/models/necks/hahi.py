"""HAHI neck (hierarchical aggregation / heterogeneous interaction) on the HIP deformable-attention op.

Interface mirror of depth/models/necks/hahi.py:82-356 (constructor kwargs, state-dict keys
``lateral_convs / trans_proj / trans_fusion / conv_proj / conv_fusion / reference_points / level_embed /
multi_att / self_attn``) and of the mmcv 1.3.13 ``MultiScaleDeformableAttention`` module it instantiates
(keys ``sampling_offsets / attention_weights / value_proj / output_proj``).

MI355X-first differences: sine position embeddings, self-attention reference points and the
cross-attention reference points are input-independent for a fixed feature shape (all masks are all-false,
``valid_ratios == 1``; the cross-attention points are ``sigmoid(Linear(pos_embed))``, hahi.py:299-300), so
the embeddings / pixel-centre points are cached per shape, and the sampling core is one gather kernel
(gedepth_amd/csrc/msda.hip) instead of mmcv's CUDA extension.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import kernels as K
from ....kernels import add_rows, concat_tokens_map, ms_deform_attn_raw, residual_dropout, tokens_from_map
from ....mmrt import bricks
from ....mmrt.bricks import BaseModule, ConvModule, build_positional_encoding, xavier_init
from ..builder import ATTENTION, NECKS


@ATTENTION.register_module()
class MultiScaleDeformableAttention(BaseModule):
    """mmcv-compatible module; forward supports the 2-d reference-point form used by HAHI."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64, dropout=0.1,
                 batch_first=False, norm_cfg=None, init_cfg=None):
        super().__init__(init_cfg)
        if embed_dims % num_heads != 0:
            raise ValueError(f'embed_dims must be divisible by num_heads, but got {embed_dims} and {num_heads}')
        if embed_dims // num_heads != 64:
            raise NotImplementedError('the HIP sampling kernel is specialised for 64 channels per head')
        self.norm_cfg, self.batch_first = norm_cfg, batch_first
        self.dropout = nn.Dropout(dropout)
        self.im2col_step = im2col_step
        self.embed_dims, self.num_levels, self.num_heads, self.num_points = embed_dims, num_levels, num_heads, num_points
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = bricks.Linear(embed_dims, embed_dims)
        self.output_proj = bricks.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        bricks.constant_init(self.sampling_offsets, 0.)
        self.sampling_offsets.bias.data = bricks.msda_offset_bias(self.num_heads, self.num_levels, self.num_points)
        bricks.constant_init(self.attention_weights, val=0., bias=0.)
        xavier_init(self.value_proj, distribution='uniform', bias=0.)
        xavier_init(self.output_proj, distribution='uniform', bias=0.)
        self._is_init = True

    def _attend(self, query, value, reference_points, spatial_shapes, key_padding_mask=None, query_shapes=None, query_order=None):
        """query (B,Nq,C) with the positional embedding already added, value (B,Nv,C) -> output_proj(sampled) (B,Nq,C).

        The two query linears run as ONE GEMM on concatenated weights (their input is the same 0.4-0.8 GB tensor), and
        view / softmax / normaliser / reference-point arithmetic is folded into the sampling kernel (kernels.ms_deform_attn_raw)."""
        bs, num_query, _ = query.shape
        num_value = value.shape[1]
        if torch.is_tensor(spatial_shapes):
            spatial_shapes = [(int(h), int(w)) for h, w in spatial_shapes.tolist()]
        assert sum(h * w for h, w in spatial_shapes) == num_value
        if reference_points.shape[-1] != 2:
            raise ValueError('only 2-d reference points are used on the GEDepth path')
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.view(bs, num_value, self.num_heads, -1)
        nH, L, P = self.num_heads, self.num_levels, self.num_points
        w = torch.cat((self.sampling_offsets.weight, self.attention_weights.weight), 0)
        b = torch.cat((self.sampling_offsets.bias, self.attention_weights.bias), 0)
        raw = bricks.linear_tokens(query, w, b)               # (B, Nq, nH*L*P*2 + nH*L*P)
        # view / softmax / normaliser / reference-point arithmetic happen inside the sampling kernel (fp32 locations and weights:
        # pixel coordinates up to ~1000 need > 8 mantissa bits); its backward returns the gradient of `raw` directly
        if raw.dtype != value.dtype:
            raw = raw.to(value.dtype)
        ref = reference_points.expand(bs, num_query, L, 2)
        if query_order is not None and 'msda_mm' not in K.DISABLED and K.msda_mm_supported(value, raw, spatial_shapes, nH, L, P):
            # MFMA decomposition on query tiles in `query_order` (csrc/msda_mm.hip); any order gives the same result
            out = K.ms_deform_attn_mm(value, raw, ref, spatial_shapes, query_order, nH, L, P)
        elif K.msda_self_split_ok(value, raw, spatial_shapes, query_shapes, nH, L, P):
            # self-attention (queries = the value levels): level-0 queries on the MFMA kernels, coarse-level queries on the window kernels
            out = K.ms_deform_attn_self_split(value, raw, ref, spatial_shapes, nH, L, P)
        else:
            out = ms_deform_attn_raw(value, raw, ref, spatial_shapes, query_shapes, nH, L, P)
        return self.output_proj(out)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, query_shapes=None, query_order=None, **kwargs):
        """mmcv MultiScaleDeformableAttention.forward; ``query_shapes`` (extension, optional): the queries as (H, W) maps in
        raster order, e.g. ``spatial_shapes`` itself for self-attention — enables the 2-D tiled sampling kernels."""
        if value is None:
            value = query
        if identity is None:
            identity = query                      # BEFORE the positional embedding is added
        if query_pos is not None:
            if query_pos.dim() == 3 and query_pos.shape[0] == 1 and query_pos.dtype == torch.float32 and query.dim() == 3 and self.batch_first:
                query = add_rows(query, query_pos[0])        # one pass; no (B, N, C) copy of the broadcast embedding
            else:
                query = query + query_pos.to(query.dtype)
        if not self.batch_first:
            query, value = query.permute(1, 0, 2), value.permute(1, 0, 2)
        out = self._attend(query, value, reference_points, spatial_shapes, key_padding_mask, query_shapes, query_order)
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        if self.training and self.dropout.p > 0 and out.is_cuda and out.dim() == 3 and self.batch_first:
            return residual_dropout(identity, out, self.dropout.p)
        return self.dropout(out) + identity

    def forward_map(self, fmap, pos_map, value, reference_points, spatial_shapes, concat_with, query_order=None, query=None):
        """Cross-attention with a feature-map query (hahi.py:303-333): query = tokens(fmap) + pos, identity = fmap, and the
        result returned as ``torch.cat([to_map(dropout(out) + identity), concat_with], 1)``.  The two layout changes carry
        the position add, the dropout, the residual and the concat write (gedepth_amd/csrc/neck.hip).  ``query``: tokens(fmap) + pos
        when the producer of ``fmap`` wrote it in the same pass (kernels.conv1x1_bn_act_pos)."""
        if query is None:
            query = tokens_from_map(fmap, pos_map)
        out = self._attend(query, value, reference_points, spatial_shapes, query_shapes=[tuple(fmap.shape[2:])], query_order=query_order)
        p = self.dropout.p if self.training else 0.0
        return concat_tokens_map(out, concat_with, identity=fmap, tokens_first=True, p_drop=p)


@NECKS.register_module()
class HAHIHeteroNeck(BaseModule):

    def __init__(self, in_channels, out_channels, embedding_dim, scales=[1, 1, 1, 1],
                 norm_cfg=dict(type='BN', requires_grad=True), act_cfg=dict(type='ReLU', inplace=True),
                 cross_att=True, self_att=True, constrain=False, positional_encoding=None, num_points=8):
        super().__init__()
        assert isinstance(in_channels, list)
        assert all(s == 1 for s in scales), 'GEDepth configs use scale 1 at every level'
        self.cross_att, self.self_att, self.constrain = cross_att, self_att, constrain
        self.in_channels, self.out_channels, self.scales = in_channels, out_channels, scales
        self.num_outs = len(scales)
        self.embedding_dim = E = embedding_dim
        cm = dict(norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.lateral_convs = nn.ModuleList(ConvModule(i, o, kernel_size=1, **cm) for i, o in zip(in_channels, out_channels))
        self.trans_proj = nn.ModuleList(ConvModule(o, E, kernel_size=1, **cm) for o in out_channels[1:])
        self.trans_fusion = nn.ModuleList(ConvModule(o + E, o, kernel_size=3, padding=1, stride=1, **cm)
                                          for o in out_channels[1:])
        self.conv_proj = nn.Sequential(ConvModule(in_channels[0], E, kernel_size=1, **cm))
        self.conv_fusion = nn.Sequential(ConvModule(in_channels[0] + E, out_channels[0], kernel_size=3, padding=1,
                                                    stride=1, **cm))
        self.trans_positional_encoding = build_positional_encoding(positional_encoding)
        self.conv_positional_encoding = build_positional_encoding(positional_encoding)
        self.reference_points = nn.Linear(E, 2)
        self.level_embed = nn.Parameter(torch.Tensor(4, E))
        att = dict(embed_dims=E, num_levels=4, num_heads=8, num_points=num_points, batch_first=True)
        self.multi_att = MultiScaleDeformableAttention(**att)
        self.self_attn = MultiScaleDeformableAttention(**att)
        self._ref_cache = {}
        self._order_cache = {}          # cross-attention query order (a speed heuristic): refreshed every ORDER_REFRESH forwards

    ORDER_REFRESH = 64

    def _cross_order(self, ref_xy, level0_hw):
        """Queries sorted by the level-0 cell of their (content-independent) reference point, so that 32 consecutive queries sample
        one compact window (csrc/msda_mm.hip).  The result of the attention does not depend on the order; the reference points move
        slowly under training, so the sort (a dozen small launches) is redone only every ORDER_REFRESH forwards."""
        key = (tuple(ref_xy.shape), tuple(level0_hw), str(ref_xy.device))
        ent = self._order_cache.get(key)
        # one entry per shape (train / validation / multi-scale shapes alternate without re-sorting); the reference points only move
        # under training, so evaluation never refreshes, and a step being captured in a hipGraph keeps the order it has
        stale = ent is not None and self.training and ent[1] >= self.ORDER_REFRESH and not torch.cuda.is_current_stream_capturing()
        if ent is None or stale:
            if len(self._order_cache) >= 8:
                self._order_cache.clear()
            ent = self._order_cache[key] = [K.msda_ref_order(ref_xy.detach(), level0_hw), 0]
        if self.training:
            ent[1] += 1
        return ent[0]

    def init_weights(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        nn.init.xavier_uniform_(self.reference_points.weight.data, gain=1.0)
        nn.init.constant_(self.reference_points.bias.data, 0.)
        nn.init.normal_(self.level_embed)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                xavier_init(m, distribution='uniform')
            if isinstance(m, MultiScaleDeformableAttention):
                m.init_weights()
        self._is_init = True

    def _pixel_centres(self, shapes, device):
        """Self-attention reference points: normalised pixel centres per level (hahi.py:220-233, ratios == 1)."""
        key = (tuple(shapes), str(device))
        if key not in self._ref_cache:
            pts = []
            for h, w in shapes:
                ry = torch.linspace(0.5, h - 0.5, h, dtype=torch.float32, device=device) / h
                rx = torch.linspace(0.5, w - 0.5, w, dtype=torch.float32, device=device) / w
                gy, gx = torch.meshgrid(ry, rx, indexing='ij')
                pts.append(torch.stack((gx.reshape(-1), gy.reshape(-1)), -1))
            ref = torch.cat(pts, 0)[None, :, None, :].expand(1, -1, len(shapes), 2).contiguous()
            self._ref_cache[key] = ref
        return self._ref_cache[key]

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        feats = [conv(x) for conv, x in zip(self.lateral_convs, inputs)]
        feat_conv, feats_trans = feats[0], feats[1:]
        bs = feat_conv.shape[0]
        dev = feat_conv.device
        shapes, srcs, poss = [], [], []
        for i, ft in enumerate(feats_trans):
            h, w = ft.shape[2:]
            shapes.append((h, w))
            pos = self.trans_positional_encoding.grid(h, w, dev).flatten(2).transpose(1, 2)
            poss.append(pos + self.level_embed[i].view(1, 1, -1))
            srcs.append(self.trans_proj[i](ft).flatten(2).transpose(1, 2))
        src_flatten = torch.cat(srcs, 1)
        pos_flatten = torch.cat(poss, 1)                      # (1, N, C) fp32: broadcast over the batch inside the add
        if self.self_att:
            ref = self._pixel_centres(shapes, dev).expand(bs, -1, -1, -1)
            # The MFMA decomposition (csrc/msda_mm.hip) on 4 x 8 patches of the token maps is available for the self-attention too
            # (GE_MSDA_MM_SELF=1) but measured SLOWER than the LDS-window gather kernels here: the offset bias spreads the 8 points of a
            # head over +-8 cells, so a 32-query patch reaches ~20 x 24 rows per level = 8 chunks of the 64-row coefficient image
            # (forward 2.49 vs 1.63 ms, d_raw 2.48 vs 1.66 ms, step 53.0 vs 50.8 ms same-session, round 4)
            order = (K.msda_tile_order(shapes, dev) if (os.environ.get('GE_MSDA_MM_SELF') == '1' and src_flatten.is_cuda
                                                        and src_flatten.dtype == torch.bfloat16) else None)
            src = self.self_attn(src_flatten, value=None, identity=None, query_pos=pos_flatten,
                                 reference_points=ref, spatial_shapes=shapes, query_shapes=shapes, query_order=order)
        else:
            src = src_flatten

        h, w = feat_conv.shape[2:]
        pos_map = self.conv_positional_encoding.grid(h, w, dev) if self.cross_att else None      # (1, C, h, w) fp32, cached
        query = None
        if self.training and len(self.conv_proj) == 1 and K.conv1x1_bn_act_pos_ok(self.conv_proj[0], feat_conv):
            # 64 -> E channels at the finest level: convolution, BatchNorm (statistics from the input's Gram matrix), ReLU and the query's
            # position add in ONE pass over the 8x wider output (csrc/conv1x1_bn.hip); the pre-BN map is never stored
            conv_skip, query = K.conv1x1_bn_act_pos(self.conv_proj[0], feat_conv, pos_map)
        else:
            conv_skip = self.conv_proj(feat_conv)
        if self.cross_att:
            # content-independent reference points: one fp32 (1, Nq, 2) evaluation, broadcast over batch and levels
            with torch.autocast('cuda', enabled=False):
                # Linear(512 -> 2) over 1e5 positions: as a GEMM with N = 2 the libraries reach ~1 TFLOP/s (0.3 ms, and
                # twice that backward); two matrix-vector products on the (C, H*W) map stream it at HBM speed instead
                pm = pos_map.flatten(2)[0]                                               # (C, H*W)
                w, b = self.reference_points.weight, self.reference_points.bias
                ref = torch.stack((torch.mv(pm.t(), w[0]), torch.mv(pm.t(), w[1])), -1).add(b).sigmoid()[None]
            order = self._cross_order(ref[0], shapes[0]) if (conv_skip.is_cuda and src.dtype == torch.bfloat16) else None
            ref = ref[:, :, None, :].expand(bs, -1, len(shapes), 2)
            fused = self.multi_att.forward_map(conv_skip, pos_map, src, ref, shapes, concat_with=feat_conv, query_order=order, query=query)
        else:
            fused = torch.cat([conv_skip, feat_conv], dim=1)
        # one split (backward: ONE concatenating write of the level gradients) instead of per-level slices, whose backward
        # zero-fills a full (B, sum HW, C) tensor per level and adds the three of them
        outs = [self.conv_fusion(fused)]
        for i, (ft, piece) in enumerate(zip(feats_trans, torch.split(src, [h * w for h, w in shapes], dim=1))):
            outs.append(self.trans_fusion[i](concat_tokens_map(piece, ft, tokens_first=False)))
        return tuple(outs)

