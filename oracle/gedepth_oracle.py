"""CPU oracle for the GEDepth hot path  —  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch (CPU, fp32; numpy float64 for the offline ground-plane maths)
functional restatement of the reference's algorithm for the training hot path
of SURVEY.md §8.  It operates directly on a state dict that uses the
reference's parameter names, so the product model's ``state_dict()`` can be fed
to it unchanged.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this file; nothing under
``gedepth_amd/`` does.

Pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md §4), so this restatement is pinned against fixtures generated in
the build container by importing the reference itself through an mmcv stand-in
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; checked by
``tests/test_oracle_golden.py``) and against the known answers of SURVEY.md
Appendix E.  The arithmetic that lives in mmcv-full 1.3.13 (not vendored in
the reference: ConvModule, FFN, MultiScaleDeformableAttention) is restated
from that release's published behaviour; its deformable-attention core was
cross-checked bit-for-bit against the independent implementation in HF
transformers (``tests/test_oracle_golden.py::test_msda_core_vs_transformers``).

Every function cites the reference file:line it follows (paths relative to the
reference root).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

WINDOW = 7


# =============================================================================== backbone
def layer_norm(x, P, prefix, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), P[prefix + '.weight'], P[prefix + '.bias'], eps)


def relative_position_index(ws=WINDOW):
    """depth/models/backbones/depthformer_swin.py:166-172,226-230 (double_step_seq)."""
    seq1 = torch.arange(0, (2 * ws - 1) * ws, 2 * ws - 1)
    seq2 = torch.arange(0, ws, 1)
    coords = (seq1[:, None] + seq2[None, :]).reshape(1, -1)
    idx = coords + coords.T
    return idx.flip(1).contiguous()


def window_msa(x, mask, P, prefix, num_heads):
    """WindowMSA.forward, depthformer_swin.py:184-224.  x: (nW*B, 49, C)."""
    Bw, N, C = x.shape
    hd = C // num_heads
    scale = hd ** -0.5
    qkv = F.linear(x, P[prefix + '.qkv.weight'], P[prefix + '.qkv.bias'])
    qkv = qkv.reshape(Bw, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * scale
    attn = q @ k.transpose(-2, -1)
    table = P[prefix + '.relative_position_bias_table']
    index = P.get(prefix + '.relative_position_index', relative_position_index())
    bias = table[index.view(-1)].view(N, N, -1).permute(2, 0, 1).contiguous()
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = attn.view(Bw // nW, nW, num_heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, num_heads, N, N)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(Bw, N, C)
    return F.linear(x, P[prefix + '.proj.weight'], P[prefix + '.proj.bias'])


def window_partition(x, ws=WINDOW):
    """depthformer_swin.py:379-393."""
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def window_reverse(windows, H, W, ws=WINDOW):
    """depthformer_swin.py:362-377."""
    B = int(windows.shape[0] / (H * W / ws / ws))
    x = windows.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def shift_mask(Hp, Wp, ws=WINDOW, shift=WINDOW // 2):
    """The attention mask of depthformer_swin.py:305-326 (built on the padded, rolled grid)."""
    img_mask = torch.zeros((1, Hp, Wp, 1))
    cnt = 0
    for h in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for w in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img_mask[:, h, w, :] = cnt
            cnt += 1
    mw = window_partition(img_mask, ws).view(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def shift_window_msa(x, hw, P, prefix, num_heads, shift):
    """ShiftWindowMSA.forward, depthformer_swin.py:285-360 (DropPath handled by caller: p=0)."""
    B, L, C = x.shape
    H, W = hw
    ws = WINDOW
    x = x.view(B, H, W, C)
    pad_r = (ws - W % ws) % ws
    pad_b = (ws - H % ws) % ws
    x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
    Hp, Wp = x.shape[1], x.shape[2]
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
        mask = shift_mask(Hp, Wp, ws, shift).to(x.dtype)
    else:
        mask = None
    xw = window_partition(x, ws).view(-1, ws * ws, C)
    aw = window_msa(xw, mask, P, prefix + '.w_msa', num_heads)
    x = window_reverse(aw.view(-1, ws, ws, C), Hp, Wp, ws)
    if shift > 0:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    if pad_r > 0 or pad_b:
        x = x[:, :H, :W, :].contiguous()
    return x.view(B, H * W, C)


def swin_block(x, hw, P, prefix, num_heads, shift):
    """SwinBlock.forward, depthformer_swin.py:461-472; FFN per mmcv: exact-erf GELU."""
    identity = x
    x = layer_norm(x, P, prefix + '.norm1')
    x = shift_window_msa(x, hw, P, prefix + '.attn', num_heads, shift) + identity
    identity = x
    x = layer_norm(x, P, prefix + '.norm2')
    x = F.linear(x, P[prefix + '.ffn.layers.0.0.weight'], P[prefix + '.ffn.layers.0.0.bias'])
    x = F.gelu(x)
    x = F.linear(x, P[prefix + '.ffn.layers.1.weight'], P[prefix + '.ffn.layers.1.bias'])
    return identity + x


def patch_merging(x, hw, P, prefix):
    """PatchMerging.forward, depthformer_swin.py:98-122 (nn.Unfold(2,2): channel-major 4C order)."""
    B, L, C = x.shape
    H, W = hw
    x = x.view(B, H, W, C).permute(0, 3, 1, 2)
    if H % 2 or W % 2:
        x = F.pad(x, (0, W % 2, 0, H % 2))
    x = F.unfold(x, kernel_size=2, stride=2).transpose(1, 2)
    x = layer_norm(x, P, prefix + '.norm')
    x = F.linear(x, P[prefix + '.reduction.weight'])
    return x, ((H + 1) // 2, (W + 1) // 2)


def batch_norm(x, P, prefix, train_bn, eps=1e-5):
    return F.batch_norm(x, P[prefix + '.running_mean'].detach().clone(), P[prefix + '.running_var'].detach().clone(),
                        P[prefix + '.weight'], P[prefix + '.bias'], training=train_bn, momentum=0.1, eps=eps)


def backbone(img, P, cfg, train_bn=False, prefix='backbone'):
    """DepthFormerSwin.forward, depthformer_swin.py:1149-1184 (USEPE=True, num_stages=0)."""
    outs = []
    x3 = img[:, 0:3]
    stem = F.conv2d(x3, P[prefix + '.conv1.weight'], None, stride=2, padding=3)
    stem = relu(batch_norm(stem, P, prefix + '.bn1', train_bn), site=prefix + '.bn1')
    outs.append(stem)
    x = img[:, 0:4]
    # PatchEmbedSwin.forward, models/utils/embed.py:282-302
    Hh, Ww = x.shape[2], x.shape[3]
    if Hh % 4:
        x = F.pad(x, (0, 0, 0, 4 - Hh % 4))
    if Ww % 4:
        x = F.pad(x, (0, 4 - Ww % 4, 0, 0))
    x = F.conv2d(x, P[prefix + '.patch_embed.projection.weight'], P[prefix + '.patch_embed.projection.bias'],
                 stride=4)
    hw = (x.shape[2], x.shape[3])
    x = x.flatten(2).transpose(1, 2)
    x = layer_norm(x, P, prefix + '.patch_embed.norm')
    C = cfg['embed_dims']
    for s, (depth, nh) in enumerate(zip(cfg['depths'], cfg['num_heads'])):
        for b in range(depth):
            x = swin_block(x, hw, P, f'{prefix}.stages.{s}.blocks.{b}', nh, 0 if b % 2 == 0 else WINDOW // 2)
        out, out_hw = x, hw
        if s < len(cfg['depths']) - 1:
            x, hw = patch_merging(x, hw, P, f'{prefix}.stages.{s}.downsample')
        out = layer_norm(out, P, f'{prefix}.norm{s}')
        outs.append(out.view(-1, *out_hw, C * 2 ** s).permute(0, 3, 1, 2).contiguous())
    return outs


# =================================================================================== neck
def sine_positional_encoding(B, H, W, num_feats=256, temperature=10000, dtype=torch.float32):
    """SinePositionalEncoding.forward on an all-false mask, depth/utils/position_encoding.py:54-89.  Always evaluated
    in fp32 like the reference (an input-independent constant); ``dtype`` only casts the result (float64 oracle mode)."""
    not_mask = torch.ones(B, H, W, dtype=torch.int)
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
    dim_t = torch.arange(num_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).view(B, H, W, -1)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).view(B, H, W, -1)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2).to(dtype)


# Bilinear sampling is continuous in the sampling location but its DERIVATIVE is not: at an integer pixel coordinate the
# left and right slopes differ.  A location that lands within rounding of such a kink gets one slope or the other depending
# on the last bits of the GEMM that produced its offset — in any fp32 implementation, the reference included — and that
# one choice moves the gradient of everything upstream by ~1e-3 (tests/f64ref.py measures it).  KINK_NUDGE = (tau_px, sign)
# resolves every location within tau_px pixels of a kink to one side; the spread between sign = +1 and -1 in float64 is
# the part of the gradient that the algorithm itself leaves undefined.  Test instrumentation only; None = off.
# The same holds for ReLU / LeakyReLU: an element whose pre-activation is within rounding of zero passes its gradient or
# not depending on the last bits (one flipped element of the 2 x 1536 x 2 x 3 top-level map moves every stage-3 gradient of
# the 64 x 96 fixture by 4e-3).  KINK_NUDGE = (tau_px, tau_act, sign) also pushes pre-activations within tau_act of zero to
# one side.
KINK_NUDGE = None
KINK_COUNT = [0, 0, 0, 0]    # [ambiguous sampling coordinates, coordinates seen, ambiguous activations, activations seen]


# KINK_FORCE = dict(act={site: bool mask}, floors=[int tensor per msda_core call, in call order]) makes this oracle take
# the SAME side as another implementation at every kink: activations use the given pass / block mask instead of the sign of
# their own pre-activation, sampling locations are moved (by less than FORCE_MAX_SHIFT pixels, asserted) into the given
# integer cell.  Two implementations that agree to rounding then agree on gradients to rounding as well — which is what
# tests/test_model_gpu.py checks for the HIP path.  Test instrumentation only; None = off.
KINK_FORCE = None
FORCE_MAX_SHIFT = 1e-3
FORCE_STATS = dict(act_flipped=0, act_seen=0, floor_flipped=0, floor_seen=0, max_shift_px=0.0, max_flipped_preact=0.0)


KINK_RECORD = None            # dict(act={}, floors=[]): filled with THIS evaluation's decisions (same layout as KINK_FORCE)


def _forced_act(x, site, slope):
    """relu / leaky_relu with the pass mask of KINK_FORCE['act'][site] (None: not forced)."""
    if KINK_RECORD is not None and site is not None:
        KINK_RECORD['act'][site] = x.detach() > 0
    if KINK_FORCE is None or site is None or site not in KINK_FORCE.get('act', {}):
        return None
    mask = KINK_FORCE['act'][site].to(torch.bool)
    assert mask.shape == x.shape, (site, mask.shape, x.shape)
    own = x.detach() > 0
    flipped = own != mask
    FORCE_STATS['act_flipped'] += int(flipped.sum())
    FORCE_STATS['act_seen'] += flipped.numel()
    if flipped.any():
        FORCE_STATS['max_flipped_preact'] = max(FORCE_STATS['max_flipped_preact'], float(x.detach().abs()[flipped].max()))
    return x * torch.where(mask, torch.ones((), dtype=x.dtype), torch.full((), float(slope), dtype=x.dtype))


def _act_nudge(x):
    if KINK_NUDGE is None:
        return x
    tau, sign = KINK_NUDGE[1], KINK_NUDGE[2]
    near = x.detach().abs() < tau
    KINK_COUNT[2] += int(near.sum())
    KINK_COUNT[3] += near.numel()
    return x + near.to(x.dtype) * (sign * tau)


def relu(x, site=None):
    forced = _forced_act(x, site, 0.0)
    return forced if forced is not None else F.relu(_act_nudge(x))


def leaky_relu(x, slope=0.01, site=None):
    forced = _forced_act(x, site, slope)
    return forced if forced is not None else F.leaky_relu(_act_nudge(x), slope)


def _nudge_off_kinks(loc, spatial_shapes, tau_px, _tau_act, sign):
    wh = torch.tensor([[float(w), float(h)] for h, w in spatial_shapes], dtype=loc.dtype)     # (L, 2) = (W_l, H_l)
    wh = wh.view(1, 1, 1, len(spatial_shapes), 1, 2)
    px = loc.detach() * wh - 0.5
    near = (px - torch.round(px)).abs() < tau_px
    KINK_COUNT[0] += int(near.sum())
    KINK_COUNT[1] += near.numel()
    return loc + near.to(loc.dtype) * (sign * tau_px) / wh


def sampling_cells(loc, spatial_shapes):
    """Integer cell floor(loc * (W_l, H_l) - 0.5) of every sampling coordinate: the bilinear kink decision."""
    wh = torch.tensor([[float(w), float(h)] for h, w in spatial_shapes], dtype=loc.dtype, device=loc.device)
    return torch.floor(loc * wh.view(1, 1, 1, len(spatial_shapes), 1, 2) - 0.5).to(torch.int32)


def _force_floors(loc, spatial_shapes, cells):
    """Move every sampling coordinate whose integer cell differs from ``cells`` just inside that cell."""
    wh = torch.tensor([[float(w), float(h)] for h, w in spatial_shapes], dtype=loc.dtype).view(1, 1, 1, len(spatial_shapes), 1, 2)
    px = loc.detach() * wh - 0.5
    own = torch.floor(px)
    cells = cells.to(loc.dtype)
    assert cells.shape == px.shape, (cells.shape, px.shape)
    mism = own != cells
    FORCE_STATS['floor_flipped'] += int(mism.sum())
    FORCE_STATS['floor_seen'] += mism.numel()
    if not mism.any():
        return loc
    target = torch.where(cells < own, cells + (1.0 - 1e-12), cells)       # the near edge of the other cell
    shift = torch.where(mism, target - px, torch.zeros_like(px))
    FORCE_STATS['max_shift_px'] = max(FORCE_STATS['max_shift_px'], float(shift.abs().max()))
    assert float(shift.abs().max()) <= FORCE_MAX_SHIFT, f'sampling cell differs by {float(shift.abs().max())} px: not a rounding-level kink'
    return loc + shift / wh


def msda_core(value, spatial_shapes, sampling_locations, attention_weights):
    """mmcv 1.3.13 ``multi_scale_deformable_attn_pytorch`` (not vendored; reference call sites
    necks/hahi.py:16,279-289,316-325).  value (B,Nv,nH,d); sampling_locations (B,Nq,nH,L,P,2) in
    [0,1]; attention_weights (B,Nq,nH,L,P) -> (B,Nq,nH*d).  bilinear, zero padding,
    align_corners=False."""
    bs, _, num_heads, dims = value.shape
    _, num_queries, _, num_levels, num_points, _ = sampling_locations.shape
    value_list = value.split([int(h) * int(w) for h, w in spatial_shapes], dim=1)
    if KINK_NUDGE is not None:
        sampling_locations = _nudge_off_kinks(sampling_locations, spatial_shapes, *KINK_NUDGE)
    if KINK_RECORD is not None:
        KINK_RECORD['floors'].append(sampling_cells(sampling_locations.detach(), spatial_shapes))
    if KINK_FORCE is not None and KINK_FORCE.get('floors'):
        sampling_locations = _force_floors(sampling_locations, spatial_shapes, KINK_FORCE['floors'].pop(0))
    grids = 2 * sampling_locations - 1
    sampled = []
    for lvl, (h, w) in enumerate(spatial_shapes):
        h, w = int(h), int(w)
        v = value_list[lvl].flatten(2).transpose(1, 2).reshape(bs * num_heads, dims, h, w)
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode='bilinear', padding_mode='zeros', align_corners=False))
    aw = attention_weights.transpose(1, 2).reshape(bs * num_heads, 1, num_queries, num_levels * num_points)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * aw).sum(-1).view(bs, num_heads * dims, num_queries)
    return out.transpose(1, 2).contiguous()


def msda_module(query, value, query_pos, reference_points, spatial_shapes, P, prefix,
                num_heads=8, num_levels=4, num_points=8):
    """mmcv 1.3.13 MultiScaleDeformableAttention.forward with batch_first=True, dropout p=0
    (SURVEY Appendix A): identity = query *before* adding query_pos; value defaults to that query."""
    identity = query
    if value is None:
        value = query
    query = query + query_pos
    bs, nq, C = query.shape
    nv = value.shape[1]
    value = F.linear(value, P[prefix + '.value_proj.weight'], P[prefix + '.value_proj.bias'])
    value = value.view(bs, nv, num_heads, -1)
    off = F.linear(query, P[prefix + '.sampling_offsets.weight'], P[prefix + '.sampling_offsets.bias'])
    off = off.view(bs, nq, num_heads, num_levels, num_points, 2)
    aw = F.linear(query, P[prefix + '.attention_weights.weight'], P[prefix + '.attention_weights.bias'])
    aw = aw.view(bs, nq, num_heads, num_levels * num_points).softmax(-1)
    aw = aw.view(bs, nq, num_heads, num_levels, num_points)
    shapes = torch.as_tensor(spatial_shapes, dtype=torch.long)
    normalizer = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
    loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    out = msda_core(value, spatial_shapes, loc, aw)
    out = F.linear(out, P[prefix + '.output_proj.weight'], P[prefix + '.output_proj.bias'])
    return out + identity


def conv_module(x, P, prefix, train_bn, padding=0, norm=True, act='relu'):
    """mmcv ConvModule conv->bn->act: bias iff no norm."""
    x = F.conv2d(x, P[prefix + '.conv.weight'], P.get(prefix + '.conv.bias'), padding=padding)
    if norm:
        x = batch_norm(x, P, prefix + '.bn', train_bn)
    if act == 'relu':
        x = relu(x, site=prefix)
    elif act == 'leaky':
        x = leaky_relu(x, 0.01, site=prefix)
    return x


def hahi_reference_points(spatial_shapes, B, dtype=torch.float32):
    """HAHIHeteroNeck.get_reference_points with valid_ratios == 1, necks/hahi.py:220-233."""
    pts = []
    for (h, w) in spatial_shapes:
        ry, rx = torch.meshgrid(torch.linspace(0.5, h - 0.5, h), torch.linspace(0.5, w - 0.5, w), indexing='ij')
        pts.append(torch.stack((rx.reshape(-1)[None] / w, ry.reshape(-1)[None] / h), -1))
    ref = torch.cat(pts, 1)
    return ref[:, :, None].repeat(B, 1, len(spatial_shapes), 1).to(dtype)


def hahi_neck(inputs, P, train_bn=False, prefix='neck', embed=512):
    """HAHIHeteroNeck.forward, necks/hahi.py:235-356 (cross_att=self_att=True, scales all 1)."""
    feats = [conv_module(inputs[i], P, f'{prefix}.lateral_convs.{i}', train_bn) for i in range(len(inputs))]
    feats_trans, feat_conv = feats[1:], feats[0]
    B = feat_conv.shape[0]
    spatial_shapes, srcs, poss = [], [], []
    for i, ft in enumerate(feats_trans):
        _, _, h, w = ft.shape
        spatial_shapes.append((h, w))
        pos = sine_positional_encoding(B, h, w, dtype=ft.dtype).flatten(2).transpose(1, 2)
        poss.append(pos + P[prefix + '.level_embed'][i].view(1, 1, -1))
        srcs.append(conv_module(ft, P, f'{prefix}.trans_proj.{i}', train_bn).flatten(2).transpose(1, 2))
    src_flatten = torch.cat(srcs, 1)
    pos_flatten = torch.cat(poss, 1)
    ref = hahi_reference_points(spatial_shapes, B, feat_conv.dtype)
    src = msda_module(src_flatten, None, pos_flatten, ref, spatial_shapes, P, prefix + '.self_attn')
    conv_skip = conv_module(feat_conv, P, f'{prefix}.conv_proj.0', train_bn)
    bs, c, h, w = conv_skip.shape
    query = conv_skip.flatten(2).transpose(1, 2)
    query_embed = sine_positional_encoding(B, h, w, dtype=query.dtype).flatten(2).transpose(1, 2)
    rp = F.linear(query_embed, P[prefix + '.reference_points.weight'], P[prefix + '.reference_points.bias']).sigmoid()
    rp = rp[:, :, None].repeat(1, 1, len(spatial_shapes), 1)
    fusion = msda_module(query, src, query_embed, rp, spatial_shapes, P, prefix + '.multi_att')
    fusion = fusion.permute(0, 2, 1).reshape(bs, c, h, w)
    outs = [conv_module(torch.cat([fusion, feat_conv], 1), P, f'{prefix}.conv_fusion.0', train_bn, padding=1)]
    start = 0
    for i, ft in enumerate(feats_trans):
        _, _, h, w = ft.shape
        feat = src[:, start:start + h * w].permute(0, 2, 1).contiguous().reshape(bs, embed, h, w)
        start += h * w
        outs.append(conv_module(torch.cat([ft, feat], 1), P, f'{prefix}.trans_fusion.{i}', train_bn, padding=1))
    return outs


def pe_trunk(inputs, P, prefix):
    """Shared trunk of LightPEMASKNeck / DynamicPENeckSOFT (necks/pemask_neck.py:52-63,
    necks/dynamicpe_neck.py:512-538)."""
    xs = list(inputs[::-1])
    size = xs[4].shape[2:]
    acc = None
    for i in range(5):
        t = F.conv2d(xs[i], P[f'{prefix}.conv{i}.weight'], P[f'{prefix}.conv{i}.bias'], padding=1)
        if i < 4:
            t = F.interpolate(t, size=size, mode='bilinear', align_corners=True)
        acc = t if acc is None else acc + t
    return acc


def pe_mask_neck(inputs, P, prefix='pe_mask_neck'):
    """LightPEMASKNeck.forward, necks/pemask_neck.py:52-64 -> y (B,1,H/2,W/2)."""
    x = pe_trunk(inputs, P, prefix)
    return torch.sigmoid(F.conv2d(x, P[prefix + '.convfinal.weight'], P[prefix + '.convfinal.bias'], padding=1))


def dynamic_pe_neck(inputs, P, prefix='dynamic_pe_neck'):
    """DynamicPENeckSOFT.forward, necks/dynamicpe_neck.py:512-539 -> logits (B,11,H/2,W/2)."""
    x = pe_trunk(inputs, P, prefix)
    return F.conv2d(x, P[prefix + '.convfinal.weight'], P[prefix + '.convfinal.bias'], padding=1)


# ======================================================================= ground embedding
def dynamic_pe(logits_lr, y, pe_raw, height=1.65, depth_scale=200.0):
    """DepthEncoderDecoder.dynamic_pe, depther/encoder_decoder.py:79-102.
    logits_lr (B,11,h,w); y (B,1,H,W) already up-sampled; pe_raw = img[:,4] (B,H,W);
    height: float or (B,) tensor.  Returns pe_mask, logits_hr, and the 0/1 validity mask."""
    pe = pe_raw.unsqueeze(1)
    logits = F.interpolate(logits_lr, size=[pe.shape[2], pe.shape[3]], mode='bilinear')
    k = F.softmax(logits, dim=1)
    indices = torch.linspace(-5, 5, 11).view(1, 11, 1, 1).to(logits.dtype)
    k = torch.sum(k * indices, dim=1).unsqueeze(1)
    k = torch.tan(torch.deg2rad(k))
    h = height.view(-1, 1, 1, 1) if torch.is_tensor(height) else height
    a = -h / (pe + 1e-8)
    off = -h / ((a - k) + 1e-8)
    m = off.clone()
    m[m < 0] = 0
    m[m > depth_scale] = 0
    m[m > 0] = 1
    return (off * m) * y, logits, m


def vanilla_pe(y, pe_norm):
    """depther/encoder_decoder.py:120-123: pe_mask = img[:,3:4] * y * 200."""
    return pe_norm.unsqueeze(1) * y * 200


# ================================================================================== head
def densedepth_head(inputs, P, prefix='decode_head'):
    """DenseDepthHead.forward (fpn=False), decode_heads/densedepth_head.py:120-131 + UpSample :14-27."""
    feats = list(inputs[::-1])
    x = F.conv2d(feats[0], P[f'{prefix}.conv_list.0.conv.weight'], P[f'{prefix}.conv_list.0.conv.bias'])
    for i in range(1, len(feats)):
        skip = feats[i]
        up = F.interpolate(x, size=[skip.size(2), skip.size(3)], mode='bilinear', align_corners=True)
        x = torch.cat([up, skip], 1)
        for c in ('convA', 'convB'):
            x = leaky_relu(F.conv2d(x, P[f'{prefix}.conv_list.{i}.{c}.conv.weight'],
                                    P[f'{prefix}.conv_list.{i}.{c}.conv.bias'], padding=1), 0.01, site=f'{prefix}.conv_list.{i}.{c}')
    return x


def depth_pred(feat, pe, y, P, prefix='decode_head', min_depth=1e-3):
    """DepthBaseDecodeHead.depth_pred, decode_heads/decode_head.py:489-508."""
    d = relu(F.conv2d(feat, P[prefix + '.conv_depth.weight'], P[prefix + '.conv_depth.bias'], padding=1), site=prefix + '.conv_depth')
    if pe is None:
        return d + min_depth
    pe = F.interpolate(pe, size=d.shape[2:], mode='bilinear', align_corners=True)
    y = F.interpolate(y, size=d.shape[2:], mode='bilinear', align_corners=True)
    return (d * (1 - y) + pe) + min_depth


def sigloss(pred, target, eps=1e-3, loss_weight=1.0):
    """SigLoss.sigloss, losses/sigloss.py:36-53 (valid_mask=True, max_depth=None, no warm-up)."""
    valid = target > 0
    g = torch.log(pred[valid] + eps) - torch.log(target[valid] + eps)
    return loss_weight * torch.sqrt(torch.var(g) + 0.15 * torch.pow(torch.mean(g), 2))


def ce_loss(logits, target, loss_weight=0.08):
    """CrossEntropyLoss(ignore_index=255)·0.08, losses/celoss.py:354-413 + decode_head.py:313-316."""
    return loss_weight * F.cross_entropy(logits, target.long(), ignore_index=255)


# ================================================================================= depther
def extract_feat(img, P, cfg, train_bn=False, height=1.65):
    """DepthEncoderDecoder.extract_feat, depther/encoder_decoder.py:105-124."""
    x = backbone(img, P, cfg, train_bn)
    x = hahi_neck(x, P, train_bn)
    y = pe_mask_neck(x, P)
    y = F.interpolate(y, size=[img.shape[2], img.shape[3]], mode='bilinear')
    if cfg.get('adaptive', False):
        pe_mask, logits, _ = dynamic_pe(dynamic_pe_neck(x, P), y, img[:, 4], height, cfg.get('depth_scale', 200.0))
        return x, y, pe_mask, logits
    return x, y, vanilla_pe(y, img[:, 3]), None


def forward_train(img, depth_gt, pe_k_gt, P, cfg, train_bn=True, height=1.65):
    """forward_train -> losses, encoder_decoder.py:170-195 + decode_head.py:415-441,511-542,582-626."""
    x, y, pe_mask, logits = extract_feat(img, P, cfg, train_bn, height)
    pred = depth_pred(densedepth_head(x, P), pe_mask, y, P)
    pred_up = F.interpolate(pred, size=depth_gt.shape[2:], mode='bilinear', align_corners=True)
    losses = {}
    if logits is not None:
        losses['decode.loss_dynamic_pe'] = ce_loss(logits, pe_k_gt)
    losses['decode.loss_depth'] = sigloss(pred_up, depth_gt)
    return losses, pred


def encode_decode(img, P, cfg, min_depth=1e-3, max_depth=80.0, height=1.65):
    """encode_decode, encoder_decoder.py:126-139: clamp THEN bilinear resize (align_corners=True)."""
    x, y, pe_mask, _ = extract_feat(img, P, cfg, False, height)
    out = depth_pred(densedepth_head(x, P), pe_mask, y, P, min_depth=min_depth)
    out = torch.clamp(out, min=min_depth, max=max_depth)
    return F.interpolate(out, size=img.shape[2:], mode='bilinear', align_corners=True)


def inference(img, img_meta, P, cfg, **kw):
    """inference + whole_inference, encoder_decoder.py:196-235: encode_decode, rescale to ``ori_shape`` (bilinear, the head's
    ``align_corners``), and the prediction of a flipped view flipped back."""
    out = encode_decode(img, P, cfg, **kw)
    ori = tuple(img_meta[0]['ori_shape'][:2])
    if tuple(out.shape[2:]) != ori:
        out = F.interpolate(out, size=ori, mode='bilinear', align_corners=True)
    if img_meta[0].get('flip'):
        direction = img_meta[0]['flip_direction']
        assert direction in ('horizontal', 'vertical')
        out = out.flip(dims=(3,)) if direction == 'horizontal' else out.flip(dims=(2,))
    return out


def aug_test(imgs, img_metas, P, cfg, **kw):
    """aug_test, encoder_decoder.py:249-274 (flip test-time augmentation): the mean of ``inference`` over the augmented views."""
    pred = inference(imgs[0], img_metas[0], P, cfg, **kw)
    for i in range(1, len(imgs)):
        pred = pred + inference(imgs[i], img_metas[i], P, cfg, **kw)
    return pred / len(imgs)


def parse_losses(losses):
    """BaseDepther._parse_losses (single process), depther/base.py:170-204."""
    log_vars = {k: v.mean() for k, v in losses.items()}
    loss = sum(v for k, v in log_vars.items() if 'loss' in k)
    log_vars['loss'] = loss
    return loss, {k: float(v) for k, v in log_vars.items()}


def log_images(img, depth_pred, depth_gt, mean, std, to_rgb=True):
    """DepthBaseDecodeHead.log_images, decode_heads/decode_head.py:628-648, for one sample: img (C>=3, H, W) normalised, depth_pred / depth_gt
    (1, H, W).  ``mmcv.imdenormalize(x, mean, std, to_bgr)`` [mmcv 1.3.13, not vendored] is ``x * std + mean`` in float32 followed by an
    RGB<->BGR swap when ``to_bgr``; the method passes ``to_bgr = to_rgb`` (:632-635), clips to [0, 255], truncates to uint8 (:636-637),
    reverses the channel axis once more (:638) and moves it to the front (:639-640); both depth maps are divided by their maximum (:642-643)."""
    show = img[:3].permute(1, 2, 0).numpy().astype(np.float32)
    show = show * np.asarray(std, np.float32).reshape(1, -1) + np.asarray(mean, np.float32).reshape(1, -1)
    if to_rgb:
        show = show[..., ::-1]
    show = np.clip(show, 0, 255).astype(np.uint8)[:, :, ::-1]
    show = show.transpose(0, 2, 1).transpose(1, 0, 2)
    return dict(img_rgb=np.ascontiguousarray(show), img_depth_pred=depth_pred / depth_pred.max(), img_depth_gt=depth_gt / depth_gt.max())


# ============================================================ offline ground-plane (float64)
def ground_plane(P2, R0_rect, Tr_velo_to_cam, height_img, width_img, cam_height=1.65):
    """tools/preprocess_data_kitti.py:29-53: depth at which each pixel's ray meets the plane
    ``cam_height`` below the camera.  P2 (3,4); R0_rect (3,3); Tr (3,4) or (4,4).  float64."""
    R0 = np.eye(4)
    R0[:3, :3] = np.asarray(R0_rect, dtype=np.float64)
    Tr = np.eye(4)
    Tr[:np.asarray(Tr_velo_to_cam).shape[0], :] = np.asarray(Tr_velo_to_cam, dtype=np.float64)
    A = np.asarray(P2, dtype=np.float64) @ R0 @ Tr
    Rinv = np.linalg.inv(A[0:3, 0:3])
    RT = Rinv @ A[0:3, 3]
    u, v = np.meshgrid(range(width_img), range(height_img), indexing='xy')
    pe = (RT[2] - cam_height) / (Rinv[2, 0] * u + Rinv[2, 1] * v + Rinv[2, 2])
    return pe, Rinv[2].copy(), float(RT[2] - cam_height)


def slope_class(gt, pe, cam_height=1.65, mode='round'):
    """tools/preprocess_data_kitti.py:59-63,83-89 (mode='round', gt float64, pe float32) and
    tools/preprocess_data_ddad.py:78 (mode='trunc').  -> float64 map in {-5..5, 255}."""
    gt = np.asarray(gt, dtype=np.float64)
    pe = np.asarray(pe, dtype=np.float32)
    with np.errstate(divide='ignore', invalid='ignore'):
        a = (np.float32(-cam_height)) / pe
        b = cam_height / gt
        k = b + a
        deg = np.rad2deg(np.arctan(k))
        k = np.around(deg) if mode == 'round' else np.trunc(deg)
    k[k > 5] = 5
    k[k < -5] = -5
    k[gt == 0] = 255
    return k


def ground_plane_ddad(intrinsics, camera_pose, lidar_pose, height_img, width_img):
    """tools/preprocess_data_ddad.py:29-41: A = K4 @ inv(camera_pose) @ lidar_pose; the plane is the lidar frame's z = 0
    (numerator RT[2], NO camera-height term — the per-camera heights enter only in find_k / dynamic_pe).  float64."""
    K4 = np.eye(4)
    K4[:3, :3] = np.asarray(intrinsics, dtype=np.float64)
    A = K4 @ np.linalg.inv(np.asarray(camera_pose, dtype=np.float64)) @ np.asarray(lidar_pose, dtype=np.float64)
    Rinv = np.linalg.inv(A[:3, :3])
    RT = Rinv @ A[0:3, 3]
    u, v = np.meshgrid(range(width_img), range(height_img), indexing='xy')
    pe = RT[2] / (Rinv[2, 0] * u + Rinv[2, 1] * v + Rinv[2, 2])
    return pe, Rinv[2].copy(), float(RT[2])


def slope_class_ddad(gt, pe, cam_height):
    """tools/preprocess_data_ddad.py:47-51,68-78: gt is the float32 depth of the .npz, pe the float64 map, h a Python
    float: ``a = -h/pe`` (float64), ``b = h/gt`` (float32: scalar / float32 array), ``k = b + a`` (float64),
    ``rad2deg(arctan(k)).astype(int64)`` (truncation), clip to +-5, 255 where gt == 0.  -> int64 map."""
    gt = np.asarray(gt, dtype=np.float32)
    pe = np.asarray(pe, dtype=np.float64)
    with np.errstate(divide='ignore', invalid='ignore'):
        a = -cam_height / pe
        b = np.float32(cam_height) / gt
        k = np.rad2deg(np.arctan(b + a)).astype(np.int64)
    k[k > 5] = 5
    k[k < -5] = -5
    k[gt == 0] = 255
    return k


DDAD_CAMERA_HEIGHTS = {'CAMERA_01': 1.56, 'CAMERA_05': 1.57, 'CAMERA_06': 1.53, 'CAMERA_09': 1.53}   # preprocess_data_ddad.py:68-75


def loader_pe_channels(pe, depth_scale=200.0):
    """datasets/pipelines/loading.py:366-403 + transforms.py:40-48: channel 3 = filtered pe / depth_scale
    (>200 -> 0, <0 -> 0), channel 4 = raw pe (float32)."""
    raw = np.asarray(pe).astype(np.float32)
    filt = raw.copy()
    filt[filt > 200] = 0
    filt[filt < 0] = 0
    norm = filt.copy()
    norm[filt > 0] = filt[filt > 0] / depth_scale
    return norm, raw


# ================================================================================ metrics
def metrics_calculate(gt, pred):
    """core/evaluation/metrics.py:8-33."""
    if gt.shape[0] == 0:
        return (np.nan,) * 9
    thresh = np.maximum(gt / pred, pred / gt)
    a1, a2, a3 = (thresh < 1.25).mean(), (thresh < 1.25 ** 2).mean(), (thresh < 1.25 ** 3).mean()
    abs_rel = np.mean(np.abs(gt - pred) / gt)
    sq_rel = np.mean(((gt - pred) ** 2) / gt)
    rmse = np.sqrt(((gt - pred) ** 2).mean())
    rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    err = np.log(pred) - np.log(gt)
    silog = np.sqrt(np.mean(err ** 2) - np.mean(err) ** 2) * 100
    if np.isnan(silog):
        silog = 0
    log_10 = np.abs(np.log10(gt) - np.log10(pred)).mean()
    return a1, a2, a3, abs_rel, rmse, log_10, rmse_log, silog, sq_rel


def metrics(gt, pred, min_depth=1e-3, max_depth=80):
    """core/evaluation/metrics.py:35-45."""
    mask = np.logical_and(gt > min_depth, gt < max_depth)
    return metrics_calculate(gt[mask], pred[mask])


SWIN_T = dict(embed_dims=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24])
SWIN_L = dict(embed_dims=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48])
