"""Generate tests/golden/*.npz by importing the REFERENCE (read-only, /root/reference) on CPU.

Runs only in the build container (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference needs mmcv-full 1.3.13, cv2, IPython, prettytable ... none of which exist here,
so this script installs a stand-in: permissive dummy modules for everything
the path never executes, and real implementations (from gedepth_amd.mmrt — restatements of the
mmcv semantics in SURVEY.md Appendix A — plus a pure-PyTorch MultiScaleDeformableAttention)
for the symbols the hot path does execute.  The reference's own Python for
backbone / necks / heads / losses / depther then runs unmodified.

Outputs are DATA (seeded inputs + reference outputs); weights are regenerated from
oracle/fill.py's name-keyed rule, so only names/shapes are stored.
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from gedepth_amd.mmrt import bricks, registry  # noqa: E402
from gedepth_amd.mmrt.config import Config, ConfigDict  # noqa: E402
from oracle import gedepth_oracle as O  # noqa: E402
from oracle.fill import fill_state_dict, load_filled  # noqa: E402


# --------------------------------------------------------------------------- mmcv stand-in
class _DummyMeta(type):
    def __getattr__(cls, name):
        if name.startswith('__') and name.endswith('__'):
            raise AttributeError(name)
        return _make_dummy(name)

    def __call__(cls, *args, **kwargs):
        if cls.__dict__.get('_is_dummy_root', False):
            inst = type.__call__(cls)
            return inst(*args, **kwargs) if (len(args) == 1 and callable(args[0]) and not kwargs) else inst
        return type.__call__(cls, *args, **kwargs)


def _make_dummy(name):
    def _call(self, *args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return self

    def _getattr(self, item):
        if item.startswith('__') and item.endswith('__'):
            raise AttributeError(item)
        return _make_dummy(item)

    return _DummyMeta(name, (), {'_is_dummy_root': True, '__call__': _call, '__getattr__': _getattr,
                                 '__init__': lambda self, *a, **k: None})


class _DummyModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__') and name.endswith('__'):
            raise AttributeError(name)
        return _make_dummy(name)


def _identity_decorator(*dargs, **dkwargs):
    def deco(fn):
        return fn
    return deco


class MultiScaleDeformableAttention(bricks.BaseModule):
    """mmcv 1.3.13 module semantics on the pure-PyTorch sampling core."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64,
                 dropout=0.1, batch_first=False, norm_cfg=None, init_cfg=None):
        super().__init__(init_cfg)
        self.dropout = nn.Dropout(dropout)
        self.batch_first = batch_first
        self.embed_dims, self.num_levels, self.num_heads, self.num_points = embed_dims, num_levels, num_heads, num_points
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        bricks.constant_init(self.sampling_offsets, 0.)
        self.sampling_offsets.bias.data = bricks.msda_offset_bias(self.num_heads, self.num_levels, self.num_points)
        bricks.constant_init(self.attention_weights, val=0., bias=0.)
        bricks.xavier_init(self.value_proj, distribution='uniform', bias=0.)
        bricks.xavier_init(self.output_proj, distribution='uniform', bias=0.)
        self._is_init = True

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, **kwargs):
        if value is None:
            value = query
        if identity is None:
            identity = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query, value = query.permute(1, 0, 2), value.permute(1, 0, 2)
        bs, num_query, _ = query.shape
        bs, num_value, _ = value.shape
        value = self.value_proj(value).view(bs, num_value, self.num_heads, -1)
        off = self.sampling_offsets(query).view(bs, num_query, self.num_heads, self.num_levels, self.num_points, 2)
        aw = self.attention_weights(query).view(bs, num_query, self.num_heads, self.num_levels * self.num_points)
        aw = aw.softmax(-1).view(bs, num_query, self.num_heads, self.num_levels, self.num_points)
        normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
        loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
        out = O.msda_core(value, [(int(h), int(w)) for h, w in spatial_shapes], loc, aw)
        out = self.output_proj(out)
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        return self.dropout(out) + identity


def install_shim():
    import logging

    def mod(name):
        m = _DummyModule(name)
        m.__path__ = []
        sys.modules[name] = m
        return m

    for n in ['cv2', 'IPython', 'prettytable', 'pyinn', 'deep_hough', 'mmseg', 'mmseg.ops']:
        mod(n)
    names = ['mmcv', 'mmcv.cnn', 'mmcv.cnn.bricks', 'mmcv.cnn.bricks.transformer', 'mmcv.cnn.bricks.registry',
             'mmcv.cnn.bricks.drop', 'mmcv.cnn.utils', 'mmcv.cnn.utils.weight_init', 'mmcv.runner',
             'mmcv.runner.base_module', 'mmcv.runner.dist_utils', 'mmcv.runner.hooks', 'mmcv.runner.hooks.logger',
             'mmcv.runner.hooks.logger.base', 'mmcv.utils', 'mmcv.utils.parrots_wrapper', 'mmcv.ops',
             'mmcv.ops.multi_scale_deform_attn', 'mmcv.parallel', 'mmcv.engine', 'mmcv.image']
    M = {n: mod(n) for n in names}
    for n in names:
        if '.' in n:
            parent, child = n.rsplit('.', 1)
            setattr(M[parent], child, M[n])
    mm = M['mmcv']
    mm.__version__ = '1.3.13'
    mm.ConfigDict = ConfigDict
    mm.Config = Config

    def imdenormalize(img, mean, std, to_bgr=True):
        img = img * np.asarray(std, dtype=np.float32).reshape(1, -1) + np.asarray(mean, dtype=np.float32).reshape(1, -1)
        return img[..., ::-1] if to_bgr else img
    mm.imdenormalize = imdenormalize

    real = dict(
        Registry=registry.Registry, build_from_cfg=registry.build_from_cfg,
        BaseModule=bricks.BaseModule, ModuleList=bricks.ModuleList, Sequential=bricks.Sequential,
        ConvModule=bricks.ConvModule, build_norm_layer=bricks.build_norm_layer,
        build_conv_layer=bricks.build_conv_layer, build_activation_layer=bricks.build_activation_layer,
        FFN=bricks.FFN, DropPath=bricks.DropPath, build_dropout=bricks.build_dropout,
        constant_init=bricks.constant_init, xavier_init=bricks.xavier_init, kaiming_init=bricks.kaiming_init,
        trunc_normal_init=bricks.trunc_normal_init, MODELS=bricks.MODELS, ATTENTION=bricks.ATTENTION,
        POSITIONAL_ENCODING=bricks.POSITIONAL_ENCODING, build_positional_encoding=bricks.build_positional_encoding,
        MultiScaleDeformableAttention=MultiScaleDeformableAttention, Linear=nn.Linear,
        auto_fp16=_identity_decorator, force_fp32=_identity_decorator,
        get_logger=lambda name, log_file=None, log_level=logging.INFO: logging.getLogger(name),
        print_log=lambda *a, **k: None, to_2tuple=lambda x: x if isinstance(x, tuple) else (x, x),
        _BatchNorm=nn.modules.batchnorm._BatchNorm, TORCH_VERSION=torch.__version__,
        digit_version=lambda v: tuple(int(x) for x in v.split('+')[0].split('.')[:3] if x.isdigit()),
        get_dist_info=lambda: (0, 1),
    )
    for m in M.values():
        for k, v in real.items():
            setattr(m, k, v)
    if not torch.cuda.is_available():
        torch.cuda.current_device = lambda: 'cpu'   # depther/encoder_decoder.py:68
    sys.path.insert(0, REF)


# --------------------------------------------------------------------------------- helpers
OUT = {}


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    conv = {}
    for k, v in arrays.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    np.savez_compressed(path, **conv)
    print(f'  wrote {name}.npz  ({os.path.getsize(path) / 1024:.0f} KiB)')


def spec_of(module):
    return json.dumps([[k, list(v.shape)] for k, v in module.state_dict().items()])


def gen(seed):
    return torch.Generator().manual_seed(seed)


def grad_sample(g, cap=50000):
    """Large gradients are stored as a strided sample (tests apply the same rule)."""
    g = g.detach().flatten()
    return g[::max(1, g.numel() // cap)].clone()


def synth_img(B, H, W, seed):
    g = gen(seed)
    img = torch.zeros(B, 5, H, W)
    img[:, 0:3] = torch.randn(B, 3, H, W, generator=g)
    v = torch.arange(H, dtype=torch.float32).view(1, H, 1).expand(B, H, W) * (352.0 / H)
    pe_raw = 1.65 * 721.5377 / (v - 172.854)
    img[:, 4] = pe_raw
    img[:, 3] = torch.where((pe_raw > 0) & (pe_raw <= 200), pe_raw, torch.zeros_like(pe_raw)) / 200.0
    valid = torch.rand(B, 1, H, W, generator=g) < 0.3
    depth_gt = torch.where(valid, 1 + 79 * torch.rand(B, 1, H, W, generator=g), torch.zeros(B, 1, H, W))
    cls = torch.clamp(torch.round(torch.randn(B, H, W, generator=g) * 1.5), -5, 5) + 5
    pe_k_gt = torch.where(valid[:, 0], cls, torch.full((B, H, W), 255.0))
    return img, depth_gt, pe_k_gt


def main():
    install_shim()
    import depth.models  # noqa: F401  (the reference, unmodified)
    from depth.models import build_depther
    from depth.models.backbones import depthformer_swin as RS
    from depth.models.necks.hahi import HAHIHeteroNeck
    from depth.models.necks.pemask_neck import LightPEMASKNeck
    from depth.models.necks.dynamicpe_neck import DynamicPENeckSOFT
    from depth.models.losses import SigLoss, CrossEntropyLoss
    from depth.models.utils import PatchEmbedSwin, swin_convert
    from depth.utils.position_encoding import SinePositionalEncoding
    from depth.core.evaluation.metrics import calculate
    from depth.models.depther.base import BaseDepther
    torch.manual_seed(0)
    torch.set_grad_enabled(False)

    # ---- 1. WindowMSA (depthformer_swin.py:184-224)
    print('window_msa')
    m = RS.WindowMSA(embed_dims=96, num_heads=3, window_size=(7, 7)).eval()
    load_filled(m, 'window_msa')
    x = torch.randn(8, 49, 96, generator=gen(1))
    mask = O.shift_mask(14, 14)  # 4 windows
    save('window_msa', spec=spec_of(m), x=x, mask=mask, out_nomask=m(x), out_mask=m(x, mask),
         rel_index=m.relative_position_index)

    # ---- 2. ShiftWindowMSA (depthformer_swin.py:285-360), padding + shift
    print('shift_window_msa')
    arrays = {}
    for tag, (H, W) in {'a': (11, 35), 'b': (10, 9)}.items():
        for shift in (0, 3):
            m = RS.ShiftWindowMSA(embed_dims=96, num_heads=3, window_size=7, shift_size=shift,
                                  dropout_layer=dict(type='DropPath', drop_prob=0.)).eval()
            load_filled(m, 'shift_window_msa')
            x = torch.randn(2, H * W, 96, generator=gen(10 + H + shift))
            arrays[f'x_{tag}{shift}'] = x
            arrays[f'out_{tag}{shift}'] = m(x, (H, W))
            arrays['spec'] = spec_of(m)
    save('shift_window_msa', **arrays)

    # ---- 3. SwinBlock + PatchMerging + PatchEmbed
    print('swin_block / patch_merging / patch_embed')
    m = RS.SwinBlock(embed_dims=96, num_heads=3, feedforward_channels=384, shift=True, drop_path_rate=0.).eval()
    load_filled(m, 'swin_block')
    x = torch.randn(2, 11 * 13, 96, generator=gen(3))
    save('swin_block', spec=spec_of(m), x=x, out=m(x, (11, 13)))
    m = RS.PatchMerging(in_channels=96, out_channels=192).eval()
    load_filled(m, 'patch_merging')
    x = torch.randn(2, 5 * 7, 96, generator=gen(4))
    out, hw = m(x, (5, 7))
    save('patch_merging', spec=spec_of(m), x=x, out=out, hw=np.array(hw))
    m = PatchEmbedSwin(in_channels=4, embed_dims=96, conv_type='Conv2d', kernel_size=4, stride=4,
                       pad_to_patch_size=True, norm_cfg=dict(type='LN')).eval()
    load_filled(m, 'patch_embed')
    x = torch.randn(2, 4, 18, 30, generator=gen(5))
    save('patch_embed', spec=spec_of(m), x=x, out=m(x), hw=np.array([m.DH, m.DW]))

    # ---- 4. MSDA core + module (mmcv restatement; independent cross-check lives in the tests)
    print('msda')
    shapes = [(11, 35), (6, 18), (3, 9), (2, 5)]
    nv = sum(h * w for h, w in shapes)
    g = gen(6)
    value = torch.randn(2, nv, 8, 64, generator=g)
    loc = torch.rand(2, 50, 8, 4, 8, 2, generator=g) * 1.3 - 0.15
    aw = torch.rand(2, 50, 8, 4, 8, generator=g).flatten(-2).softmax(-1).view(2, 50, 8, 4, 8)
    save('msda_core', value=value, loc=loc, aw=aw, shapes=np.array(shapes), out=O.msda_core(value, shapes, loc, aw))
    m = MultiScaleDeformableAttention(embed_dims=512, num_levels=4, num_heads=8, num_points=8, batch_first=True).eval()
    load_filled(m, 'msda_module')
    q = torch.randn(2, 60, 512, generator=g)
    v = torch.randn(2, nv, 512, generator=g)
    qp = torch.randn(2, 60, 512, generator=g)
    ref = torch.rand(2, 60, 4, 2, generator=g)
    out = m(q, value=v, query_pos=qp, reference_points=ref, spatial_shapes=torch.as_tensor(shapes))
    save('msda_module', spec=spec_of(m), q=q, v=v, qp=qp, ref=ref, shapes=np.array(shapes), out=out)

    # ---- 5. SinePositionalEncoding (depth/utils/position_encoding.py:54-89)
    print('sine pos')
    pe = SinePositionalEncoding(num_feats=256)
    save('sine_pos', out=pe(torch.zeros(1, 5, 7, dtype=torch.bool)))

    # ---- 6. HAHI neck (hahi.py:235-356), Swin-T widths, tiny spatial, eval-BN and train-BN
    print('hahi')
    chans = [64, 96, 192, 384, 768]
    sizes = [(32, 48), (16, 24), (8, 12), (4, 6), (2, 3)]
    m = HAHIHeteroNeck(in_channels=chans, out_channels=chans, embedding_dim=512, scales=[1] * 5,
                       positional_encoding=dict(type='SinePositionalEncoding', num_feats=256))
    load_filled(m, 'hahi')
    m.multi_att.dropout.p = 0.0
    m.self_attn.dropout.p = 0.0
    g = gen(7)
    feats = [torch.randn(2, c, h, w, generator=g) for c, (h, w) in zip(chans, sizes)]
    arrays = {f'in{i}': f for i, f in enumerate(feats)}
    m.eval()
    for i, o in enumerate(m(feats)):
        arrays[f'eval_out{i}'] = o
    m.train()
    for i, o in enumerate(m(feats)):
        arrays[f'train_out{i}'] = o
    save('hahi', spec=spec_of(m), **arrays)

    # ---- 7. PE necks (pemask_neck.py:29-64, dynamicpe_neck.py:490-539), Swin-L widths
    print('pe necks')
    chans = [64, 192, 384, 768, 1536]
    sizes = [(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)]
    g = gen(8)
    feats = [torch.randn(2, c, h, w, generator=g) * 0.5 for c, (h, w) in zip(chans, sizes)]
    m1 = LightPEMASKNeck().eval()
    load_filled(m1, 'pe_mask_neck')
    m2 = DynamicPENeckSOFT().eval()
    load_filled(m2, 'dynamic_pe_neck')
    y, _ = m1(feats)
    save('pe_necks', spec_mask=spec_of(m1), spec_dyn=spec_of(m2), y=y, logits=m2(feats),
         **{f'in{i}': f for i, f in enumerate(feats)})

    # ---- 8. e2e models (Swin-T-V, Swin-L-A) at 2x5x64x96: losses, pred, eval depth, a few grads
    def cfg_for(arch, adaptive):
        e = dict(T=dict(embed_dims=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24]),
                 L=dict(embed_dims=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48]))[arch]
        C = e['embed_dims']
        ch = [64, C, 2 * C, 4 * C, 8 * C]
        model = dict(
            type='DepthEncoderDecoder', pretrained=None,
            backbone=dict(type='DepthFormerSwin', pretrain_img_size=224, patch_size=4, window_size=7, mlp_ratio=4,
                          strides=(4, 2, 2, 2), out_indices=(0, 1, 2, 3), qkv_bias=True, qk_scale=None,
                          patch_norm=True, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.0,
                          use_abs_pos_embed=False, act_cfg=dict(type='GELU'), norm_cfg=dict(type='LN', requires_grad=True),
                          pretrain_style='official', conv_norm_cfg=dict(type='BN', requires_grad=True), depth=50,
                          num_stages=0, USEPE=True, **e),
            neck=dict(type='HAHIHeteroNeck', positional_encoding=dict(type='SinePositionalEncoding', num_feats=256),
                      in_channels=ch, out_channels=ch, embedding_dim=512, scales=[1, 1, 1, 1, 1]),
            pe_mask_neck=dict(type='LightPEMASKNeck'),
            decode_head=dict(type='DenseDepthHead', act_cfg=dict(type='LeakyReLU', inplace=True), in_channels=ch,
                             up_sample_channels=ch, channels=64, align_corners=True, min_depth=1e-3, max_depth=80,
                             loss_decode=dict(type='SigLoss', valid_mask=True, loss_weight=1.0)),
            train_cfg=dict(), test_cfg=dict(mode='whole'))
        if adaptive:
            model['dynamic_pe_neck'] = dict(type='DynamicPENeckSOFT')
        return ConfigDict(Config(dict(model=model)).model), ch

    for arch, adaptive in (('T', False), ('L', True), ('T', True)):
        tag = f'e2e_{arch}_{"A" if adaptive else "V"}'
        print(tag)
        with torch.enable_grad():
            cfg, ch = cfg_for(arch, adaptive)
            model = build_depther(cfg)
            if arch == 'T':  # the PE necks hard-code Swin-L widths -> harness-side conv swap
                for neck in [model.pe_mask_neck] + ([model.dynamic_pe_neck] if adaptive else []):
                    for i, c in enumerate(ch[::-1][:4]):
                        setattr(neck, f'conv{i}', nn.Conv2d(c, 64, kernel_size=3, padding=1, stride=1))
            load_filled(model, 'e2e')
            model.neck.multi_att.dropout.p = 0.0
            model.neck.self_attn.dropout.p = 0.0
            img, depth_gt, pe_k_gt = synth_img(2, 64, 96, seed=99)
            metas = [dict(img_norm_cfg=dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True),
                          flip=False, ori_shape=(64, 96, 3))] * 2
            model.eval()
            with torch.no_grad():
                depth_eval = model.encode_decode(img, metas)
            if arch == 'T':
                # the test path through the reference's OWN methods (encoder_decoder.py:196-274): simple_test on the plain view and
                # aug_test over (view, horizontally flipped view) with the flip-back inside `inference`; a second pair of views comes
                # at another size so that the rescale inside encode_decode is exercised through aug_test as well
                metas_f = [dict(metas[0], flip=True, flip_direction='horizontal')] * 2
                metas_n = [dict(metas[0], flip_direction='horizontal')] * 2
                zero_pt = [torch.zeros(2, 2), torch.zeros(2, 2)]
                with torch.no_grad():
                    kw_t = dict(pe_ori_point=zero_pt)
                    if adaptive:
                        kw_t['pe_k_gt'] = [pe_k_gt, pe_k_gt.flip(-1)]
                    simple = np.stack(model.simple_test(img, metas_n, rescale=True, **kw_t))
                    aug = np.stack(model.aug_test([img, img.flip(3)], [metas_n, metas_f], rescale=True, **kw_t))
                    vflip = [dict(metas[0], flip=True, flip_direction='vertical')] * 2
                    aug_v = np.stack(model.aug_test([img, img.flip(2)], [metas_n, vflip], rescale=True, **kw_t))
                save(f'test_path_{arch}_{"A" if adaptive else "V"}', img=img, simple=simple, aug_h=aug, aug_v=aug_v)
            model.train()
            state_before = {k: v.clone() for k, v in model.state_dict().items()}
            kw = dict(pe_k_gt=pe_k_gt) if adaptive else {}
            losses = model.forward_train(img, metas, depth_gt, **kw)
            real = {k: v for k, v in losses.items() if 'img' not in k}
            loss, log_vars = BaseDepther._parse_losses(real)
            loss.backward()
            names = ['backbone.patch_embed.projection.weight', 'backbone.conv1.weight',
                     'backbone.stages.0.blocks.1.attn.w_msa.relative_position_bias_table',
                     'backbone.stages.0.blocks.1.attn.w_msa.qkv.bias',
                     'backbone.stages.0.blocks.0.attn.w_msa.qkv.bias',
                     'backbone.stages.2.blocks.1.ffn.layers.1.weight',
                     'neck.multi_att.sampling_offsets.bias', 'neck.self_attn.attention_weights.bias',
                     'neck.level_embed', 'neck.reference_points.weight', 'neck.conv_fusion.0.bn.weight',
                     'pe_mask_neck.convfinal.weight', 'decode_head.conv_depth.weight',
                     'decode_head.conv_list.1.convA.conv.bias']
            if adaptive:
                names.append('dynamic_pe_neck.convfinal.bias')
            params = dict(model.named_parameters())
            assert all(p.grad is not None for p in params.values()), 'unused parameter'
            arrays = {'grad::' + n: grad_sample(params[n].grad) for n in names}
            arrays['grad_norm_total'] = torch.sqrt(sum((p.grad ** 2).sum() for p in params.values()))
            save(tag, spec=json.dumps([[k, list(v.shape)] for k, v in state_before.items()]),
                 img=img, depth_gt=depth_gt, pe_k_gt=pe_k_gt, depth_eval=depth_eval,
                 loss_names=json.dumps(list(log_vars.keys())), loss_values=np.array(list(log_vars.values())),
                 **arrays)

    # ---- 9. closed-form known answers
    print('known answers')
    sl = SigLoss(valid_mask=True, loss_weight=1.0)
    pred = torch.tensor([[1., 2., 4., 8.], [10., 20., 40., 79.]])
    gt = torch.tensor([[1.5, 0., 3., 8.], [12., 25., 0., 60.]])
    ce = CrossEntropyLoss(loss_weight=0.08)
    logits = torch.randn(2, 11, 4, 5, generator=gen(11))
    tgt = torch.randint(0, 11, (2, 4, 5), generator=gen(12)).float()
    tgt[0, 0, :3] = 255
    mt = calculate(np.array([1.5, 3, 8, 12, 25, 60.]), np.array([1., 4, 8, 10, 20, 79.]))
    # swin_convert key/permute map (models/utils/ckpt_convert.py:5-56) on a tiny synthetic official-style ckpt
    off = {'layers.0.blocks.0.attn.qkv.weight': torch.arange(6.).view(2, 3),
           'layers.0.blocks.0.mlp.fc1.weight': torch.arange(4.).view(2, 2),
           'layers.0.blocks.0.mlp.fc2.bias': torch.arange(2.),
           'layers.0.downsample.reduction.weight': torch.arange(16.).view(2, 8),
           'layers.0.downsample.norm.weight': torch.arange(8.),
           'patch_embed.proj.weight': torch.arange(4.), 'head.weight': torch.zeros(1), 'norm.weight': torch.ones(2)}
    conv = swin_convert(off)
    save('known_answers', sig_pred=pred, sig_gt=gt, sig=sl(pred, gt), ce_logits=logits, ce_target=tgt,
         ce=ce(logits, tgt.long()), metrics=np.array(mt, dtype=np.float64),
         swin_convert_keys=json.dumps(list(conv.keys())),
         swin_convert_reduction=conv['stages.0.downsample.reduction.weight'],
         swin_convert_norm=conv['stages.0.downsample.norm.weight'])

    # ---- 10. dynamic_pe / vanilla (encoder_decoder.py:79-102,120-123) through the reference method
    print('dynamic_pe')
    cfg, _ = cfg_for('L', True)
    with torch.enable_grad():
        dm = build_depther(cfg)

    class _Neck(nn.Module):
        def __init__(self, t):
            super().__init__()
            self.t = t

        def forward(self, x):
            return self.t
    g = gen(13)
    img, _, _ = synth_img(2, 24, 40, seed=5)
    img[:, 4] += 0.05 * torch.randn(2, 24, 40, generator=g)
    logits_lr = 2.0 * torch.randn(2, 11, 12, 20, generator=g)
    y = torch.rand(2, 1, 24, 40, generator=g)
    dm.dynamic_pe_neck = _Neck(logits_lr)
    pe_mask, logits_hr = dm.dynamic_pe(None, y, img, None)
    heights = torch.tensor([1.56, 1.53])
    pe_mask_h, _ = dm.dynamic_pe(None, y, img, None, height=heights)
    save('dynamic_pe', img=img, logits_lr=logits_lr, y=y, pe_mask=pe_mask, logits_hr=logits_hr,
         heights=heights, pe_mask_h=pe_mask_h, vanilla=img[:, 3:4] * y * 200)
    print('done')


if __name__ == '__main__':
    main()
