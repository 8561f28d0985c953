"""Per-module parity ON THE GPU against the fixtures the reference itself produced (tests/golden/make_golden.py): the product
modules (HIP kernels + library GEMM / conv) are fed the reference's inputs with the reference's weights (oracle/fill.py: the same
deterministic fill by parameter name and shape) and must reproduce the reference's outputs.  These are the fixtures that
tests/test_oracle_golden.py uses to pin the CPU oracle; here they reach the HIP path directly, module by module, at geometries the
2 x 5 x 64 x 96 end-to-end fixtures do not visit (explicit-mask windows without padding, odd widths, un-padded patch grids)."""
import numpy as np
import pytest
import torch

from oracle.fill import load_filled

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    return torch.device('cuda:0')


def close(a, b, rtol=1e-4, atol=1e-5, what=''):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f'{what}: max abs err {err.max():.3e}, max rel {(err / (b.abs() + 1e-12)).max():.3e}'


def window_reverse(windows, H, W):
    """(B * nW, 49, C) windows in raster order -> (B, H, W, C) (depthformer_swin.py:362-376)."""
    nW = (H // 7) * (W // 7)
    B = windows.shape[0] // nW
    x = windows.view(B, H // 7, W // 7, 7, 7, -1)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


def test_window_msa_explicit_mask_fixture(dev, golden):
    """`window_msa.npz`: the reference's WindowMSA on 8 ready-made windows, without a mask and with the shift mask of a 14 x 14 map
    passed explicitly.  The product has no mask argument — pad / roll / partition / mask are index arithmetic inside the HIP kernel —
    so the windows are laid back into the 2 x 14 x 14 token maps they are windows OF (with shift 3: rolled back by +3) and the
    kernel's own analytic mask must reproduce `out_mask`."""
    from gedepth_amd.depth.models.backbones.depthformer_swin import WindowMSA
    g = golden('window_msa')
    m = WindowMSA(embed_dims=96, num_heads=3, window_size=(7, 7)).eval()
    load_filled(m, 'window_msa')
    m = m.to(dev)
    x, ref_plain, ref_mask = T(g['x']), T(g['out_nomask']), T(g['out_mask'])
    for variant in (1, 0):                                              # exact-fp32 VALU kernel, then the default (fp32 input -> exact too)
        with torch.no_grad():
            out = m(window_reverse(x, 14, 14).reshape(2, 196, 96).to(dev), (14, 14), 0, variant)
        close(out.view(2, 14, 14, 96), window_reverse(ref_plain, 14, 14), what='no mask')
        tokens = torch.roll(window_reverse(x, 14, 14), shifts=(3, 3), dims=(1, 2))
        with torch.no_grad():
            out = m(tokens.reshape(2, 196, 96).to(dev), (14, 14), 3, variant)
        expect = torch.roll(window_reverse(ref_mask, 14, 14), shifts=(3, 3), dims=(1, 2))
        close(out.view(2, 14, 14, 96), expect, what='explicit shift mask')


def test_swin_block_fixture(dev, golden):
    """`swin_block.npz`: SwinBlock(96, 3 heads, shift) on an 11 x 13 map (padded to 14 x 14 inside the attention)."""
    from gedepth_amd.depth.models.backbones.depthformer_swin import SwinBlock
    g = golden('swin_block')
    m = SwinBlock(embed_dims=96, num_heads=3, feedforward_channels=384, shift=True, drop_path_rate=0.).eval()
    load_filled(m, 'swin_block')
    m = m.to(dev)
    m.attn.kernel_variant = 1
    with torch.no_grad():
        out = m(T(g['x']).to(dev), (11, 13))
    close(out, g['out'], what='swin block')


def test_patch_merging_and_embed_fixtures(dev, golden):
    """`patch_merging.npz` (5 x 7: odd height and width -> zero row / column before the 2 x 2 unfold) and `patch_embed.npz`
    (18 x 30 input: padded to the patch size)."""
    from gedepth_amd.depth.models.backbones.depthformer_swin import PatchMerging
    from gedepth_amd.depth.models.utils import PatchEmbedSwin
    g = golden('patch_merging')
    m = PatchMerging(in_channels=96, out_channels=192).eval()
    load_filled(m, 'patch_merging')
    m = m.to(dev)
    with torch.no_grad():
        out, hw = m(T(g['x']).to(dev), (5, 7))
    assert tuple(hw) == tuple(int(v) for v in g['hw'])
    close(out, g['out'], what='patch merging')
    g = golden('patch_embed')
    m = PatchEmbedSwin(in_channels=4, embed_dims=96, conv_type='Conv2d', kernel_size=4, stride=4, pad_to_patch_size=True,
                       norm_cfg=dict(type='LN')).eval()
    load_filled(m, 'patch_embed')
    m = m.to(dev)
    with torch.no_grad():
        out, hw = m(T(g['x']).to(dev))
    assert tuple(hw) == tuple(int(v) for v in g['hw'])
    close(out, g['out'], what='patch embed')


@pytest.mark.parametrize('layout', ['nchw', 'nhwc'])
def test_hahi_neck_fixture(dev, golden, layout):
    """`hahi.npz`: the whole HAHI neck (Swin-T widths, 32 x 48 ... 2 x 3 maps) in eval-BN and train-BN mode: lateral / projection
    convs, both deformable attentions (window kernels on the query grids), token <-> map glue, fusion convs."""
    from gedepth_amd.depth.models.necks.hahi import HAHIHeteroNeck
    g = golden('hahi')
    chans = [64, 96, 192, 384, 768]
    m = HAHIHeteroNeck(in_channels=chans, out_channels=chans, embedding_dim=512, scales=[1] * 5,
                       positional_encoding=dict(type='SinePositionalEncoding', num_feats=256))
    load_filled(m, 'hahi')
    m.multi_att.dropout.p = 0.0
    m.self_attn.dropout.p = 0.0
    m = m.to(dev)
    feats = [T(g[f'in{i}']).to(dev) for i in range(5)]
    if layout == 'nhwc':
        from gedepth_amd.depth.models.utils import to_channels_last
        to_channels_last(m)
        feats = [f.contiguous(memory_format=torch.channels_last) for f in feats]
    for mode in ('eval', 'train'):
        m.train(mode == 'train')
        with torch.no_grad():
            outs = m(feats)
        for i, o in enumerate(outs):
            close(o, g[f'{mode}_out{i}'], rtol=2e-4, atol=2e-5, what=f'hahi {mode} out{i} ({layout})')


def test_pe_necks_fixture(dev, golden):
    """`pe_necks.npz`: LightPEMASKNeck / DynamicPENeckSOFT (Swin-L widths): five 3 x 3 convs + bias, align_corners up-sampling, sum,
    final conv (+ sigmoid)."""
    from gedepth_amd.depth.models.necks.pe_necks import DynamicPENeckSOFT, LightPEMASKNeck
    g = golden('pe_necks')
    feats = [T(g[f'in{i}']).to(dev) for i in range(5)]
    m1 = LightPEMASKNeck().eval()
    load_filled(m1, 'pe_mask_neck')
    m2 = DynamicPENeckSOFT().eval()
    load_filled(m2, 'dynamic_pe_neck')
    with torch.no_grad():
        y, _ = m1.to(dev)(feats)
        logits = m2.to(dev)(feats)
    close(y, g['y'], what='pe mask neck')
    close(logits, g['logits'], rtol=2e-4, atol=2e-5, what='dynamic pe neck')


def test_sine_positional_encoding_fixture(dev, golden):
    from gedepth_amd.depth.utils.position_encoding import SinePositionalEncoding
    pe = SinePositionalEncoding(num_feats=256)
    out = pe(torch.zeros(1, 5, 7, dtype=torch.bool, device=dev))
    close(out, golden('sine_pos')['out'], what='sine positional encoding')


def test_log_images_fixture_and_train_step(dev, golden):
    """`log_images.npz` (the reference's own DepthBaseDecodeHead.log_images, decode_head.py:628-648): the product method on DEVICE tensors gives the
    same uint8 RGB image and max-normalised depth maps, both channel-order cases, bit for bit.  Then through the model with
    ``decode_head.log_images=True``: train_step splits the three ``img_*`` entries into ``log_imgs`` (reference depther/base.py:141-148), they
    describe sample 0 of the batch, and neither the loss nor ``log_vars`` sees them."""
    import os
    from gedepth_amd.depth.datasets.synthetic import synthetic_batch
    from gedepth_amd.depth.models import build_depther
    from gedepth_amd.depth.models.decode_heads.decode_head import DepthBaseDecodeHead
    from gedepth_amd.mmrt.config import Config
    from oracle import gedepth_oracle as O
    g = golden('log_images')
    img, pred, gt = T(g['img']).to(dev), T(g['depth_pred']).to(dev), T(g['depth_gt']).to(dev)
    for tag, to_rgb in (('rgb', True), ('bgr', False)):
        r = DepthBaseDecodeHead.log_images(None, img, pred, gt, dict(img_norm_cfg=dict(mean=g['mean'], std=g['std'], to_rgb=to_rgb)))
        assert r['img_rgb'].dtype == np.uint8 and np.array_equal(r['img_rgb'], g[f'img_rgb_{tag}'])
        assert not r['img_depth_pred'].is_cuda and np.array_equal(r['img_depth_pred'].numpy(), g[f'img_depth_pred_{tag}'])
        assert np.array_equal(r['img_depth_gt'].numpy(), g[f'img_depth_gt_{tag}'])

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, 'configs', 'depthformer', 'depthformer_swint_v.py'))
    cfg.model.pretrained = None
    cfg.model.backbone.drop_path_rate = 0.0
    outs = {}
    for flag in (False, True):
        cfg.model.decode_head.log_images = flag
        torch.manual_seed(0)
        model = build_depther(cfg.model)
        model.neck.multi_att.dropout.p = 0.0
        model.neck.self_attn.dropout.p = 0.0
        load_filled(model, 'logimg')
        model = model.to(dev).train()
        batch = synthetic_batch(2, 64, 96, seed=5, valid_fraction=0.3)
        batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
        outs[flag] = (model.train_step(batch, None), batch)
    (plain, _), (logged, batch) = outs[False], outs[True]
    assert plain['log_imgs'] == {} and sorted(logged['log_imgs']) == ['decode.img_depth_gt', 'decode.img_depth_pred', 'decode.img_rgb']      # add_prefix(..., 'decode'), encoder_decoder.py
    assert not any('img' in k for k in logged['log_vars'].keys())
    assert abs(float(plain['loss']) - float(logged['loss'])) <= 1e-6 * abs(float(plain['loss']))
    norm = batch['img_metas'][0]['img_norm_cfg']
    want = O.log_images(batch['img'][0].cpu(), torch.ones(1, 1, 1), batch['depth_gt'][0].cpu(), norm['mean'], norm['std'], norm['to_rgb'])
    li = {k.split('.', 1)[1]: v for k, v in logged['log_imgs'].items()}
    assert np.array_equal(li['img_rgb'], want['img_rgb']) and li['img_rgb'].shape == (3, 64, 96)
    assert torch.equal(li['img_depth_gt'], want['img_depth_gt'])
    p = li['img_depth_pred']
    assert p.shape[-2:] == (64, 96) or p.shape[-2:] == (32, 48), p.shape        # the head's own resolution (the reference logs it before the loss resize)
    assert float(p.max()) == 1.0 and float(p.min()) > 0
