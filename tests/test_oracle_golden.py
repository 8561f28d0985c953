"""Pin the CPU oracle (oracle/gedepth_oracle.py) against fixtures produced by the reference itself
(tests/golden/make_golden.py) and against the closed-form known answers of SURVEY.md Appendix E."""
import json

import numpy as np
import pytest
import torch

from oracle import gedepth_oracle as O
from oracle.fill import fill_state_dict

RTOL, ATOL = 1e-4, 1e-5      # fp32 tolerance stated by BASELINE.json north_star (1e-4 rel)


def T(a):
    return torch.from_numpy(np.asarray(a))


def weights(g, salt, key='spec', prefix=''):
    spec = json.loads(str(g[key]))
    P = fill_state_dict([(n, s) for n, s in spec], salt)
    return {prefix + k: v for k, v in P.items()}


def close(a, b, rtol=RTOL, atol=ATOL):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f'max abs err {err.max():.3e}, max rel {(err / (b.abs() + 1e-12)).max():.3e}'


def test_relative_position_index(golden):
    g = golden('window_msa')
    idx = O.relative_position_index()
    assert torch.equal(idx, T(g['rel_index']))
    i = torch.arange(49) // 7
    j = torch.arange(49) % 7
    analytic = (i[:, None] - i[None, :] + 6) * 13 + (j[:, None] - j[None, :] + 6)
    assert torch.equal(idx, analytic)


def test_window_msa(golden):
    g = golden('window_msa')
    P = weights(g, 'window_msa', prefix='w.')
    close(O.window_msa(T(g['x']), None, P, 'w', 3), g['out_nomask'])
    close(O.window_msa(T(g['x']), T(g['mask']), P, 'w', 3), g['out_mask'])


@pytest.mark.parametrize('tag,hw', [('a', (11, 35)), ('b', (10, 9))])
@pytest.mark.parametrize('shift', [0, 3])
def test_shift_window_msa(golden, tag, hw, shift):
    g = golden('shift_window_msa')
    P = weights(g, 'shift_window_msa', prefix='a.')
    close(O.shift_window_msa(T(g[f'x_{tag}{shift}']), hw, P, 'a', 3, shift), g[f'out_{tag}{shift}'])


def test_analytic_shift_mask():
    """region id r(h)=0 if h<Hp-7, 1 if h<Hp-3 else 2; id=3 r(h)+r(w)."""
    for Hp, Wp in [(14, 35), (21, 28), (7, 7)]:
        ref = O.shift_mask(Hp, Wp)
        r = lambda t, n: (t >= n - 7).long() + (t >= n - 3).long()
        hh, ww = torch.meshgrid(torch.arange(Hp), torch.arange(Wp), indexing='ij')
        ids = (3 * r(hh, Hp) + r(ww, Wp)).float().view(1, Hp, Wp, 1)
        mw = O.window_partition(ids).view(-1, 49)
        m = (mw[:, None, :] != mw[:, :, None]).float() * -100.0
        assert torch.equal(m, ref)


def test_swin_block_merging_embed(golden):
    g = golden('swin_block')
    close(O.swin_block(T(g['x']), (11, 13), weights(g, 'swin_block', prefix='b.'), 'b', 3, 3), g['out'])
    g = golden('patch_merging')
    out, hw = O.patch_merging(T(g['x']), (5, 7), weights(g, 'patch_merging', prefix='d.'), 'd')
    close(out, g['out'])
    assert tuple(hw) == tuple(g['hw'])


def test_msda(golden):
    g = golden('msda_core')
    shapes = [tuple(int(v) for v in s) for s in g['shapes']]
    close(O.msda_core(T(g['value']), shapes, T(g['loc']), T(g['aw'])), g['out'])
    g = golden('msda_module')
    P = weights(g, 'msda_module', prefix='m.')
    close(O.msda_module(T(g['q']), T(g['v']), T(g['qp']), T(g['ref']), shapes, P, 'm'), g['out'])


def test_msda_core_vs_transformers(golden):
    """Independent pin of the un-vendored mmcv sampling core: HF transformers' own pure-PyTorch deformable
    attention, written independently of this repository, must reproduce the fixture bit for bit.  A hard requirement (no
    skip): transformers ships in the build image, and this is the only check of `msda_core.npz` that does not go through
    the oracle that wrote it."""
    from transformers.models.deformable_detr import modeling_deformable_detr as tr
    cls = tr.MultiScaleDeformableAttention
    g = golden('msda_core')
    shapes = [tuple(int(v) for v in s) for s in g['shapes']]
    out = cls()(T(g['value']), torch.as_tensor(shapes), shapes, None, T(g['loc']), T(g['aw']), 64)
    assert torch.equal(out, T(g['out']))
    # and the oracle's function against the same independent implementation on a second, ragged geometry with out-of-range points
    gen = torch.Generator().manual_seed(5)
    shapes2 = [(9, 13), (5, 7), (3, 4), (1, 2)]
    nv = sum(h * w for h, w in shapes2)
    value = torch.randn(2, nv, 8, 64, generator=gen)
    loc = torch.rand(2, 37, 8, 4, 8, 2, generator=gen) * 1.4 - 0.2
    aw = torch.rand(2, 37, 8, 32, generator=gen).softmax(-1).view(2, 37, 8, 4, 8)
    assert torch.equal(cls()(value, torch.as_tensor(shapes2), shapes2, None, loc, aw, 64), O.msda_core(value, shapes2, loc, aw))


def test_sine_pos(golden):
    close(O.sine_positional_encoding(1, 5, 7), golden('sine_pos')['out'])


def test_hahi(golden):
    g = golden('hahi')
    P = weights(g, 'hahi', prefix='neck.')
    feats = [T(g[f'in{i}']) for i in range(5)]
    for mode, train in (('eval', False), ('train', True)):
        outs = O.hahi_neck(feats, P, train_bn=train)
        for i, o in enumerate(outs):
            close(o, g[f'{mode}_out{i}'], rtol=2e-4, atol=2e-5)


def test_pe_necks(golden):
    g = golden('pe_necks')
    feats = [T(g[f'in{i}']) for i in range(5)]
    close(O.pe_mask_neck(feats, weights(g, 'pe_mask_neck', 'spec_mask', 'pe_mask_neck.')), g['y'])
    close(O.dynamic_pe_neck(feats, weights(g, 'dynamic_pe_neck', 'spec_dyn', 'dynamic_pe_neck.')), g['logits'])


def test_dynamic_pe(golden):
    g = golden('dynamic_pe')
    img = T(g['img'])
    pe_mask, logits, m = O.dynamic_pe(T(g['logits_lr']), T(g['y']), img[:, 4])
    close(logits, g['logits_hr'])
    close(pe_mask, g['pe_mask'], atol=1e-4)
    assert set(np.unique(m.numpy())) <= {0.0, 1.0}
    pe_mask_h, _, _ = O.dynamic_pe(T(g['logits_lr']), T(g['y']), img[:, 4], height=T(g['heights']))
    close(pe_mask_h, g['pe_mask_h'], atol=1e-4)
    close(O.vanilla_pe(T(g['y']), img[:, 3]), g['vanilla'])


def test_dynamic_pe_integer_mask_vs_reference(golden):
    """The 0/1 pe_offset_mask (encoder_decoder.py:97-100) on EVERY pixel, against the mask read off the reference's own
    dynamic_pe (tests/golden/make_golden_mask.py): 2x24x40 and 2x88x280, default and per-sample camera heights."""
    g = golden('dynamic_pe_mask')
    for tag in 'ab':
        pe_raw, lg = T(g[f'{tag}_pe_raw']), T(g[f'{tag}_logits_lr'])
        ones = torch.ones(pe_raw.shape[0], 1, *pe_raw.shape[1:])
        for sfx, h in (('', 1.65), ('_h', T(g[f'{tag}_heights']))):
            prod, _, m = O.dynamic_pe(lg, ones, pe_raw, h)
            assert torch.equal(m[:, 0].to(torch.uint8), T(g[f'{tag}_mask{sfx}'])[:, 0])
            assert torch.equal(prod, T(g[f'{tag}_offset_masked{sfx}']))          # same torch CPU ops: bit-identical


def test_log_images_vs_reference_method(golden):
    """oracle.log_images and the product's DepthBaseDecodeHead.log_images (host arithmetic: runs on CPU tensors) against the arrays the
    reference's own method returned (tests/golden/make_golden_logimg.py), both channel-order cases: uint8 image bit-exact, depth maps bit-exact."""
    from gedepth_amd.depth.models.decode_heads.decode_head import DepthBaseDecodeHead
    g = golden('log_images')
    img, pred, gt = T(g['img']), T(g['depth_pred']), T(g['depth_gt'])
    for tag, to_rgb in (('rgb', True), ('bgr', False)):
        got = O.log_images(img, pred, gt, g['mean'], g['std'], to_rgb)
        meta = dict(img_norm_cfg=dict(mean=g['mean'], std=g['std'], to_rgb=to_rgb))
        prod = DepthBaseDecodeHead.log_images(None, img, pred, gt, meta)
        for r in (got, prod):
            assert r['img_rgb'].dtype == np.uint8 and np.array_equal(r['img_rgb'], g[f'img_rgb_{tag}'])
            assert np.array_equal(np.asarray(r['img_depth_pred']), g[f'img_depth_pred_{tag}'])
            assert np.array_equal(np.asarray(r['img_depth_gt']), g[f'img_depth_gt_{tag}'])
    assert 0 < (g['img_rgb_rgb'] == 0).mean() < 0.5 and 0 < (g['img_rgb_rgb'] == 255).mean() < 0.5          # both clip edges are exercised


def test_known_answers(golden):
    g = golden('known_answers')
    close(O.sigloss(T(g['sig_pred']), T(g['sig_gt'])), g['sig'])
    assert abs(float(g['sig']) - 0.28163391) < 1e-6
    close(O.ce_loss(T(g['ce_logits']), T(g['ce_target'])), g['ce'])
    mt = O.metrics_calculate(np.array([1.5, 3, 8, 12, 25, 60.]), np.array([1., 4, 8, 10, 20, 79.]))
    np.testing.assert_allclose(mt, g['metrics'], rtol=1e-12)
    np.testing.assert_allclose(mt[3], 0.225, rtol=1e-9)
    np.testing.assert_allclose(mt[4], 8.0751677, rtol=1e-7)


def test_product_metrics_vs_reference_known_answers(golden):
    """The PRODUCT's depth/core/evaluation.py (not the oracle) against the 9-vector the reference's
    core/evaluation/metrics.py:8-33 ``calculate`` returned (tests/golden/make_golden.py section 9), and the masked entry
    point against the oracle on random maps with out-of-range ground truth."""
    from gedepth_amd.depth.core import evaluation as E
    g = golden('known_answers')
    gt, pred = np.array([1.5, 3, 8, 12, 25, 60.]), np.array([1., 4, 8, 10, 20, 79.])
    np.testing.assert_allclose(E.calculate(gt, pred), g['metrics'], rtol=1e-12)
    assert E.METRIC_NAMES[3] == 'abs_rel' and abs(E.calculate(gt, pred)[3] - 0.225) < 1e-12
    rs = np.random.RandomState(1)
    gt = np.where(rs.rand(40, 60) < 0.4, rs.rand(40, 60) * 100, 0.0)
    pred = rs.rand(40, 60) * 80 + 0.5
    np.testing.assert_allclose(E.metrics(gt, pred), O.metrics(gt, pred), rtol=1e-12)
    assert all(np.isnan(v) for v in E.metrics(np.zeros((3, 3)), np.ones((3, 3))))               # empty selection


def test_ground_plane_known_answers():
    """SURVEY Appendix E (values measured on the reference formula, tools/preprocess_data_kitti.py:47-53)."""
    P2 = np.array([[7.215377e+02, 0.0, 6.095593e+02, 4.485728e+01],
                   [0.0, 7.215377e+02, 1.728540e+02, 2.163791e-01],
                   [0.0, 0.0, 1.0, 2.745884e-03]])
    Tr = np.array([[0., -1, 0, 0], [0, 0, -1, -0.08], [1, 0, 0, -0.27], [0, 0, 0, 1]])
    pe, _, _ = O.ground_plane(P2, np.eye(3), Tr, 375, 1242)
    np.testing.assert_allclose(pe[374, 621], 5.630516794, rtol=1e-8)
    np.testing.assert_allclose(pe[200, 100], 41.720913986, rtol=1e-8)
    np.testing.assert_allclose(pe[100, 621], -15.545555921, rtol=1e-8)
    k = O.slope_class(np.array([[10., 0], [30, 5]]), np.array([[11, 20], [25, 5.2]], dtype=np.float32))
    assert k.tolist() == [[1, 255], [-1, 1]]


def test_ground_plane_and_slope_classes_vs_reference_scripts(golden):
    """Bit-for-bit against arrays written by the reference's own tools/preprocess_data_kitti.py:29-92 and
    preprocess_data_ddad.py:29-82, run unmodified on a toy calibration tree (tests/golden/make_golden_ground.py)."""
    g = golden('ground_plane')
    for d in range(2):
        ref = g[f'kitti{d}_pe']
        pe, row2, num = O.ground_plane(g[f'kitti{d}_P2'], g[f'kitti{d}_R0'], g[f'kitti{d}_Tr'], *ref.shape)
        assert pe.dtype == np.float64 and np.array_equal(pe, ref)
        assert (ref < 0).any() and (ref > 0).any()              # rows above and below the horizon
        for f in range(2):
            k = O.slope_class(g[f'kitti{d}_gt16_{f}'] / 256, pe.astype(np.float32), 1.65, 'round')
            assert np.array_equal(k, g[f'kitti{d}_k_{f}'])
            assert set(np.unique(k)) == set(range(-5, 6)) | {255}
    for i, cam in enumerate(('CAMERA_01', 'CAMERA_05', 'CAMERA_06', 'CAMERA_09')):
        ref = g[f'ddad{i}_pe']
        pe, _, _ = O.ground_plane_ddad(g[f'ddad{i}_K'], g[f'ddad{i}_pose'], g['ddad_lidar_pose'], *ref.shape)
        assert np.array_equal(pe, ref)
        k = O.slope_class_ddad(g[f'ddad{i}_gt'], pe, O.DDAD_CAMERA_HEIGHTS[cam])
        assert k.dtype == np.int64 and np.array_equal(k, g[f'ddad{i}_k'])


@pytest.mark.parametrize('tag,arch,adaptive', [('e2e_T_V', O.SWIN_T, False), ('e2e_T_A', O.SWIN_T, True),
                                               ('e2e_L_A', O.SWIN_L, True)])
def test_e2e(golden, tag, arch, adaptive):
    g = golden(tag)
    P = weights(g, 'e2e')
    for v in P.values():
        v.requires_grad_(v.is_floating_point())
    for k in P:
        if k.endswith('running_mean') or k.endswith('running_var'):
            P[k].requires_grad_(False)
    cfg = dict(arch, adaptive=adaptive)
    img, gt, kgt = T(g['img']), T(g['depth_gt']), T(g['pe_k_gt'])
    with torch.no_grad():
        close(O.encode_decode(img, P, cfg), g['depth_eval'], rtol=3e-4, atol=1e-4)
    losses, _ = O.forward_train(img, gt, kgt, P, cfg, train_bn=True)
    loss, log_vars = O.parse_losses(losses)
    names = json.loads(str(g['loss_names']))
    for n, v in zip(names, g['loss_values']):
        assert abs(log_vars[n] - v) <= 2e-4 * abs(v) + 1e-5, (n, log_vars[n], v)
    loss.backward()
    for k in g.files:
        if k.startswith('grad::'):
            gr = P[k[6:]].grad.flatten()
            gr = gr[::max(1, gr.numel() // 50000)]
            ref = T(g[k])
            scale = ref.abs().max().item() + 1e-12
            assert (gr - ref).abs().max().item() <= 2e-3 * scale, (k, (gr - ref).abs().max().item(), scale)
    total = torch.sqrt(sum((v.grad ** 2).sum() for v in P.values() if v.grad is not None))
    assert abs(total.item() - float(g['grad_norm_total'])) <= 1e-3 * float(g['grad_norm_total'])


@pytest.mark.parametrize('tag,adaptive', [('T_V', False), ('T_A', True)])
def test_test_path_vs_reference_methods(golden, tag, adaptive):
    """The oracle's restatement of the test path (``inference`` / ``aug_test``, encoder_decoder.py:196-274) against arrays written
    by the reference's OWN ``simple_test`` and ``aug_test`` (tests/golden/make_golden.py: plain view, horizontal- and vertical-flip
    pairs, flip-back inside ``inference``) on the e2e toy model — the last oracle function that had no reference-generated pin."""
    g, ge = golden(f'test_path_{tag}'), golden(f'e2e_{tag}')
    P = weights(ge, 'e2e')
    cfg = dict(O.SWIN_T, adaptive=adaptive)
    img = T(g['img'])
    meta = dict(ori_shape=(64, 96, 3), flip=False, flip_direction='horizontal')
    with torch.no_grad():
        simple = O.inference(img, [meta] * 2, P, cfg)
        close(simple, g['simple'], rtol=3e-4, atol=1e-4)
        for direction, dim, key in (('horizontal', 3, 'aug_h'), ('vertical', 2, 'aug_v')):
            flipped = dict(meta, flip=True, flip_direction=direction)
            aug = O.aug_test([img, img.flip(dim)], [[meta] * 2, [flipped] * 2], P, cfg)
            close(aug, g[key], rtol=3e-4, atol=1e-4)
    # the flip augmentation is not a no-op on this model: the two views disagree by far more than the tolerance
    assert np.abs(g['aug_h'] - g['simple']).max() > 1e-2


@pytest.mark.parametrize('tag', ['e2e_T_V', 'e2e_L_A'])
def test_reference_fixture_vs_float64_oracle(golden, tag):
    """How far the REFERENCE's own fp32 (CPU) results are from the same algorithm in float64 (tests/f64ref.py): the
    yardstick the GPU parity test uses.  Losses agree to 1e-7 and the eval depth to < 1e-4 rel, but a few backbone
    gradients (patch embed, relative-position tables, q/k/v biases of stage 0 — sums that cancel to rounding level) are
    only good to ~1e-3 in ANY fp32 evaluation; the bounds below are those measured values with 2x head-room."""
    import f64ref
    r = f64ref.run(tag)
    g = r['fixture']
    for n, v in zip(json.loads(str(g['loss_names'])), g['loss_values']):
        assert abs(v - r['log_vars'][n]) <= 2e-7 * abs(v), (n, v, r['log_vars'][n])
    ref = T(g['depth_eval']).double()
    rel = ((ref - r['depth_eval']).abs() / r['depth_eval'].abs().clamp_min(1e-3)).max().item()
    assert rel <= 1e-4, rel
    worst = {}
    for k in g.files:
        if k.startswith('grad::'):
            worst[k[6:]] = f64ref.l2rel(T(g[k]), f64ref.sample(r['grads'][k[6:]]))
    soft = ('patch_embed', 'relative_position_bias_table', 'qkv.bias')
    for k, e in worst.items():
        assert e <= (2e-3 if any(s in k for s in soft) else 1e-4), (k, e)
    if tag == 'e2e_L_A':
        assert max(worst.values()) >= 1e-4          # the point: the reference itself is not within 1e-4 of the truth


def test_fp32_gradient_gap_is_kink_decisions():
    """Why two correct fp32 evaluations of this network differ by up to ~5e-3 on some gradients while every layer is good
    to 1e-6: ReLU / LeakyReLU (and bilinear sampling) have kinks, and the 64 x 96 Swin-L fixture has activations within
    2.5e-6 of zero.  The fp32 oracle (= the reference's arithmetic) flips a handful of its 4.1 M activation decisions against
    float64; with exactly those decisions imposed on the float64 evaluation (oracle.KINK_FORCE) the SAME fp32 gradients agree
    with float64 to 1e-4 on every tensor.  tests/test_model_gpu.py applies the identical procedure to the HIP path."""
    import f64ref
    tag = 'e2e_L_A'
    r32 = f64ref.run(tag, dtype=torch.float32, record=True)
    forced = f64ref.run(tag, force=r32['decisions'])
    st = forced['force_stats']
    assert st['act_seen'] > 4e6 and st['floor_seen'] > 2e6
    assert st['act_flipped'] <= 12 and st['max_flipped_preact'] <= 1e-5, st          # a handful, all at rounding level
    assert st['floor_flipped'] <= 4 and st['max_shift_px'] <= 1e-4, st
    total = torch.sqrt(sum((v.double() ** 2).sum() for v in forced['grads'].values())).item()
    worst = 0.0
    for k, v in forced['grads'].items():
        if v.norm().item() < 1e-10 * total:
            continue
        worst = max(worst, f64ref.l2rel(r32['grads'][k], v))
    assert worst <= 1e-4, worst
