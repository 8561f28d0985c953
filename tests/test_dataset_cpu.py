"""KITTI dataset / pipeline / evaluation protocol on a toy tree; host-side only."""
import os
import random
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from gedepth_amd.depth.apis.test import single_gpu_test                      # noqa: E402
from gedepth_amd.depth.datasets import KITTIDataset, build_dataloader, build_dataset  # noqa: E402
from gedepth_amd.depth.datasets.pipelines import imageops as I              # noqa: E402
from gedepth_amd.mmrt.config import Config                                   # noqa: E402
from toy_kitti import make_toy_kitti                                         # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def toy(tmp_path_factory):
    root = str(tmp_path_factory.mktemp('kitti'))
    return root, make_toy_kitti(root)


def _cfg(root, split):
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', 'depthformer_swint_v.py'))
    for part in ('train', 'val', 'test'):
        cfg.data[part].data_root = root
        cfg.data[part].split = split
    return cfg


def test_image_ops_follow_the_documented_sampling_rules():
    g = np.random.default_rng(0)
    img = g.random((13, 17, 5)).astype(np.float32)
    t = torch.from_numpy(img).permute(2, 0, 1)[None]
    for size in ((31, 22), (9, 7), (17, 13)):                                            # (w, h)
        ref = F.interpolate(t, size=(size[1], size[0]), mode='bilinear', align_corners=False)[0].permute(1, 2, 0).numpy()
        assert np.allclose(I.imresize(img, size), ref, atol=1e-6)
        refn = F.interpolate(t, size=(size[1], size[0]), mode='nearest')[0].permute(1, 2, 0).numpy()
        assert np.array_equal(I.imresize(img, size, interpolation='nearest'), refn)
        assert np.allclose(I._imresize_numpy(img, size), ref, atol=1e-6)                 # the rule, spelled out
    for angle in (2.5, -17.0, 90.0):                                                      # grid_sample path == numpy rule
        for mode, border in (('bilinear', 0), ('nearest', 255)):
            a = I.imrotate(img, angle, border_value=border, interpolation=mode)
            b = I._imrotate_numpy(img, angle, border_value=border, interpolation=mode)
            assert (np.abs(a - b) > 1e-3).mean() <= (0.0 if mode == 'bilinear' else 0.02)  # nearest: ties may round either way
    assert I.rescale_size((1216, 352), 0.5) == (608, 176)
    assert I.rescale_size((1216, 352), (2000, 500)) == (int(1216 * 500 / 352 + 0.5), 500)
    sq = g.random((9, 9)).astype(np.float32)
    assert np.allclose(I.imrotate(sq, 0.0), sq)
    assert np.allclose(I.imrotate(sq, 90.0, interpolation='nearest'), np.rot90(sq, -1))          # clockwise
    assert np.allclose(I.imrotate(sq, 90.0), np.rot90(sq, -1), atol=1e-5)
    wide = np.ones((5, 11), np.float32)
    r = I.imrotate(wide, 45.0, border_value=255, interpolation='nearest')
    assert r.shape == wide.shape and set(np.unique(r)) == {1.0, 255.0}
    bgr = g.integers(0, 256, (4, 6, 3), dtype=np.uint8)
    mean, std = np.array([123.675, 116.28, 103.53], np.float32), np.array([58.395, 57.12, 57.375], np.float32)
    assert np.allclose(I.imnormalize(bgr, mean, std, True), (bgr[..., ::-1].astype(np.float32) - mean) / std, atol=1e-6)
    assert np.array_equal(I.imflip(bgr), bgr[:, ::-1])


def test_dataset_index_and_train_sample(toy):
    root, split = toy
    cfg = _cfg(root, split)
    ds = build_dataset(cfg.data.train)
    assert isinstance(ds, KITTIDataset) and len(ds) == 4 and ds.invalid_depth_num == 1
    names = [i['filename'] for i in ds.img_infos]
    assert names == sorted(names)
    random.seed(3); np.random.seed(3)
    s = ds[0]
    assert set(s) == {'img_metas', 'img', 'depth_gt', 'pe_ori_point', 'pe_k_gt'}
    assert s['img'].shape == (5, 352, 704) and s['img'].dtype == torch.float32
    assert s['depth_gt'].shape == (1, 352, 704) and s['pe_k_gt'].shape == (352, 704)
    k = np.unique(s['pe_k_gt'].numpy())
    assert set(k.tolist()) <= set(range(11)) | {255}
    assert s['img_metas']['cam_intrinsic'][0][0] == pytest.approx(721.5377)
    pe, raw = s['img'][3].numpy(), s['img'][4].numpy()
    assert pe.min() >= 0 and pe.max() <= 1.0 + 1e-6                          # filtered ground depth / 200
    assert float(s['pe_ori_point']) == pytest.approx(1.65 * 721.5377 / (374 - 172.854), rel=1e-6)
    assert raw.min() < 0 or raw.max() > 1                                    # the raw channel is not rescaled


def test_train_pipeline_geometry_is_consistent_across_fields(toy):
    """Deterministic variant (no rotate / scale 1 / no colour): KB crop + flip + random crop keep RGB, ground depth,
    LiDAR depth and slope classes aligned with a direct numpy computation."""
    root, split = toy
    cfg = _cfg(root, split)
    pipe = [dict(t) for t in cfg.data.train.pipeline]
    for t in pipe:
        if t['type'] == 'Resize':
            t['ratio_range'] = (1.0, 1.0)
        if t['type'] in ('RandomRotate', 'ColorAug'):
            t['prob'] = 0.0
        if t['type'] == 'RandomFlip':
            t['prob'] = 1.0
    tr = dict(cfg.data.train); tr['pipeline'] = pipe
    ds = build_dataset(tr)
    np.random.seed(11); random.seed(11)
    s = ds[1]
    np.random.seed(11)
    np.random.rand(); np.random.uniform(-2.5, 2.5)        # RandomRotate draws; Resize draws one more
    from PIL import Image
    info = ds.img_infos[1]
    rgb = np.asarray(Image.open(os.path.join(ds.img_dir, info['filename'])).convert('RGB'))
    gt = np.asarray(Image.open(os.path.join(ds.ann_dir, info['ann']['depth_map'])), dtype=np.float32) / 256
    top, left = 375 - 352, int((1242 - 1216) / 2)
    rgb, gt = rgb[top:, left:left + 1216][:, ::-1], gt[top:, left:left + 1216][:, ::-1]
    # locate the random crop from the depth map, then check every field against it
    d = s['depth_gt'][0].numpy()
    hits = [x for x in range(1216 - 704 + 1) if np.array_equal(gt[:, x:x + 704], d)]
    assert len(hits) == 1
    x = hits[0]
    mean, std = np.array([123.675, 116.28, 103.53], np.float32), np.array([58.395, 57.12, 57.375], np.float32)
    assert np.allclose(s['img'][:3].permute(1, 2, 0).numpy(), (rgb[:, x:x + 704].astype(np.float32) - mean) / std, atol=1e-5)
    pe = np.load(os.path.join(root, 'input', '2011_09_26', 'pe', 'pe_165.npy')).astype(np.float32)[top:, left:left + 1216][:, ::-1]
    assert np.allclose(s['img'][4].numpy(), pe[:, x:x + 704])
    assert np.allclose(s['img'][3].numpy(), np.where((pe > 0) & (pe <= 200), pe, 0)[:, x:x + 704] / 200, atol=1e-7)
    assert s['img_metas']['flip'] is True and s['img_metas']['flip_direction'] == 'horizontal'


def test_test_pipeline_tta_and_collate(toy):
    root, split = toy
    cfg = _cfg(root, split)
    ds = build_dataset(cfg.data.test, dict(test_mode=True))
    s = ds[0]
    assert isinstance(s['img'], list) and len(s['img']) == 2 and s['img'][0].shape == (5, 352, 1216)
    assert torch.equal(s['img'][1], s['img'][0].flip(2))
    assert [m['flip'] for m in s['img_metas']] == [False, True]
    loader = build_dataloader(ds, 2, 0, dist=False, shuffle=False, pin_memory=False)
    batch = next(iter(loader))
    assert len(batch['img']) == 2 and batch['img'][0].shape == (2, 5, 352, 1216)
    assert len(batch['img_metas']) == 2 and len(batch['img_metas'][0]) == 2 and batch['pe_ori_point'][0].shape == (2,)


class _Oracle(torch.nn.Module):
    """Stands in for a depther in the evaluation loop: returns the KB-cropped ground truth (+ a known offset)."""

    def __init__(self, ds, offset):
        super().__init__()
        self.ds, self.offset, self.calls = ds, offset, 0
        self.p = torch.nn.Parameter(torch.zeros(1))

    def forward(self, img, img_metas, return_loss=True, **kw):
        assert return_loss is False and isinstance(img, list) and len(img) == 2
        out = []
        for m in img_metas[0]:
            idx = [i['filename'] for i in self.ds.img_infos].index(m['ori_filename'])
            gt = self.ds.eval_kb_crop(self.ds._gt(idx))
            out.append(np.where(gt > 0, gt * (1 + self.offset), 1.0).astype(np.float32))
        self.calls += 1
        return out


def test_eval_protocol(toy):
    root, split = toy
    cfg = _cfg(root, split)
    ds = build_dataset(cfg.data.test, dict(test_mode=True))
    gt = ds.eval_kb_crop(ds._gt(0))
    assert gt.shape == (1, 352, 1216)
    mask = ds.eval_mask(gt)
    rows, cols = np.where(mask[0])
    assert rows.min() >= int(0.40810811 * 352) and rows.max() < int(0.99189189 * 352)
    assert cols.min() >= int(0.03594771 * 1216) and cols.max() < int(0.96405229 * 1216)
    loader = build_dataloader(ds, 1, 0, dist=False, shuffle=False, pin_memory=False)
    res = single_gpu_test(_Oracle(ds, 0.0), loader, pre_eval=True, device='cpu')
    assert len(res) == 4
    summary = ds.evaluate(res)
    assert summary['abs_rel'] == pytest.approx(0.0, abs=1e-7) and summary['a1'] == pytest.approx(1.0)
    res = single_gpu_test(_Oracle(ds, 0.1), loader, pre_eval=True, device='cpu')
    summary = ds.evaluate(res)
    assert summary['abs_rel'] == pytest.approx(0.1, rel=1e-4) and summary['sq_rel'] > 0
    preds = single_gpu_test(_Oracle(ds, 0.1), loader, pre_eval=False, device='cpu')
    assert ds.evaluate(preds)['abs_rel'] == pytest.approx(0.1, rel=1e-4)


# ------------------------------------------------------------------------------------------------ DDAD
def _make_toy_ddad(root, frames=2, seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    H, W = 96, 160                                          # stands in for 1216 x 1936
    lines = []
    for cam in ('CAMERA_01', 'CAMERA_05', 'CAMERA_07'):
        os.makedirs(os.path.join(root, 'pe', cam), exist_ok=True)
        pe = np.where(np.arange(H)[:, None] > 40, 300.0 / (np.arange(H)[:, None] - 39.5), -3.0) * np.ones((1, W))
        np.savez(os.path.join(root, 'pe', cam, 'ddad_pe.npz'), pe=pe.astype(np.float32))
        for f in range(frames):
            rgb_dir, d_dir = os.path.join(root, '000001', 'rgb', cam), os.path.join(root, '000001', 'depth', cam)
            os.makedirs(rgb_dir, exist_ok=True); os.makedirs(d_dir, exist_ok=True)
            Image.fromarray(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).save(os.path.join(rgb_dir, f'{f}.png'))
            depth = np.where(rng.random((H, W)) < 0.1, rng.uniform(1, 150, (H, W)), 0.0).astype(np.float32)
            np.savez(os.path.join(d_dir, f'{f}.npz'), depth=depth)
            k = np.where(depth > 0, np.clip(np.rint(rng.normal(0, 1.5, (H, W))), -5, 5), 255).astype(np.float32)
            np.savez(os.path.join(d_dir, f'{f}_slope_public_debug.npz'), k_img=k)
            lines.append(f'{os.path.join(rgb_dir, f"{f}.png")} {os.path.join(d_dir, f"{f}.npz")}')
    split = os.path.join(root, 'ddad_split.txt')
    with open(split, 'w') as fh:
        fh.write('\n'.join(lines) + '\n')
    return split


def test_area_resize_and_sparse_depth_splat():
    g = np.random.default_rng(1)
    img = g.random((12, 20, 3)).astype(np.float32)
    t = torch.from_numpy(img).permute(2, 0, 1)[None]
    assert np.allclose(I.imresize_area(img, (10, 6)), F.adaptive_avg_pool2d(t, (6, 10))[0].permute(1, 2, 0).numpy(), atol=1e-6)
    assert np.allclose(I.imresize_area(img, (7, 5)).mean(), img.mean(), atol=1e-3)          # area averaging preserves the mean
    from gedepth_amd.depth.datasets.pipelines import DDADResize
    depth = np.zeros((8, 8), np.float32); depth[1, 2] = 5.0; depth[7, 7] = 9.0; depth[6, 7] = 4.0
    out = DDADResize((4, 4))(dict(img=np.zeros((8, 8, 3), np.uint8), depth_gt=depth))['depth_gt']
    assert out.shape == (4, 4) and out[0, 1] == 5.0 and out[3, 3] == 9.0 and (out > 0).sum() == 2       # (6,7) and (7,7) collide


def test_ddad_dataset_pipeline_and_eval(tmp_path):
    root = str(tmp_path)
    split = _make_toy_ddad(root)
    norm = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
    shape = (48, 80)
    train_pipeline = [
        dict(type='LoadDDADImageFromFile', USEPE=True, USE_DYNAMIC_PE=True, pe_root=os.path.join(root, 'pe')),
        dict(type='DDADDepthLoadAnnotations', USE_DYNAMIC_PE=True),
        dict(type='LoadDDADCamIntrinsic'),
        dict(type='DDADResize', shape=shape, USE_DYNAMIC_PE=True),
        dict(type='Resize', ratio_range=(0.5, 2.0)),
        dict(type='Padding', img_padding_value=(0, 0, 0), depth_padding_value=255, pe_k=True, ori_h=shape[0], ori_w=shape[1]),
        dict(type='RandomRotate', prob=0.5, degree=2.5),
        dict(type='RandomFlip', prob=0.0),
        dict(type='RandomCrop', crop_size=shape),
        dict(type='ColorAug', prob=0.5),
        dict(type='Normalize', depth_scale=250, **norm),
        dict(type='DefaultFormatBundle'),
        dict(type='Collect', keys=['img', 'depth_gt', 'pe_k_gt', 'height'],
             meta_keys=('filename', 'ori_filename', 'ori_shape', 'img_shape', 'pad_shape', 'scale_factor', 'flip',
                        'flip_direction', 'img_norm_cfg', 'cam_intrinsic')),
    ]
    ds = build_dataset(dict(type='DDADDataset', pipeline=train_pipeline, split=split, max_depth=200,
                            cameras=['CAMERA_%02d' % i for i in (1, 5, 6, 9)]))
    assert len(ds) == 4                                                       # CAMERA_07 is filtered out
    random.seed(2); np.random.seed(2)
    s = ds[0]
    assert s['img'].shape == (5,) + shape and s['depth_gt'].shape == (1,) + shape and s['pe_k_gt'].shape == shape
    assert float(s['height']) == pytest.approx(1.56) and s['img_metas']['cam_intrinsic'][0][0] == pytest.approx(2181.5303)
    assert s['img'][3].max() <= 1.0 + 1e-6                                    # ground depth / 250
    test_pipeline = [
        dict(type='LoadDDADImageFromFile', USEPE=True, USE_DYNAMIC_PE=True, pe_root=os.path.join(root, 'pe')),
        dict(type='DDADResize', shape=shape, depth=False),
        dict(type='MultiScaleFlipAug', img_scale=shape, flip=False, flip_direction='horizontal', transforms=[
            dict(type='Normalize', depth_scale=250, **norm), dict(type='ImageToTensor', keys=['img']),
            dict(type='Collect', keys=['img', 'height', 'test'],
                 meta_keys=('filename', 'ori_filename', 'ori_shape', 'img_shape', 'pad_shape', 'scale_factor', 'flip',
                            'flip_direction', 'img_norm_cfg'))])]
    dt = build_dataset(dict(type='DDADDataset', pipeline=test_pipeline, split=split, test_mode=True,
                            cameras=['CAMERA_01', 'CAMERA_05']))
    batch = next(iter(build_dataloader(dt, 2, 0, dist=False, shuffle=False, pin_memory=False)))
    assert batch['img'][0].shape == (2, 5) + shape and batch['height'][0].shape == (2,) and int(batch['test'][0][0]) == 0
    gt = np.load(dt.img_infos[0]['ann']['depth_map'])['depth']
    res, _ = dt.pre_eval([np.where(gt > 0, gt, 1.0)[None].astype(np.float32)], [0])      # full-resolution prediction
    assert dt.evaluate(res)['abs_rel'] == pytest.approx(0.0, abs=1e-6)
