"""the device-side KITTI training pipeline (gedepth_amd/depth/datasets/gpu_pipeline.py + csrc/aug.hip)
against the host pipeline (the reference's transform chain restated in depth/datasets/pipelines) on the toy tree, sample by
sample with identical seeded random draws."""
import os
import random

import numpy as np
import pytest
import torch

from toy_kitti import make_toy_kitti

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _host_dataset(root, split):
    from gedepth_amd.depth.datasets import build_dataset
    from gedepth_amd.mmrt.config import Config
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', 'depthformer_a.py'))
    d = cfg.data.train
    d.data_root, d.split = str(root), split
    return build_dataset(d)


def test_draw_params_follow_the_host_generators():
    """CPU: ``draw_params`` consumes np.random / random exactly like Resize, Padding, RandomRotate, RandomFlip, RandomCrop and
    ColorAug do (same order, same counts), for up- and down-scaling draws."""
    from gedepth_amd.depth.datasets.gpu_pipeline import draw_params
    from gedepth_amd.depth.datasets.pipelines import transforms as Tr
    for seed in range(12):
        np.random.seed(seed)
        random.seed(seed)
        p = draw_params()
        np.random.seed(seed)
        random.seed(seed)
        res = dict(img=np.zeros((352, 1216, 5), np.float32), depth_gt=np.zeros((352, 1216), np.float32),
                   pe_k_gt=np.zeros((352, 1216), np.float32), depth_fields=['depth_gt', 'pe_k_gt'])
        res = Tr.Resize(ratio_range=(0.5, 2.0))(res)
        assert res['img'].shape[:2] == p['resize']
        before = res['img'].shape[:2]
        st = random.getstate()
        res = Tr.Padding((0, 0, 0), 255, pe_k=True)(res)
        padded = res['img'].shape[:2] != before or (before[0] < 352 or before[1] < 1216)
        assert (p['pad'] is not None) == padded
        if padded:                                                  # the offsets Padding drew
            random.setstate(st)
            assert p['pad'] == (random.randint(0, 352 - before[0]), random.randint(0, 1216 - before[1]))
        rot = bool(np.random.rand() < 0.5)
        deg = np.random.uniform(-2.5, 2.5)
        assert (p['rotate'] is not None) == rot and (not rot or p['rotate'] == float(deg))
        assert p['flip'] == bool(np.random.rand() < 0.5)
        H, W = res['img'].shape[:2]
        assert p['crop'] == (np.random.randint(0, max(H - 352, 0) + 1), np.random.randint(0, max(W - 704, 0) + 1))
        if np.random.rand() < 0.5:
            g, b = np.random.uniform(0.9, 1.1), np.random.uniform(0.9, 1.1)
            c = np.random.uniform(0.9, 1.1, size=3)
            assert p['color'] == (float(g), float(b), [float(v) for v in c])
        else:
            assert p['color'] is None


@pytest.mark.gpu
def test_device_pipeline_matches_host_pipeline(tmp_path):
    """Every train sample of the toy tree through both pipelines with the same seeds: crop / pad / flip / nearest paths
    (depth, slope classes, image samples that were neither rescaled-bilinear nor rotated) are bit-exact; bilinear resize /
    rotation agree to fp32 rounding; the uint8 truncation inside Normalize may move a pixel whose value sits within
    rounding of an integer by 1/std — bounded and counted."""
    from gedepth_amd.depth.datasets.gpu_pipeline import KITTIGPUPipeline, KITTIRawDataset, draw_params
    assert torch.cuda.is_available()
    split = make_toy_kitti(str(tmp_path))
    host = _host_dataset(tmp_path, split)
    raw = KITTIRawDataset(img_dir='input', ann_dir='gt_depth', split=split, data_root=str(tmp_path))
    pipe = KITTIGPUPipeline(data_root=str(tmp_path), pe_source='npy')
    assert len(raw) == len(host) == 4
    seen = dict(pad=0, rotate=0, flip=0, color=0, up=0)
    for seed in range(10):
        idx = seed % len(host)
        np.random.seed(100 + seed)
        random.seed(100 + seed)
        ref = host[idx]
        np.random.seed(100 + seed)
        random.seed(100 + seed)
        params = draw_params()
        out = pipe(raw[idx], params)
        for k, hit in (('pad', params['pad'] is not None), ('rotate', params['rotate'] is not None), ('flip', params['flip']),
                       ('color', params['color'] is not None), ('up', params['resize'][0] > 352)):
            seen[k] += int(hit)
        img, img_ref = out['img'].cpu(), ref['img']
        assert img.shape == img_ref.shape == (5, 352, 704)
        assert out['img_metas']['flip'] == ref['img_metas']['flip']
        # integer-valued maps: nearest / index-only everywhere
        assert torch.equal(out['pe_k_gt'].cpu(), ref['pe_k_gt']), f'seed {seed}: slope classes differ'
        assert torch.equal(out['depth_gt'].cpu(), ref['depth_gt']), f'seed {seed}: depth differs'
        assert float(out['pe_ori_point']) == float(ref['pe_ori_point'])
        # ground-depth channels: bilinear (resize, rotate) in fp32
        for c in (3, 4):
            err = (img[c] - img_ref[c]).abs()
            assert err.max().item() <= 1e-5 * max(1.0, img_ref[c].abs().max().item()), (seed, c, err.max().item())
        # colour channels: (uint8 - mean) / std; a unit step of the uint8 truncation = 1/57 ~ 0.0175
        err = (img[:3] - img_ref[:3]).abs()
        steps = (err > 1e-5)
        assert err.max().item() <= 0.0176, (seed, err.max().item())
        assert steps.float().mean().item() <= 2e-3, (seed, steps.float().mean().item())
    assert all(v > 0 for v in seen.values()), seen                 # every branch of the pipeline was exercised


@pytest.mark.gpu
def test_ground_depth_channels_from_calibration_on_device(tmp_path):
    """pe_source='calib': channels 3-4 come from ge_ground_plane on the calibration files — never from disk — and equal what
    the host pipeline loads from the pe_165.npy that the reference's preprocessing (oracle.ground_plane) writes."""
    from gedepth_amd.depth.datasets.gpu_pipeline import KITTIGPUPipeline, KITTIRawDataset, draw_params
    from oracle import gedepth_oracle as O
    split = make_toy_kitti(str(tmp_path))
    date = '2011_09_26'
    P2 = np.array([[7.215377e+02, 0.0, 6.095593e+02, 4.485728e+01], [0.0, 7.215377e+02, 1.728540e+02, 2.163791e-01],
                   [0.0, 0.0, 1.0, 2.745884e-03]])
    R0 = np.eye(3)
    Tr = np.array([[7.533745e-03, -9.999714e-01, -6.166020e-04, -4.069766e-03], [1.480249e-02, 7.280733e-04, -9.998902e-01, -7.631618e-02],
                   [9.998621e-01, 7.523790e-03, 1.480755e-02, -2.717806e-01]])
    fmt = lambda v: ' '.join(f'{x:.6e}' for x in np.asarray(v).reshape(-1))
    cam = ['calib_time: x', 'corner_dist: 9.95e-02']
    for c in range(4):
        cam += [f'S_0{c}: 1 1', f'K_0{c}: ' + fmt(np.eye(3)), f'D_0{c}: ' + fmt(np.zeros(5)), f'R_0{c}: ' + fmt(np.eye(3)),
                f'T_0{c}: ' + fmt(np.zeros(3)), f'S_rect_0{c}: 1 1', f'R_rect_0{c}: ' + fmt(R0), f'P_rect_0{c}: ' + fmt(P2 if c == 2 else np.eye(3, 4))]
    d = tmp_path / 'input' / date
    (d / 'calib_cam_to_cam.txt').write_text('\n'.join(cam) + '\n')
    (d / 'calib_velo_to_cam.txt').write_text('calib_time: x\nR: ' + fmt(Tr[:, :3]) + '\nT: ' + fmt(Tr[:, 3]) + '\n')
    # what the reference's offline step stores for this calibration
    rd = lambda s: np.array([float(x) for x in s.split(' ')[1:]])
    Trf = np.eye(4)
    Trf[:3, :3] = rd('R: ' + fmt(Tr[:, :3])).reshape(3, 3)
    Trf[:3, 3] = rd('T: ' + fmt(Tr[:, 3]))
    pe, _, _ = O.ground_plane(rd(cam[25]).reshape(3, 4), rd(cam[8]).reshape(3, 3), Trf, 375, 1242)
    np.save(d / 'pe' / 'pe_165.npy', pe)
    host = _host_dataset(tmp_path, split)
    raw = KITTIRawDataset(img_dir='input', ann_dir='gt_depth', split=split, data_root=str(tmp_path))
    pipe = KITTIGPUPipeline(data_root=str(tmp_path), pe_source='calib')
    assert np.array_equal(pipe.ground_depth(date, 375, 1242).cpu().numpy(), pe.astype(np.float32))     # bit-exact, on device
    np.random.seed(7)
    random.seed(7)
    ref = host[1]
    np.random.seed(7)
    random.seed(7)
    out = pipe(raw[1], draw_params())
    for c in (3, 4):
        err = (out['img'][c].cpu() - ref['img'][c]).abs().max().item()
        assert err <= 1e-5 * max(1.0, ref['img'][c].abs().max().item()), (c, err)
    assert float(out['pe_ori_point']) == float(ref['pe_ori_point'])


@pytest.mark.gpu
def test_batches_feed_a_training_step(tmp_path):
    """``pipe.batch`` -> the model's train_step: shapes, dtypes and keys are what the host loader's collate produces."""
    from gedepth_amd.depth.datasets.gpu_pipeline import KITTIGPUPipeline, KITTIRawDataset, raw_collate
    from gedepth_amd.depth.models import build_depther
    from gedepth_amd.mmrt.config import Config
    split = make_toy_kitti(str(tmp_path))
    raw = KITTIRawDataset(img_dir='input', ann_dir='gt_depth', split=split, data_root=str(tmp_path))
    loader = torch.utils.data.DataLoader(raw, batch_size=2, num_workers=2, collate_fn=raw_collate)
    pipe = KITTIGPUPipeline(data_root=str(tmp_path), pe_source='npy')
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', 'depthformer_swint_a.py'))
    cfg.model.pretrained = None
    model = build_depther(cfg.model).cuda().train()
    n = 0
    for samples in loader:
        batch = pipe.batch(samples)
        assert batch['img'].shape == (2, 5, 352, 704) and batch['depth_gt'].shape == (2, 1, 352, 704)
        assert batch['pe_k_gt'].shape == (2, 352, 704) and batch['pe_ori_point'].shape == (2,)
        assert batch['img'].is_cuda and batch['img'].dtype == torch.float32
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out = model.train_step(batch, None)
        out['loss'].backward()
        assert np.isfinite(out['log_vars']['loss'])
        n += 1
    assert n == 2


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(48, 80), (36, 64)])
def test_ddad_device_pipeline_matches_host_pipeline(tmp_path, shape):
    """DDAD on the device pipeline (DDADGPUPipeline: ge_aug_area_u8 / nearest / ge_aug_splat front end = DDADResize,
    transforms.py:735-783, then the shared augmentation kernels) against the host pipeline — whose image operations are pinned against
    scipy / PIL in tests/test_imageops_independent.py — on the toy DDAD tree with identical seeded draws; integer (2x) and non-integer
    (2.67 x 2.5) shrink factors.  Depth and slope classes (index-only: re-projection, nearest, pad, crop) are bit-exact."""
    from test_dataset_cpu import _make_toy_ddad
    from gedepth_amd.depth.datasets import build_dataset
    from gedepth_amd.depth.datasets.gpu_pipeline import DDADGPUPipeline, DDADRawDataset, draw_params
    root = str(tmp_path)
    split = _make_toy_ddad(root, frames=2)
    norm = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
    pipeline = [
        dict(type='LoadDDADImageFromFile', USEPE=True, USE_DYNAMIC_PE=True, pe_root=os.path.join(root, 'pe')),
        dict(type='DDADDepthLoadAnnotations', USE_DYNAMIC_PE=True),
        dict(type='DDADResize', shape=shape, USE_DYNAMIC_PE=True),
        dict(type='Resize', ratio_range=(0.5, 2.0)),
        dict(type='Padding', img_padding_value=(0, 0, 0), depth_padding_value=255, pe_k=True, ori_h=shape[0], ori_w=shape[1]),
        dict(type='RandomRotate', prob=0.5, degree=2.5),
        dict(type='RandomFlip', prob=0.0),
        dict(type='RandomCrop', crop_size=shape),
        dict(type='ColorAug', prob=0.5, gamma_range=[0.9, 1.1], brightness_range=[0.9, 1.1], color_range=[0.9, 1.1]),
        dict(type='Normalize', depth_scale=250, **norm),
        dict(type='DefaultFormatBundle'),
        dict(type='Collect', keys=['img', 'depth_gt', 'pe_k_gt', 'height'], meta_keys=('filename', 'flip')),
    ]
    cams = ['CAMERA_%02d' % i for i in (1, 5, 6, 9)]
    host = build_dataset(dict(type='DDADDataset', pipeline=pipeline, split=split, max_depth=200, cameras=cams))
    raw = DDADRawDataset(split=split, cameras=cams)
    pipe = DDADGPUPipeline(pe_root=os.path.join(root, 'pe'), shape=shape)
    assert len(raw) == len(host) == 4
    seen = dict(pad=0, rotate=0, color=0, up=0)
    for seed in range(10):
        idx = seed % len(host)
        np.random.seed(200 + seed); random.seed(200 + seed)
        ref = host[idx]
        np.random.seed(200 + seed); random.seed(200 + seed)
        params = draw_params(shape[0], shape[1], canvas=shape, crop_size=shape, flip_prob=0.0)
        out = pipe(raw[idx], params)
        for k, hit in (('pad', params['pad'] is not None), ('rotate', params['rotate'] is not None), ('color', params['color'] is not None),
                       ('up', params['resize'][0] > shape[0])):
            seen[k] += int(hit)
        img, img_ref = out['img'].cpu(), ref['img']
        assert img.shape == img_ref.shape == (5,) + tuple(shape) and not params['flip']
        assert torch.equal(out['pe_k_gt'].cpu(), ref['pe_k_gt']), f'seed {seed}: slope classes differ'
        assert torch.equal(out['depth_gt'].cpu().double(), ref['depth_gt'].double()), f'seed {seed}: depth differs'
        assert float(out['height']) == pytest.approx(float(ref['height']))
        for c in (3, 4):
            err = (img[c] - img_ref[c]).abs()
            assert err.max().item() <= 1e-5 * max(1.0, img_ref[c].abs().max().item()), (seed, c, err.max().item())
        err = (img[:3] - img_ref[:3]).abs()                                            # one uint8 step of the truncation inside Normalize = 1 / 57
        assert err.max().item() <= 0.0176 and (err > 1e-5).float().mean().item() <= 5e-3, (seed, err.max().item(), (err > 1e-5).float().mean().item())
    assert all(v > 0 for v in seen.values()), seen
