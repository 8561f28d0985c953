"""The KITTI evaluation and training entry points end to end on MI355X with the toy tree."""
import os
import random
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from gedepth_amd.depth.apis.test import single_gpu_test                      # noqa: E402
from gedepth_amd.depth.datasets import build_dataloader, build_dataset      # noqa: E402
from gedepth_amd.depth.models import build_depther                           # noqa: E402
from gedepth_amd.mmrt.config import Config                                   # noqa: E402
from gedepth_amd.mmrt.optim import build_optimizer                           # noqa: E402
from toy_kitti import make_toy_kitti                                         # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def setup(tmp_path_factory):
    root = str(tmp_path_factory.mktemp('kitti'))
    split = make_toy_kitti(root)
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', 'depthformer_swint_v.py'))
    for part in ('train', 'val', 'test'):
        cfg.data[part].data_root = root
        cfg.data[part].split = split
    cfg.model.pretrained = None
    torch.manual_seed(0)
    model = build_depther(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))
    model.init_weights()
    return cfg, model.cuda()


def test_eigen_protocol_with_flip_tta(setup):
    cfg, model = setup
    ds = build_dataset(cfg.data.test, dict(test_mode=True))
    loader = build_dataloader(ds, 1, 0, dist=False, shuffle=False)
    preds = single_gpu_test(model, loader, pre_eval=False)
    assert len(preds) == 4 and preds[0].shape == (1, 352, 1216) and np.isfinite(preds[0]).all()
    assert preds[0].min() >= 1e-3 and preds[0].max() <= 80.0                     # clamped to the configured depth range
    with torch.autocast('cuda', dtype=torch.bfloat16):
        res = single_gpu_test(model, loader, pre_eval=True)
    summary = ds.evaluate(res)
    assert len(res) == 4 and all(np.isfinite(v) for v in summary.values()) and 0 <= summary['a1'] <= 1
    # flip-TTA really averages the two views: a single un-flipped view gives a different map
    batch = next(iter(loader))
    with torch.no_grad():
        single = model([batch['img'][0].cuda()], [batch['img_metas'][0]], return_loss=False,
                       pe_ori_point=[batch['pe_ori_point'][0].cuda()])[0]
        flipped = model([batch['img'][1].cuda()], [batch['img_metas'][1]], return_loss=False,
                        pe_ori_point=[batch['pe_ori_point'][1].cuda()])[0]
    assert np.allclose(preds[0], 0.5 * (single + flipped), rtol=1e-4, atol=1e-4)


def test_flip_tta_prediction_vs_oracle(setup):
    """The product's whole test path — dataset pipeline (KB crop, MultiScaleFlipAug with flip), ``single_gpu_test``, the model's
    ``aug_test`` / ``inference`` / flip-back — against the CPU oracle's restatement of encoder_decoder.py:196-274 (oracle.aug_test)
    fed with the SAME pipeline tensors and the same weights: fp32, exact-fp32 window attention, 1e-4 relative on every pixel."""
    from oracle import gedepth_oracle as O
    cfg, model = setup
    ds = build_dataset(cfg.data.test, dict(test_mode=True))
    loader = build_dataloader(ds, 1, 0, dist=False, shuffle=False)
    batch = next(iter(loader))
    assert len(batch['img']) == 2 and batch['img_metas'][1][0]['flip'] and not batch['img_metas'][0][0]['flip']
    variants = []
    for mod in model.modules():
        if hasattr(mod, 'kernel_variant'):
            variants.append((mod, mod.kernel_variant))
            mod.kernel_variant = 1
    try:
        model.eval()
        with torch.no_grad():
            got = model([t.cuda() for t in batch['img']], batch['img_metas'], return_loss=False,
                        pe_ori_point=[t.cuda() for t in batch['pe_ori_point']])[0]
    finally:
        for mod, v in variants:
            mod.kernel_variant = v
    P = {k: (v.detach().float() if v.is_floating_point() else v.detach()).cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        ref = O.aug_test([t.float() for t in batch['img']], batch['img_metas'], P, dict(O.SWIN_T, adaptive=False))[0].numpy()
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)
    print(f'\n[flip-TTA 352x1216 vs oracle] max rel {rel.max():.2e} mean {rel.mean():.2e}')
    assert got.shape == ref.shape == (1, 352, 1216)
    assert rel.max() <= 2e-4 and rel.mean() <= 1e-5, (rel.max(), rel.mean())          # two fp32 evaluations, as in test_full_size_forward_vs_oracle


def test_train_step_on_pipeline_batch(setup):
    cfg, model = setup
    random.seed(0); np.random.seed(0)
    ds = build_dataset(cfg.data.train)
    loader = build_dataloader(ds, 2, 0, dist=False, shuffle=False, drop_last=True)
    batch = next(iter(loader))
    batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    assert batch['img'].shape == (2, 5, 352, 704)
    model.train()
    opt = build_optimizer(model, cfg.optimizer, cfg.optimizer_config.get('grad_clip'))
    opt.zero_grad()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out = model.train_step(batch, opt)
    out['loss'].backward()
    opt.step()
    assert np.isfinite(float(out['log_vars']['loss']))


def test_ddad_adaptive_train_and_eval_with_camera_heights(tmp_path):
    """§8 f4: GEDepth-Adaptive on DDAD-style samples — per-sample camera ``height`` reaches the ground-embedding kernel in
    training (tensor) and in testing (list over augmentations + ``test`` flag, encoder_decoder.py:88-94)."""
    from test_dataset_cpu import _make_toy_ddad
    root = str(tmp_path)
    split = _make_toy_ddad(root)
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', 'depthformer_swint_a.py'))
    cfg.model.pretrained = None
    cfg.model.depth_scale = 250
    torch.manual_seed(0)
    model = build_depther(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))
    model.init_weights()
    model = model.cuda()
    ddad = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', 'depthformer_a_ddad.py'))
    shape = (96, 160)

    def patch(pipeline):
        out = []
        for t in pipeline:
            t = dict(t)
            if t['type'] == 'LoadDDADImageFromFile':
                t['pe_root'] = os.path.join(root, 'pe')
            if t['type'] == 'DDADResize':
                t['shape'] = shape
            if t['type'] == 'Padding':
                t['ori_h'], t['ori_w'] = shape
            if t['type'] == 'RandomCrop':
                t['crop_size'] = shape
            if t['type'] == 'MultiScaleFlipAug':
                t['img_scale'] = shape
            out.append(t)
        return out
    tr = dict(ddad.data.train); tr.update(pipeline=patch(ddad.data.train.pipeline), split=split)
    te = dict(ddad.data.test); te.update(pipeline=patch(ddad.data.test.pipeline), split=split)
    random.seed(0); np.random.seed(0)
    train_set = build_dataset(tr)
    batch = next(iter(build_dataloader(train_set, 2, 0, dist=False, shuffle=False, drop_last=True)))
    batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    assert batch['height'].shape == (2,)
    model.train()
    opt = build_optimizer(model, cfg.optimizer, cfg.optimizer_config.get('grad_clip'))
    opt.zero_grad()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out = model.train_step(batch, opt)
    out['loss'].backward()
    opt.step()
    assert np.isfinite(float(out['log_vars']['loss'])) and 'decode.loss_dynamic_pe' in out['log_vars']
    test_set = build_dataset(te, dict(test_mode=True))
    res = single_gpu_test(model, build_dataloader(test_set, 1, 0, dist=False, shuffle=False), pre_eval=True)
    summary = test_set.evaluate(res)
    assert len(res) == len(test_set) == 4 and all(np.isfinite(v) for v in summary.values())
