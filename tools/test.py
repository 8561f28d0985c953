#!/usr/bin/env python
"""Evaluate / run a depther — CLI of the reference's tools/test.py:21-68 (config, checkpoint, --eval, --options).

The KITTI Eigen evaluation loop (dataset, Garg crop, flip TTA collation) is the next scope row;
this entry point already covers model construction, checkpoint loading in the mmcv layout, the ``forward_test``
protocol (``return_loss=False``, flip test-time augmentation) and the metric code on synthetic KITTI-shaped inputs.
"""
import argparse
import os.path as osp
import sys

import numpy as np
import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))

from gedepth_amd.depth.core import eval_metrics  # noqa: E402
from gedepth_amd.depth.datasets.synthetic import synthetic_batch  # noqa: E402
from gedepth_amd.depth.models import build_depther  # noqa: E402
from gedepth_amd.mmrt.checkpoint import load_checkpoint  # noqa: E402
from gedepth_amd.mmrt.config import Config, DictAction  # noqa: E402


def main():
    p = argparse.ArgumentParser(description='depth test (and eval) a model')
    p.add_argument('config')
    p.add_argument('checkpoint', nargs='?', default=None)
    p.add_argument('--eval', nargs='+', default=None)
    p.add_argument('--options', nargs='+', default=None)
    p.add_argument('--synthetic', type=int, default=2)
    p.add_argument('--flip-tta', action='store_true')
    args = p.parse_args()
    cfg = Config.fromfile(args.config)
    if args.options:
        cfg.merge_from_dict(DictAction.parse(args.options))
    cfg.model.pretrained = None
    cfg.model.train_cfg = None
    model = build_depther(cfg.model, test_cfg=cfg.get('test_cfg'))
    if args.checkpoint:
        load_checkpoint(model, args.checkpoint, map_location='cpu')
    model = model.cuda().eval()
    res = []
    for i in range(args.synthetic):
        b = synthetic_batch(1, 352, 1120, seed=100 + i, device='cuda', valid_fraction=0.05)
        imgs, metas = [b['img']], [b['img_metas']]
        if args.flip_tta:
            m = dict(b['img_metas'][0], flip=True, flip_direction='horizontal')
            imgs.append(b['img'].flip(3)); metas.append([m])
        with torch.no_grad():
            pred = model(imgs, metas, return_loss=False, pe_ori_point=[None] * len(imgs))[0]
        gt = b['depth_gt'][0, 0].cpu().numpy()
        res.append(eval_metrics(gt, np.clip(pred[0], 1e-3, 80)))
    for k in res[0]:
        print(f'{k}: {np.nanmean([r[k] for r in res]):.4f}')


if __name__ == '__main__':
    main()
