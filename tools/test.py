#!/usr/bin/env python
"""Evaluate / run a depther — CLI of the reference's tools/test.py:21-68 (config, checkpoint, --eval, --options).

With a KITTI tree at ``cfg.data.test.data_root`` this is the Eigen-split protocol of the reference: test pipeline with
flip test-time augmentation, ``forward_test`` (``return_loss=False``), KB crop + Garg crop, per-image metrics, nan-mean
summary (gedepth_amd/depth/apis/test.py, gedepth_amd/depth/datasets/kitti.py).  ``--synthetic N`` runs the same model
protocol on N synthetic KITTI-shaped inputs instead (no dataset needed).
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
    p.add_argument('--synthetic', type=int, default=2, help='N synthetic inputs; 0 = evaluate cfg.data.test')
    p.add_argument('--flip-tta', action='store_true')
    p.add_argument('--bf16', action='store_true', help='bf16 autocast inference')
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
    data_root = cfg.data.test.get('data_root')
    if args.synthetic <= 0 or (args.synthetic == 2 and data_root and osp.isdir(data_root) and args.eval):
        from gedepth_amd.depth.apis.test import single_gpu_test
        from gedepth_amd.depth.datasets import build_dataloader, build_dataset
        dataset = build_dataset(cfg.data.test, dict(test_mode=True))
        loader = build_dataloader(dataset, 1, cfg.data.workers_per_gpu, dist=False, shuffle=False)
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=args.bf16):
            results = single_gpu_test(model, loader, pre_eval=True)
        dataset.evaluate(results)
        return
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
