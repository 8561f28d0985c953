"""A tiny KITTI-shaped tree (2 drives x 2 frames, raw-size 375 x 1242 images) for the dataset / pipeline / eval tests."""
import os
import os.path as osp

import numpy as np
from PIL import Image


def make_toy_kitti(root, seed=0, frames=2):
    rng = np.random.default_rng(seed)
    H, W = 375, 1242
    date = '2011_09_26'
    lines = []
    pe_dir = osp.join(root, 'input', date, 'pe')
    os.makedirs(pe_dir, exist_ok=True)
    v = np.arange(H, dtype=np.float64).reshape(H, 1)
    pe = np.where(v > 173.0, 1.65 * 721.5377 / np.maximum(v - 172.854, 1e-6), -5.0) * np.ones((1, W))
    np.save(osp.join(pe_dir, 'pe_165.npy'), pe)
    for drive in ('0001', '0002'):
        d = f'{date}_drive_{drive}_sync'
        img_dir = osp.join(root, 'input', date, d, 'image_02', 'data')
        gt_dir = osp.join(root, 'gt_depth', d, 'proj_depth', 'groundtruth', 'image_02')
        k_dir = osp.join(root, 'slope_range_5_5_interval_1', d, 'proj_depth', 'groundtruth', 'image_02')
        for p in (img_dir, gt_dir, k_dir):
            os.makedirs(p, exist_ok=True)
        for f in range(frames):
            name = f'{f + 5:010d}.png'
            Image.fromarray(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).save(osp.join(img_dir, name))
            depth = np.where(rng.random((H, W)) < 0.05, rng.uniform(1.0, 80.0, (H, W)), 0.0)
            Image.fromarray((depth * 256).astype(np.uint16)).save(osp.join(gt_dir, name))
            k = np.where(depth > 0, np.clip(np.rint(rng.normal(0, 1.5, (H, W))), -5, 5), 255).astype(np.float32)
            np.savez(osp.join(k_dir, name.replace('.png', '.npz')), k_img=k)
            lines.append(f'{date}/{d}/image_02/data/{name} {d}/proj_depth/groundtruth/image_02/{name} 721.5377')
    lines.append(f'{date}/{date}_drive_0002_sync/image_02/data/0000000099.png None 721.5377')
    split = osp.join(root, 'split.txt')
    with open(split, 'w') as fh:
        fh.write('\n'.join(reversed(lines)) + '\n')
    return split
