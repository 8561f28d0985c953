#!/usr/bin/env python
"""Ground-embedding preprocessing on the GPU — the computation of the reference's tools/preprocess_data_kitti.py:29-92:
per calibration the ground-plane depth map ``pe_165.npy`` and per frame the integer slope-class map.

    python tools/preprocess_data_kitti.py --calib-cam calib_cam_to_cam.txt --calib-velo calib_velo_to_cam.txt \
        --size 375 1242 --out pe_165.npy [--gt gt.png --out-k slope.npz]
"""
import argparse
import os.path as osp
import sys

import numpy as np
import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
from gedepth_amd.kernels import ground_plane, slope_class  # noqa: E402


def read_calib(cam_path, velo_path):
    cam = open(cam_path).readlines()
    velo = open(velo_path).readlines()
    f = lambda line: [float(x) for x in line.strip('\n').split(' ')[1:]]
    P2 = np.array(f(cam[25])).reshape(3, 4)
    R0 = np.eye(4); R0[:3, :3] = np.array(f(cam[8])).reshape(3, 3)
    Tr = np.eye(4); Tr[:3, :3] = np.array(f(velo[1])).reshape(3, 3); Tr[:3, 3] = np.array(f(velo[2]))
    return P2, R0, Tr


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--calib-cam', required=True); p.add_argument('--calib-velo', required=True)
    p.add_argument('--size', type=int, nargs=2, required=True, metavar=('H', 'W'))
    p.add_argument('--height', type=float, default=1.65)
    p.add_argument('--out', required=True)
    p.add_argument('--gt', default=None, help='16-bit KITTI depth PNG'); p.add_argument('--out-k', default=None)
    a = p.parse_args()
    P2, R0, Tr = read_calib(a.calib_cam, a.calib_velo)
    A = P2 @ R0 @ Tr
    Rinv = np.linalg.inv(A[:3, :3])
    RT = Rinv @ A[:3, 3]
    pe64, pe32 = ground_plane(Rinv[2], float(RT[2] - a.height), a.size[0], a.size[1])
    np.save(a.out, pe64.cpu().numpy())
    if a.gt:
        from PIL import Image
        gt = np.asarray(Image.open(a.gt), dtype=np.float64) / 256
        k = slope_class(torch.from_numpy(gt).cuda(), pe32, a.height, 'round').cpu().numpy().astype(np.float64)
        np.savez_compressed(a.out_k, k_img=k)


if __name__ == '__main__':
    main()
