#!/usr/bin/env python
"""DDAD ground-embedding preprocessing on the GPU — the computation of the reference's tools/preprocess_data_ddad.py:
per camera the ground-plane depth map (:29-44: the lidar frame's z = 0 plane seen through K @ inv(camera_pose) @
lidar_pose, no height term) and per frame the truncated integer slope-class map (:47-51,68-82).

    python tools/preprocess_data_ddad.py --calib cam.npz --size 1216 1936 --out ddad_pe.npz \
        [--camera CAMERA_01 --gt depth.npz --out-k depth_slope.npz]

``cam.npz`` holds ``intrinsics`` (3,3), ``camera_pose`` (4,4), ``lidar_pose`` (4,4) as dgp reports them (dgp itself is
not a dependency of this repo)."""
import argparse
import os.path as osp
import sys

import numpy as np
import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
from gedepth_amd.kernels import ground_plane, slope_class_ddad  # noqa: E402

CAMERA_HEIGHTS = {'CAMERA_01': 1.56, 'CAMERA_05': 1.57, 'CAMERA_06': 1.53, 'CAMERA_09': 1.53}


def plane_coefficients(intrinsics, camera_pose, lidar_pose):
    """-> (row 2 of inv(A[:3,:3]), RT[2]) with A = K4 @ inv(camera_pose) @ lidar_pose (float64, host: a 4x4 product)."""
    K4 = np.eye(4)
    K4[:3, :3] = np.asarray(intrinsics, dtype=np.float64)
    A = K4 @ np.linalg.inv(np.asarray(camera_pose, dtype=np.float64)) @ np.asarray(lidar_pose, dtype=np.float64)
    Rinv = np.linalg.inv(A[:3, :3])
    RT = Rinv @ A[0:3, 3]
    return Rinv[2].copy(), float(RT[2])


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--calib', required=True)
    p.add_argument('--size', type=int, nargs=2, required=True, metavar=('H', 'W'))
    p.add_argument('--out', required=True)
    p.add_argument('--camera', default='CAMERA_01', choices=sorted(CAMERA_HEIGHTS))
    p.add_argument('--gt', default=None, help='DDAD depth .npz (key "depth", float32)')
    p.add_argument('--out-k', default=None)
    a = p.parse_args()
    if a.gt and not a.out_k:
        p.error('--gt needs --out-k (where to write the slope classes)')
    c = np.load(a.calib)
    row2, num = plane_coefficients(c['intrinsics'], c['camera_pose'], c['lidar_pose'])
    pe64, _ = ground_plane(row2, num, a.size[0], a.size[1])
    np.savez_compressed(a.out, pe=pe64.cpu().numpy())
    if a.gt:
        gt = torch.from_numpy(np.load(a.gt)['depth'].astype(np.float32)).cuda()
        k = slope_class_ddad(gt, pe64, CAMERA_HEIGHTS[a.camera]).cpu().numpy().astype(np.int64)
        np.savez_compressed(a.out_k, k_img=k)


if __name__ == '__main__':
    main()
