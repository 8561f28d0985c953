"""KITTI-shaped synthetic batches (SURVEY.md §8 d) for throughput runs, smoke tests and parity at full size.

Channel layout as the reference's loader produces it (depth/datasets/pipelines/loading.py:490-526):
0-2 normalised RGB, 3 filtered ground depth / depth_scale, 4 raw ground depth (negative above the horizon).
Generated on the host with a seeded torch.Generator so that the CPU oracle and the GPU run see bit-identical
inputs.
"""
import torch

IMG_NORM_CFG = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)


def synthetic_batch(batch, H=352, W=1120, seed=1234, device='cpu', valid_fraction=0.05):
    g = torch.Generator().manual_seed(seed)
    img = torch.zeros(batch, 5, H, W)
    img[:, 0:3] = torch.randn(batch, 3, H, W, generator=g)
    # flat ground, KITTI 2011_09_26 intrinsics (reference depth/datasets/kitti.py:182-184), rows rescaled to 352
    v = torch.arange(H, dtype=torch.float32).view(1, H, 1).expand(batch, H, W) * (352.0 / H)
    pe_raw = 1.65 * 721.5377 / (v - 172.854)
    img[:, 4] = pe_raw
    img[:, 3] = torch.where((pe_raw > 0) & (pe_raw <= 200), pe_raw, torch.zeros_like(pe_raw)) / 200.0
    valid = torch.rand(batch, 1, H, W, generator=g) < valid_fraction
    depth_gt = torch.where(valid, 1 + 79 * torch.rand(batch, 1, H, W, generator=g), torch.zeros(batch, 1, H, W))
    cls = torch.clamp(torch.round(torch.randn(batch, H, W, generator=g) * 1.5), -5, 5) + 5
    pe_k_gt = torch.where(valid[:, 0], cls, torch.full((batch, H, W), 255.0))
    metas = [dict(img_norm_cfg=IMG_NORM_CFG, flip=False, flip_direction=None, ori_shape=(H, W, 3), img_shape=(H, W, 3),
                  pad_shape=(H, W, 3), filename=f'synthetic_{seed}_{i}') for i in range(batch)]
    return dict(img=img.to(device), img_metas=metas, depth_gt=depth_gt.to(device), pe_k_gt=pe_k_gt.to(device))
