"""Generate tests/golden/log_images.npz by calling the reference's DepthBaseDecodeHead.log_images
(depth/models/decode_heads/decode_head.py:628-648) itself, imported through the stand-in of make_golden.py
(``mmcv.imdenormalize`` = img * std + mean, then RGB->BGR when ``to_bgr``, the published mmcv 1.3.13 arithmetic in float32).
Build-container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_logimg.py

The method reads nothing from ``self``, so it is called unbound.  Two cases: the KITTI configs' ``to_rgb=True`` and ``to_rgb=False``
(the final channel order differs: the method flips the channels once more after imdenormalize)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_golden as MG  # noqa: E402


def main():
    MG.install_shim()
    import depth.models  # noqa: F401
    from depth.models.decode_heads.decode_head import DepthBaseDecodeHead
    g = torch.Generator().manual_seed(11)
    H, W = 24, 40
    img = torch.randn(5, H, W, generator=g) * 1.3          # normalised RGB + the two ground channels; some values clip at 0 / 255
    img[3:] = torch.rand(2, H, W, generator=g)
    pred = 0.001 + 80 * torch.rand(1, H, W, generator=g)
    gt = torch.where(torch.rand(1, H, W, generator=g) < 0.3, 1 + 79 * torch.rand(1, H, W, generator=g), torch.zeros(1, H, W))
    out = dict(img=img, depth_pred=pred, depth_gt=gt, mean=np.asarray([123.675, 116.28, 103.53], np.float32), std=np.asarray([58.395, 57.12, 57.375], np.float32))
    for tag, to_rgb in (('rgb', True), ('bgr', False)):
        meta = dict(img_norm_cfg=dict(mean=out['mean'], std=out['std'], to_rgb=to_rgb))
        r = DepthBaseDecodeHead.log_images(None, img, pred, gt, meta)
        out[f'img_rgb_{tag}'] = np.ascontiguousarray(r['img_rgb'])
        out[f'img_depth_pred_{tag}'] = r['img_depth_pred']
        out[f'img_depth_gt_{tag}'] = r['img_depth_gt']
    MG.save('log_images', **out)


if __name__ == '__main__':
    main()
