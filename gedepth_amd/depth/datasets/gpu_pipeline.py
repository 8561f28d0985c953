"""Device-side KITTI training pipeline for the ground embedding.

The host pipeline of the reference (configs/depthformer/depthformer_v.py:13-28) decodes a frame, concatenates two
full-resolution float ground-depth channels read from ``pe_165.npy``, and then resizes / pads / rotates / flips / crops a
five-channel float image up to 704 x 2432 on the CPU — ~150 ms per sample and worker.  Here the loader workers
only decode files (``KITTIRawDataset``: uint8 BGR image, uint16 depth PNG, slope-class map) and draw the random
augmentation parameters; everything else runs on the MI355X:

* the ground depth of the frame's calibration day is computed ON DEVICE from the calibration files with ``ge_ground_plane``
  (tools/preprocess_data_kitti.py:29-56 of the reference; cached per day) — or, when a tree only ships ``pe_165.npy``,
  uploaded once per day — and filtered into channels 3 / 4 by ``ge_aug_load`` (loading.py:397-403);
* KBCrop, Resize (bilinear for the 5 channels, nearest for depth / slope classes), Padding, RandomRotate, RandomFlip,
  RandomCrop, ColorAug and Normalize are the kernels of csrc/aug.hip, applied with the SAME parameters to all five image
  channels, the depth map and the class map.

``draw_params`` consumes ``np.random`` / ``random`` in exactly the order the host transforms do (transforms.py), so that a
seeded run of either pipeline sees the same augmentation — which is how tests/test_gpu_pipeline.py compares them.
"""
import ctypes
import os.path as osp
import random

import numpy as np
import torch
from PIL import Image

from ... import hip
from .builder import DATASETS
from .kitti import KITTIDataset
from .pipelines.imageops import _inverse_rotation, rescale_size

_f32 = torch.float32


# ------------------------------------------------------------------------------------------------ host side
@DATASETS.register_module()
class KITTIRawDataset(KITTIDataset):
    """File decoding only: what a loader worker does when the pipeline runs on the GPU.  ``pipeline`` is ignored."""

    def __init__(self, pipeline=None, **kw):
        super().__init__(pipeline=[], **kw)

    def __getitem__(self, idx):
        info = self.img_infos[idx]
        name = info['filename']
        filename = osp.join(self.img_dir, name)
        bgr = np.ascontiguousarray(np.asarray(Image.open(filename).convert('RGB'))[..., ::-1])
        out = dict(filename=filename, ori_filename=name, date=name.split('/')[0], bgr=torch.from_numpy(bgr))
        if 'ann' in info:
            depth_file = osp.join(self.ann_dir, info['ann']['depth_map'])
            out['depth_png'] = torch.from_numpy(np.asarray(Image.open(depth_file)).astype(np.uint16))
            k_file = depth_file.replace('.png', '.npz').replace('gt_depth', 'slope_range_5_5_interval_1')
            if osp.isfile(k_file):
                k = np.load(k_file)['k_img'].astype(np.float32) + 5            # DepthLoadAnnotations, loading.py:136-150
                k[k == 260] = 255
                out['pe_k'] = torch.from_numpy(k)
        return out


def raw_collate(samples):
    return samples                                           # frames of different days have different sizes: keep the list


def draw_params(h=352, w=1216, ratio_range=(0.5, 2.0), canvas=(352, 1216), rotate_prob=0.5, degree=2.5, flip_prob=0.5,
                crop_size=(352, 704), color_prob=0.5, gamma_range=(0.9, 1.1), brightness_range=(0.9, 1.1), color_range=(0.9, 1.1)):
    """The random draws of Resize -> Padding -> RandomRotate -> RandomFlip -> RandomCrop -> ColorAug in the host
    transforms' own order and generators (transforms.py:Resize.random_sample_ratio, Padding, RandomRotate, RandomFlip,
    RandomCrop.get_crop_bbox, ColorAug)."""
    p = {}
    lo, hi = ratio_range
    ratio = np.random.random_sample() * (hi - lo) + lo
    scale = (int(w * ratio), int(h * ratio))
    (nw, nh), _ = rescale_size((w, h), scale, return_scale=True)
    p['resize'] = (nh, nw)
    H, W = nh, nw
    p['pad'] = None
    if H < canvas[0] or W < canvas[1]:
        p['pad'] = (random.randint(0, canvas[0] - H), random.randint(0, canvas[1] - W))
        H, W = canvas
    rotate = bool(np.random.rand() < rotate_prob)
    deg = np.random.uniform(-degree, degree)
    p['rotate'] = float(deg) if rotate else None
    p['flip'] = bool(np.random.rand() < flip_prob)
    margin_h, margin_w = max(H - crop_size[0], 0), max(W - crop_size[1], 0)
    p['crop'] = (int(np.random.randint(0, margin_h + 1)), int(np.random.randint(0, margin_w + 1)))
    p['color'] = None
    if np.random.rand() < color_prob:
        gamma = np.random.uniform(min(*gamma_range), max(*gamma_range))
        brightness = np.random.uniform(min(*brightness_range), max(*brightness_range))
        colors = np.random.uniform(min(*color_range), max(*color_range), size=3)
        p['color'] = (float(gamma), float(brightness), [float(c) for c in colors])
    return p


def read_kitti_calibration(cam_path, velo_path):
    """P_rect_02, R_rect_00 (4x4), Tr_velo_to_cam (4x4) as the reference parses them (preprocess_data_kitti.py:21-36)."""
    cam, velo = open(cam_path).readlines(), open(velo_path).readlines()
    f = lambda line: [float(x) for x in line.strip('\n').split(' ')[1:]]
    P2 = np.array(f(cam[25])).reshape(3, 4)
    R0 = np.eye(4)
    R0[:3, :3] = np.array(f(cam[8])).reshape(3, 3)
    Tr = np.eye(4)
    Tr[:3, :3] = np.array(f(velo[1])).reshape(3, 3)
    Tr[:3, 3] = np.array(f(velo[2]))
    return P2, R0, Tr


# ---------------------------------------------------------------------------------------------- device side
def _lib():
    return hip.lib()


def _planes(C, H, W, dev):
    return torch.empty(C, H, W, device=dev, dtype=_f32)


class KITTIGPUPipeline:
    """``pipe(sample, params)`` -> ``dict(img (5,352,704), depth_gt (1,352,704), pe_k_gt (352,704), pe_ori_point)`` on
    ``device``; ``pipe.batch(samples)`` draws parameters per sample and stacks."""

    def __init__(self, data_root=None, img_dir='input', device='cuda', pe_source='calib', cam_height=1.65, mean=(123.675, 116.28, 103.53),
                 std=(58.395, 57.12, 57.375), to_rgb=True, depth_scale=256.0, pe_depth_scale=200.0, kb_crop=(352, 1216),
                 crop_size=(352, 704), **draw_kw):
        assert pe_source in ('calib', 'npy')
        self.root = img_dir if data_root is None or osp.isabs(img_dir) else osp.join(data_root, img_dir)
        self.device = torch.device(device)
        self.pe_source, self.cam_height = pe_source, cam_height
        # Normalize holds mean / std as float32 and widens them to float64 (imageops.imnormalize)
        self.mean = (ctypes.c_double * 3)(*[float(np.float32(m)) for m in mean])
        self.std = (ctypes.c_double * 3)(*[float(np.float32(s)) for s in std])
        self.to_rgb, self.depth_scale, self.pe_depth_scale = to_rgb, float(depth_scale), float(pe_depth_scale)
        self.kb_crop, self.crop_size = tuple(kb_crop), tuple(crop_size)
        self.draw_kw = dict(draw_kw, canvas=self.kb_crop, crop_size=self.crop_size)
        self._pe = {}

    # ---- ground depth of a calibration day, resident on the device
    def ground_depth(self, date, H, W):
        key = (date, H, W)
        if key not in self._pe:
            if self.pe_source == 'calib':
                from ...kernels import ground_plane
                d = osp.join(self.root, date)
                P2, R0, Tr = read_kitti_calibration(osp.join(d, 'calib_cam_to_cam.txt'), osp.join(d, 'calib_velo_to_cam.txt'))
                A = P2 @ R0 @ Tr
                Rinv = np.linalg.inv(A[:3, :3])
                RT = Rinv @ A[:3, 3]
                _, pe32 = ground_plane(Rinv[2], float(RT[2] - self.cam_height), H, W, device=self.device, want_f64=False)
            else:
                pe = np.load(osp.join(self.root, date, 'pe', 'pe_165.npy')).astype(np.float32)
                assert pe.shape == (H, W), (pe.shape, H, W)
                pe32 = torch.from_numpy(pe).to(self.device)
            self._pe[key] = pe32.contiguous()
        return self._pe[key]

    # ---- kernels
    def _resize(self, x, size, mode):
        C, Hs, Ws = x.shape
        out = _planes(C, size[0], size[1], x.device)
        hip.check(_lib().ge_aug_resize(hip.ptr(x), hip.ptr(out), C, Hs, Ws, size[0], size[1], mode, hip.stream()), 'ge_aug_resize')
        return out

    def _window(self, x, size, oy, ox, flip=False, fill=0.0):
        C, Hs, Ws = x.shape
        out = _planes(C, size[0], size[1], x.device)
        hip.check(_lib().ge_aug_window(hip.ptr(x), hip.ptr(out), C, Hs, Ws, size[0], size[1], int(oy), int(ox), int(flip), float(fill),
                                       hip.stream()), 'ge_aug_window')
        return out

    def _rotate(self, x, angle, border, mode):
        C, H, W = x.shape
        inv, off = _inverse_rotation(H, W, float(angle), None, 1.0)                 # float64, as the host
        m = np.array([inv[0, 0], inv[0, 1], off[0], inv[1, 0], inv[1, 1], off[1]]).astype(np.float32)
        arr = (ctypes.c_float * 6)(*[float(v) for v in m])
        out = _planes(C, H, W, x.device)
        hip.check(_lib().ge_aug_rotate(hip.ptr(x), hip.ptr(out), C, H, W, ctypes.cast(arr, ctypes.c_void_p), float(border), mode,
                                       hip.stream()), 'ge_aug_rotate')
        return out

    def _front(self, sample):
        """Decoded files -> the planar maps the augmentation chain works on: img (5, h, w), depth (1, h, w) | None, slope classes
        (1, h, w) | None, pe_ori_point.  KITTI: KB crop window (transforms.py:150-205)."""
        dev = self.device
        bgr = sample['bgr'].to(dev, non_blocking=True).contiguous()
        H, W = bgr.shape[:2]
        pe = self.ground_depth(sample['date'], H, W)
        kh, kw = self.kb_crop
        top, left = int(H - kh), int((W - kw) / 2)                                   # KBCrop, transforms.py:150-205
        img = _planes(5, kh, kw, dev)
        hip.check(_lib().ge_aug_load(hip.ptr(bgr), hip.ptr(pe), hip.ptr(img), H, W, top, left, kh, kw, 200.0, hip.stream()), 'ge_aug_load')
        depth = k = None
        if 'depth_png' in sample:
            png = sample['depth_png'].to(dev, non_blocking=True).contiguous()
            depth = _planes(1, kh, kw, dev)
            hip.check(_lib().ge_aug_depth(hip.ptr(png), hip.ptr(depth), H, W, top, left, kh, kw, self.depth_scale, hip.stream()), 'ge_aug_depth')
        if 'pe_k' in sample:
            kmap = sample['pe_k'].to(dev, non_blocking=True).contiguous()
            if tuple(kmap.shape) != (H, W):                                          # loading.py:146: nearest resize to the depth map
                kmap = self._resize(kmap[None], (H, W), 0)[0]
            k = self._window(kmap[None], (kh, kw), top, left, fill=255.0)
        return img, depth, k, pe[-1, -1]

    def __call__(self, sample, params):
        img, depth, k, pe_ori_point = self._front(sample)
        kh, kw = self.kb_crop
        maps = [(img, 1, 0.0), (depth, 0, 0.0), (k, 0, 255.0)]                        # (planes, interpolation, border / fill)
        # Resize
        maps = [(None if x is None else self._resize(x, params['resize'], mode), mode, fill) for x, mode, fill in maps]
        # Padding
        if params['pad'] is not None:
            oy, ox = params['pad']
            maps = [(None if x is None else self._window(x, self.kb_crop, -oy, -ox, fill=fill), mode, fill) for x, mode, fill in maps]
        # RandomRotate
        if params['rotate'] is not None:
            maps = [(None if x is None else self._rotate(x, params['rotate'], fill, mode), mode, fill) for x, mode, fill in maps]
        # RandomFlip + RandomCrop
        oy, ox = params['crop']
        maps = [None if x is None else self._window(x, self.crop_size, oy, ox, flip=params['flip'], fill=fill) for x, mode, fill in maps]
        img, depth, k = maps
        out_img = torch.empty_like(img)
        col = params['color']
        colors = (ctypes.c_double * 3)(*(col[2] if col else (1.0, 1.0, 1.0)))
        hip.check(_lib().ge_aug_color_normalize(hip.ptr(img), hip.ptr(out_img), img.shape[1], img.shape[2], int(col is not None),
                                                float(col[0]) if col else 1.0, float(col[1]) if col else 1.0,
                                                ctypes.cast(colors, ctypes.c_void_p), ctypes.cast(self.mean, ctypes.c_void_p),
                                                ctypes.cast(self.std, ctypes.c_void_p), self.pe_depth_scale, int(self.to_rgb),
                                                hip.stream()), 'ge_aug_color_normalize')
        out = dict(img=out_img, pe_ori_point=pe_ori_point,
                   img_metas=dict(filename=sample['filename'], ori_filename=sample['ori_filename'], ori_shape=(kh, kw, 5),
                                  img_shape=tuple(out_img.shape[1:]) + (5,), pad_shape=tuple(out_img.shape[1:]) + (5,),
                                  flip=params['flip'], flip_direction='horizontal'))
        if depth is not None:
            out['depth_gt'] = depth
        if k is not None:
            out['pe_k_gt'] = k[0]
        if 'height' in sample:
            out['height'] = torch.tensor(float(sample['height']), device=self.device)
        return out

    def batch(self, samples):
        outs = [self(s, draw_params(self.kb_crop[0], self.kb_crop[1], **self.draw_kw)) for s in samples]
        data = {k: torch.stack([o[k] for o in outs], 0) for k in outs[0] if k != 'img_metas'}
        data['img_metas'] = [o['img_metas'] for o in outs]
        return data


# ------------------------------------------------------------------------------------------------ DDAD
@DATASETS.register_module()
class DDADRawDataset:
    """File decoding only for DDAD (what LoadDDADImageFromFile / DDADDepthLoadAnnotations read, loading.py:743-955): uint8 BGR frame,
    the float32 sparse depth of the .npz, the slope classes (+5, 255 = ignore), camera name and height."""

    def __init__(self, split, cameras=('CAMERA_01', 'CAMERA_05', 'CAMERA_06', 'CAMERA_09'), pipeline=None, **kw):
        from .ddad import DDADDataset
        self._ds = DDADDataset(pipeline=[], cameras=cameras, split=split, **kw)
        self.img_infos = self._ds.img_infos

    def __len__(self):
        return len(self.img_infos)

    def __getitem__(self, idx):
        from .pipelines.loading import _DDAD_CAMERA_HEIGHT
        info = self.img_infos[idx]
        filename = info['filename']
        bgr = np.ascontiguousarray(np.asarray(Image.open(filename).convert('RGB'))[..., ::-1])
        depth_file = info['ann']['depth_map']
        camera = depth_file.split('/')[-2]
        out = dict(filename=filename, ori_filename=filename, date=camera, camera=camera, bgr=torch.from_numpy(bgr),
                   height=next(h for cam, h in _DDAD_CAMERA_HEIGHT.items() if cam in filename))
        if osp.isfile(depth_file):
            out['depth'] = torch.from_numpy(np.load(depth_file)['depth'].astype(np.float32))
            k_file = depth_file.replace('depth_val', 'depth').replace('.npz', '_slope_public_debug.npz')
            if osp.isfile(k_file):
                k = np.load(k_file)['k_img'].astype(np.float32)
                ignore = k == 255
                k = k + 5
                k[ignore] = 255
                out['pe_k'] = torch.from_numpy(k)
        return out


class DDADGPUPipeline(KITTIGPUPipeline):
    """The DDAD training pipeline (configs/_base_/datasets/ddad_gedepth.py) on the device: DDADResize's area / nearest / sparse
    re-projection front end (ge_aug_area_u8, ge_aug_resize, ge_aug_splat: transforms.py:735-783) to ``shape``, then the shared chain
    Resize -> Padding -> RandomRotate -> RandomFlip(prob 0) -> RandomCrop -> ColorAug -> Normalize(depth_scale = 250) of aug.hip.  The
    per-camera ground depth ``<pe_root>/<camera>/ddad_pe.npz`` is uploaded once per camera."""

    def __init__(self, pe_root, shape=(384, 640), device='cuda', pe_depth_scale=250.0, **kw):
        kw.setdefault('flip_prob', 0.0)
        super().__init__(data_root=None, img_dir='', device=device, pe_source='npy', pe_depth_scale=pe_depth_scale, kb_crop=tuple(shape),
                         crop_size=tuple(shape), **kw)
        self.pe_root = pe_root

    def ground_depth(self, camera, H, W):
        key = (camera, H, W)
        if key not in self._pe:
            pe = np.load(osp.join(self.pe_root, camera, 'ddad_pe.npz'))['pe'].astype(np.float32)
            assert pe.shape == (H, W), (pe.shape, H, W)
            self._pe[key] = torch.from_numpy(pe).to(self.device).contiguous()
        return self._pe[key]

    def _front(self, sample):
        dev = self.device
        bgr = sample['bgr'].to(dev, non_blocking=True).contiguous()
        H, W = bgr.shape[:2]
        oh, ow = self.kb_crop
        pe_raw = self.ground_depth(sample['camera'], H, W)
        img = _planes(5, oh, ow, dev)
        hip.check(_lib().ge_aug_area_u8(hip.ptr(bgr), hip.ptr(img), H, W, oh, ow, hip.stream()), 'ge_aug_area_u8')
        pe = pe_raw.clone()                                                         # LoadDDADImageFromFile: channel 3 = pe with (250, inf) and (-inf, 0) zeroed
        pe[pe > 250] = 0
        pe[pe < 0] = 0
        img[3:5] = self._resize(torch.stack((pe, pe_raw)), (oh, ow), 0)             # cv2.INTER_NEAREST
        depth = k = None
        if 'depth' in sample:
            d = sample['depth'].to(dev, non_blocking=True).contiguous()
            depth = _planes(1, oh, ow, dev)
            hip.check(_lib().ge_aug_splat(hip.ptr(d), hip.ptr(depth), d.shape[0], d.shape[1], oh, ow, hip.stream()), 'ge_aug_splat')
        if 'pe_k' in sample:
            kk = sample['pe_k'].to(dev, non_blocking=True).contiguous()
            k = _planes(1, oh, ow, dev)
            hip.check(_lib().ge_aug_splat(hip.ptr(kk), hip.ptr(k), kk.shape[0], kk.shape[1], oh, ow, hip.stream()), 'ge_aug_splat')
        return img, depth, k, pe_raw[-1, -1]
