"""Geometric / photometric transforms of the KITTI pipeline on the 5-channel (BGR, pe, pe_raw) image, the depth map and
the slope-class map.  Restates depth/datasets/pipelines/transforms.py: Normalize :13-62, Padding :65-111, KBCrop :150-205,
RandomRotate :209-297, RandomFlip :300-354, RandomCrop :357-418, ColorAug :421-482, Resize :485-733 (ratio_range /
single-scale modes).  Random draws use the same generators in the same order (``np.random`` / ``random``)."""
import random

import numpy as np

from ..builder import PIPELINES
from .imageops import imflip, imnormalize, imrescale, imresize, imresize_area, imrotate


@PIPELINES.register_module()
class Normalize:
    """RGB: BGR->RGB, (x - mean) / std.  Channel 3 (filtered ground depth): positive values / depth_scale (200).
    Channel 4 (raw ground depth) untouched."""

    def __init__(self, mean, std, depth_scale=200, to_rgb=True):
        self.mean, self.std = np.array(mean, dtype=np.float32), np.array(std, dtype=np.float32)
        self.to_rgb, self.depth_scale = to_rgb, depth_scale

    def __call__(self, results):
        img_pe = results['img']
        if img_pe.shape[-1] == 5:
            rgb = imnormalize(img_pe[:, :, 0:3].copy().astype(np.uint8), self.mean, self.std, self.to_rgb)
            pe = img_pe[:, :, 3].copy()
            pe[pe > 0] = pe[pe > 0] / self.depth_scale
            results['img'] = np.concatenate([rgb, pe[:, :, None], img_pe[:, :, 4].copy()[:, :, None]], axis=-1)
        else:
            results['img'] = imnormalize(img_pe.copy(), self.mean, self.std, self.to_rgb)
        results['img_norm_cfg'] = dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb)
        return results

    def __repr__(self):
        return f'{self.__class__.__name__}(mean={self.mean}, std={self.std}, to_rgb={self.to_rgb})'


@PIPELINES.register_module()
class Padding:
    """After a down-scaling Resize: paste the sample at a random offset into a zero (352, 1216) canvas; the slope-class
    canvas is 255 (ignore).  (``img_padding_value`` / ``depth_padding_value`` are accepted and unused, as upstream.)"""

    def __init__(self, img_padding_value, depth_padding_value, normals=False, pe_k=False, ori_h=352, ori_w=1216):
        assert not normals
        self.img_padding_value, self.depth_padding_value = img_padding_value, depth_padding_value
        self.pe_k, self.ori_h, self.ori_w = pe_k, ori_h, ori_w

    def __call__(self, results):
        image, depth = results['img'].copy(), results['depth_gt'].copy()
        h, w = image.shape[:2]
        if h < self.ori_h or w < self.ori_w:
            new_img = np.zeros((self.ori_h, self.ori_w, 5)).astype(image.dtype)
            new_depth = np.zeros((self.ori_h, self.ori_w)).astype(depth.dtype)
            h_off = random.randint(0, self.ori_h - h)
            w_off = random.randint(0, self.ori_w - w)
            new_img[h_off:h_off + h, w_off:w_off + w] = image
            new_depth[h_off:h_off + h, w_off:w_off + w] = depth
            results['img'], results['depth_gt'] = new_img, new_depth
            if self.pe_k:
                k = results['pe_k_gt'].copy()
                new_k = 255 + np.zeros((self.ori_h, self.ori_w)).astype(k.dtype)
                new_k[h_off:h_off + h, w_off:w_off + w] = k
                results['pe_k_gt'] = new_k
        return results

    def __repr__(self):
        return f'{self.__class__.__name__}(ori=({self.ori_h}, {self.ori_w}), pe_k={self.pe_k})'


@PIPELINES.register_module()
class KBCrop:
    """KITTI benchmark crop: bottom-aligned, horizontally centred (352, 1216) window."""

    def __init__(self, depth=False, submodel=False, height=352, width=1216, normals=False, pe_k=False):
        self.depth, self.height, self.width, self.pe_k = depth, height, width, pe_k

    def __call__(self, results):
        height, width = results['img_shape'][0], results['img_shape'][1]
        top, left = int(height - self.height), int((width - self.width) / 2)
        if self.depth:
            results['depth_gt'] = results['depth_gt'][top:top + self.height, left:left + self.width]
            results['depth_shape'] = results['depth_gt'].shape
        if self.pe_k:
            results['pe_k_gt'] = results['pe_k_gt'][top:top + self.height, left:left + self.width]
        results['img'] = results['img'][top:top + self.height, left:left + self.width, :]
        results['ori_shape'] = results['img'].shape
        return results

    def __repr__(self):
        return self.__class__.__name__


@PIPELINES.register_module()
class RandomRotate:
    """Image bilinear with ``pad_val``; depth fields nearest with 0, slope classes ("pe" in the key) with 255."""

    def __init__(self, prob, degree, pad_val=0, depth_pad_val=0, center=None, auto_bound=False, normals=False):
        assert 0 <= prob <= 1 and not normals
        if isinstance(degree, (float, int)):
            assert degree > 0, f'degree {degree} should be positive'
            degree = (-degree, degree)
        assert len(degree) == 2
        self.prob, self.degree, self.pal_val, self.depth_pad_val = prob, degree, pad_val, depth_pad_val
        self.center, self.auto_bound = center, auto_bound

    def __call__(self, results):
        rotate = bool(np.random.rand() < self.prob)
        degree = np.random.uniform(min(*self.degree), max(*self.degree))
        if rotate:
            results['img'] = imrotate(results['img'], angle=degree, border_value=self.pal_val, center=self.center,
                                      auto_bound=self.auto_bound)
            for key in results.get('depth_fields', []):
                results[key] = imrotate(results[key], angle=degree, border_value=255 if 'pe' in key else self.depth_pad_val,
                                        center=self.center, auto_bound=self.auto_bound, interpolation='nearest')
        return results

    def __repr__(self):
        return f'{self.__class__.__name__}(prob={self.prob}, degree={self.degree})'


@PIPELINES.register_module()
class RandomFlip:

    def __init__(self, prob=None, direction='horizontal', normals=False):
        assert direction in ('horizontal', 'vertical') and not normals
        assert prob is None or 0 <= prob <= 1
        self.prob, self.direction = prob, direction

    def __call__(self, results):
        if 'flip' not in results:
            results['flip'] = bool(np.random.rand() < self.prob)
        if 'flip_direction' not in results:
            results['flip_direction'] = self.direction
        if results['flip']:
            results['img'] = imflip(results['img'], direction=results['flip_direction'])
            for key in results.get('depth_fields', []):
                results[key] = imflip(results[key], direction=results['flip_direction']).copy()
        return results

    def __repr__(self):
        return f'{self.__class__.__name__}(prob={self.prob})'


@PIPELINES.register_module()
class RandomCrop:

    def __init__(self, crop_size, normals=False):
        assert crop_size[0] > 0 and crop_size[1] > 0 and not normals
        self.crop_size = crop_size

    def get_crop_bbox(self, img):
        margin_h = max(img.shape[0] - self.crop_size[0], 0)
        margin_w = max(img.shape[1] - self.crop_size[1], 0)
        off_h = np.random.randint(0, margin_h + 1)
        off_w = np.random.randint(0, margin_w + 1)
        return off_h, off_h + self.crop_size[0], off_w, off_w + self.crop_size[1]

    @staticmethod
    def crop(img, bbox):
        y1, y2, x1, x2 = bbox
        return img[y1:y2, x1:x2, ...]

    def __call__(self, results):
        bbox = self.get_crop_bbox(results['img'])
        results['img'] = self.crop(results['img'], bbox)
        results['img_shape'] = results['img'].shape
        for key in results.get('depth_fields', []):
            results[key] = self.crop(results[key], bbox)
        results['depth_shape'] = results['img_shape']
        return results

    def __repr__(self):
        return f'{self.__class__.__name__}(crop_size={self.crop_size})'


@PIPELINES.register_module()
class ColorAug:
    """Gamma, brightness and per-channel colour jitter on the three colour channels (0..255 range), in place."""

    def __init__(self, prob=None, gamma_range=(0.9, 1.1), brightness_range=(0.9, 1.1), color_range=(0.9, 1.1)):
        assert prob is None or 0 <= prob <= 1
        self.prob, self.gamma_range, self.brightness_range, self.color_range = prob, gamma_range, brightness_range, color_range

    def __call__(self, results):
        if np.random.rand() < self.prob:
            image = results['img'][:, :, 0:3]
            gamma = np.random.uniform(min(*self.gamma_range), max(*self.gamma_range))
            aug = image ** gamma
            aug = aug * np.random.uniform(min(*self.brightness_range), max(*self.brightness_range))
            colors = np.random.uniform(min(*self.color_range), max(*self.color_range), size=3)
            aug = np.clip(aug * colors.reshape(1, 1, 3), 0, 255)
            results['img'][:, :, 0:3] = aug
        return results

    def __repr__(self):
        return f'{self.__class__.__name__}(prob={self.prob})'


@PIPELINES.register_module()
class Resize:
    """Rescale image (bilinear) and depth fields (nearest).  Modes: ``ratio_range`` on the current size (the training
    configs), a single ``img_scale`` (optionally with ``ratio_range``), or a scale preset in ``results['scale']`` (TTA)."""

    def __init__(self, img_scale=None, multiscale_mode='range', ratio_range=None, keep_ratio=True, normals=False):
        assert not normals
        if img_scale is not None and not isinstance(img_scale, list):
            img_scale = [img_scale]
        if ratio_range is not None:
            assert img_scale is None or len(img_scale) == 1
        else:
            assert multiscale_mode in ('value', 'range')
        self.img_scale, self.multiscale_mode, self.ratio_range, self.keep_ratio = img_scale, multiscale_mode, ratio_range, keep_ratio

    @staticmethod
    def random_sample_ratio(img_scale, ratio_range):
        lo, hi = ratio_range
        assert lo <= hi
        ratio = np.random.random_sample() * (hi - lo) + lo
        return (int(img_scale[0] * ratio), int(img_scale[1] * ratio)), None

    def _random_scale(self, results):
        if self.ratio_range is not None:
            if self.img_scale is None:
                h, w = results['img'].shape[:2]
                scale, idx = self.random_sample_ratio((w, h), self.ratio_range)
            else:
                scale, idx = self.random_sample_ratio(self.img_scale[0], self.ratio_range)
        elif len(self.img_scale) == 1:
            scale, idx = self.img_scale[0], 0
        elif self.multiscale_mode == 'value':
            idx = np.random.randint(len(self.img_scale))
            scale = self.img_scale[idx]
        else:
            long_e = [max(s) for s in self.img_scale]
            short_e = [min(s) for s in self.img_scale]
            scale = (np.random.randint(min(long_e), max(long_e) + 1), np.random.randint(min(short_e), max(short_e) + 1))
            idx = None
        results['scale'], results['scale_idx'] = scale, idx

    def __call__(self, results):
        if 'scale' not in results:
            self._random_scale(results)
        h, w = results['img'].shape[:2]
        if self.keep_ratio:
            img = imrescale(results['img'], results['scale'])
            new_h, new_w = img.shape[:2]
            w_scale, h_scale = new_w / w, new_h / h
        else:
            img, w_scale, h_scale = imresize(results['img'], results['scale'], return_scale=True)
        results['img'] = img
        results['img_shape'] = results['pad_shape'] = img.shape
        results['scale_factor'] = np.array([w_scale, h_scale, w_scale, h_scale], dtype=np.float32)
        results['keep_ratio'] = self.keep_ratio
        for key in results.get('depth_fields', []):
            fn = imrescale if self.keep_ratio else imresize
            results[key] = fn(results[key], results['scale'], interpolation='nearest')
        return results

    def __repr__(self):
        return (f'{self.__class__.__name__}(img_scale={self.img_scale}, multiscale_mode={self.multiscale_mode}, '
                f'ratio_range={self.ratio_range}, keep_ratio={self.keep_ratio})')


@PIPELINES.register_module()
class DDADResize:
    """transforms.py:736-783: colour by area averaging, ground-depth channels by nearest neighbour; the sparse LiDAR depth
    (and slope classes) are re-projected point by point — each valid pixel lands at ``int(coord * scale)`` — so no depth is
    interpolated."""

    def __init__(self, shape, depth=True, USE_DYNAMIC_PE=False):
        self.shape, self.depth, self.USE_DYNAMIC_PE = tuple(shape), depth, USE_DYNAMIC_PE

    def _splat(self, x):
        h, w = x.shape
        ys, xs = np.nonzero(x > 0)
        val = x[ys, xs]
        ys = (ys * (self.shape[0] / h)).astype(np.int32)
        xs = (xs * (self.shape[1] / w)).astype(np.int32)
        keep = (ys < self.shape[0]) & (xs < self.shape[1])
        out = np.zeros(self.shape)
        out[ys[keep], xs[keep]] = val[keep]                 # row-major order: the last source pixel wins, as upstream
        return out

    def __call__(self, results):
        img_pe = results['img']
        size = self.shape[::-1]
        if img_pe.shape[-1] == 5:
            img = imresize_area(img_pe[:, :, 0:3].copy().astype(np.uint8), size)
            pe = imresize(img_pe[:, :, 3].copy().astype(np.float32), size, interpolation='nearest')
            pe_raw = imresize(img_pe[:, :, 4].copy().astype(np.float32), size, interpolation='nearest')
            results['img'] = np.concatenate([img, pe[:, :, None], pe_raw[:, :, None]], axis=-1).astype(np.float32)
        else:
            results['img'] = imresize_area(img_pe, size)
        if self.depth:
            results['depth_gt'] = self._splat(results['depth_gt'])
            if self.USE_DYNAMIC_PE:
                results['pe_k_gt'] = self._splat(results['pe_k_gt'])
        return results

    def __repr__(self):
        return f'{self.__class__.__name__}(shape={self.shape}, depth={self.depth})'
