"""cv2-free numpy restatements of the ``mmcv.image`` functions the KITTI pipeline calls (mmcv 1.3.13, cv2 backend):
``imflip``, ``imnormalize``, ``rescale_size`` / ``imrescale`` / ``imresize`` and ``imrotate``.  The reference uses them in
depth/datasets/pipelines/transforms.py:30-47 (Normalize), :249-286 (RandomRotate), :321-350 (RandomFlip), :655-693 (Resize)
and loading.py:146 (nearest resize of the slope-class map).

Parity status: cv2 is not installed in the build image, so these are restated from OpenCV's documented sampling rules
(half-pixel-centre bilinear without antialiasing, ``floor(dst * scale)`` nearest, inverse-mapped affine warp with a constant
border, pixel-area averaging) and pinned against INDEPENDENT implementations of those rules in tests/test_imageops_independent.py:
scipy.ndimage.map_coordinates / affine_transform (generic samplers driven by the coordinate rules), PIL's float-mode resize /
rotate / BOX filter, and brute-force supersampling for the area filter.  What OpenCV does beyond the rules is fixed-point
arithmetic — warpAffine quantises source coordinates to 1/32 pixel, uint8 resize uses 11-bit weights — which is not reproduced;
the resulting bound (|slope| / 32 per axis, one grey level) is stated in that test file.
"""
import numpy as np
import torch
import torch.nn.functional as F

_INTERP = ('nearest', 'bilinear')


def imflip(img, direction='horizontal'):
    assert direction in ('horizontal', 'vertical', 'diagonal')
    if direction == 'horizontal':
        return np.flip(img, axis=1)
    if direction == 'vertical':
        return np.flip(img, axis=0)
    return np.flip(img, axis=(0, 1))


def imnormalize(img, mean, std, to_rgb=True):
    """(img[BGR->RGB] - mean) / std in float32 (mmcv multiplies by the float64 reciprocal of std)."""
    img = np.asarray(img).astype(np.float32)
    if to_rgb:
        img = img[..., ::-1]
    mean = np.float64(np.asarray(mean).reshape(1, -1))
    stdinv = 1 / np.float64(np.asarray(std).reshape(1, -1))
    return ((img - mean) * stdinv).astype(np.float32)


def rescale_size(old_size, scale, return_scale=False):
    """mmcv.rescale_size: ``scale`` is a factor or a (long edge, short edge) bound; sizes are (w, h)."""
    w, h = old_size
    if isinstance(scale, (float, int)):
        if scale <= 0:
            raise ValueError(f'Invalid scale {scale}, must be positive.')
        scale_factor = scale
    elif isinstance(scale, tuple):
        max_long_edge, max_short_edge = max(scale), min(scale)
        scale_factor = min(max_long_edge / max(h, w), max_short_edge / min(h, w))
    else:
        raise TypeError(f'Scale must be a number or tuple of int, but got {type(scale)}')
    new_size = (int(w * float(scale_factor) + 0.5), int(h * float(scale_factor) + 0.5))
    return (new_size, scale_factor) if return_scale else new_size


def _resize_axis_bilinear(n_in, n_out):
    src = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
    i0 = np.floor(src).astype(np.int64)
    w1 = (src - i0).astype(np.float32)
    return np.clip(i0, 0, n_in - 1), np.clip(i0 + 1, 0, n_in - 1), w1


def imresize(img, size, return_scale=False, interpolation='bilinear'):
    """Resize to ``size`` = (w, h).  bilinear: half-pixel centres, edge replication, no antialiasing (cv2.INTER_LINEAR);
    nearest: ``src = min(floor(dst * in / out), in - 1)`` (cv2.INTER_NEAREST).  Both rules are exactly those of
    ``F.interpolate(..., align_corners=False)`` / ``mode='nearest'``, whose CPU kernels do the work here (the numpy
    formulation of the same rules, ``_imresize_numpy``, is kept for the tests)."""
    assert interpolation in _INTERP
    h, w = img.shape[:2]
    ow, oh = int(size[0]), int(size[1])
    if interpolation == 'nearest':
        ys = np.minimum(np.floor(np.arange(oh) * (h / oh)).astype(np.int64), h - 1)
        xs = np.minimum(np.floor(np.arange(ow) * (w / ow)).astype(np.int64), w - 1)
        out = img[ys][:, xs]
    else:
        t = torch.from_numpy(np.ascontiguousarray(img if img.ndim == 3 else img[..., None]))
        t = t.permute(2, 0, 1)[None].to(torch.float64 if img.dtype == np.float64 else torch.float32)
        o = F.interpolate(t, size=(oh, ow), mode='bilinear', align_corners=False)[0].permute(1, 2, 0).numpy()
        if img.ndim == 2:
            o = o[..., 0]
        out = np.clip(np.rint(o), 0, 255).astype(np.uint8) if img.dtype == np.uint8 else o.astype(img.dtype)
    if return_scale:
        return out, ow / w, oh / h
    return out


def _imresize_numpy(img, size):
    """The bilinear rule of ``imresize`` spelled out in numpy (reference for the tests)."""
    h, w = img.shape[:2]
    ow, oh = int(size[0]), int(size[1])
    src = img.astype(np.float32) if img.dtype != np.float64 else img
    y0, y1, wy = _resize_axis_bilinear(h, oh)
    x0, x1, wx = _resize_axis_bilinear(w, ow)
    wy = wy.reshape((-1, 1) + (1,) * (img.ndim - 2))
    wx = wx.reshape((1, -1) + (1,) * (img.ndim - 2))
    top = src[y0][:, x0] * (1 - wx) + src[y0][:, x1] * wx
    bot = src[y1][:, x0] * (1 - wx) + src[y1][:, x1] * wx
    return (top * (1 - wy) + bot * wy).astype(img.dtype if img.dtype != np.uint8 else np.float32)


def imrescale(img, scale, return_scale=False, interpolation='bilinear'):
    h, w = img.shape[:2]
    new_size, scale_factor = rescale_size((w, h), scale, return_scale=True)
    out = imresize(img, new_size, interpolation=interpolation)
    return (out, scale_factor) if return_scale else out


def _inverse_rotation(h, w, angle, center, scale):
    """2x2 matrix and offset of dst -> src for cv2.getRotationMatrix2D(center, -angle, scale) + warpAffine (float64)."""
    cx, cy = ((w - 1) * 0.5, (h - 1) * 0.5) if center is None else center
    a = np.deg2rad(-angle)
    alpha, beta = scale * np.cos(a), scale * np.sin(a)
    m = np.array([[alpha, beta, (1 - alpha) * cx - beta * cy], [-beta, alpha, beta * cx + (1 - alpha) * cy]], dtype=np.float64)
    det = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    inv = np.array([[m[1, 1], -m[0, 1]], [-m[1, 0], m[0, 0]]]) / det              # dst(x, y) = src(M^-1 (x, y))
    return inv, -inv @ m[:, 2]


def _rotation_source_coords(h, w, angle, center, scale, dtype=np.float64):
    """Source (x, y) of every destination pixel."""
    inv, off = _inverse_rotation(h, w, angle, center, scale)
    inv, off = inv.astype(dtype), off.astype(dtype)
    xs, ys = np.arange(w, dtype=dtype)[None, :], np.arange(h, dtype=dtype)[:, None]
    return inv[0, 0] * xs + inv[0, 1] * ys + off[0], inv[1, 0] * xs + inv[1, 1] * ys + off[1]


_GRID_CACHE = {}


def _rotation_grid(h, w, angle, center, scale):
    """Normalised (align_corners=True) sampling grid of a rotation; the image, the depth map and the slope classes of one
    sample are rotated with the same parameters, so the last grid is kept."""
    key = (h, w, angle, center, scale)
    if key not in _GRID_CACHE:
        sx, sy = _rotation_source_coords(h, w, angle, center, scale, np.float32)   # matrix in float64, the map in float32
        grid = np.empty((1, h, w, 2), dtype=np.float32)
        grid[0, :, :, 0] = sx * (2.0 / (w - 1)) - 1.0
        grid[0, :, :, 1] = sy * (2.0 / (h - 1)) - 1.0
        _GRID_CACHE.clear()
        _GRID_CACHE[key] = torch.from_numpy(grid)
    return _GRID_CACHE[key]


def imrotate(img, angle, center=None, scale=1.0, border_value=0, interpolation='bilinear', auto_bound=False):
    """Rotate clockwise by ``angle`` degrees about ``center`` (default: the image centre), output size = input size:
    ``cv2.warpAffine(img, cv2.getRotationMatrix2D(center, -angle, scale), (w, h), borderValue=border_value)`` — inverse
    mapping, constant border.  The sampling itself runs in ``F.grid_sample`` (align_corners=True puts its normalised grid on
    pixel centres; zero padding after subtracting the border value gives the constant border); ``_imrotate_numpy`` is the
    same rule in numpy, kept for the tests."""
    assert interpolation in _INTERP
    if auto_bound:
        raise NotImplementedError('auto_bound is not used on the GEDepth path')
    h, w = img.shape[:2]
    if h < 2 or w < 2:
        return _imrotate_numpy(img, angle, center, scale, border_value, interpolation)
    grid = _rotation_grid(h, w, float(angle), None if center is None else tuple(center), float(scale))
    src = img[..., None] if img.ndim == 2 else img
    border = np.broadcast_to(np.asarray(border_value, dtype=np.float32), (src.shape[2],)).reshape(1, -1, 1, 1)
    t = torch.from_numpy(np.ascontiguousarray(src)).permute(2, 0, 1)[None].float() - torch.from_numpy(border.copy())
    o = F.grid_sample(t, grid, mode=interpolation, padding_mode='zeros', align_corners=True) + torch.from_numpy(border.copy())
    out = o[0].permute(1, 2, 0).numpy()
    if img.dtype == np.uint8:
        out = np.clip(np.rint(out), 0, 255)
    out = out.astype(img.dtype)
    return out[..., 0] if img.ndim == 2 else out


def _imrotate_numpy(img, angle, center=None, scale=1.0, border_value=0, interpolation='bilinear'):
    h, w = img.shape[:2]
    sx, sy = _rotation_source_coords(h, w, angle, center, scale)
    sx, sy = np.broadcast_to(sx, (h, w)), np.broadcast_to(sy, (h, w))
    squeeze = img.ndim == 2
    src = img[..., None] if squeeze else img
    border = np.broadcast_to(np.asarray(border_value, dtype=np.float64), (src.shape[2],))

    def fetch(yi, xi):
        inside = (yi >= 0) & (yi < h) & (xi >= 0) & (xi < w)
        v = src[np.clip(yi, 0, h - 1), np.clip(xi, 0, w - 1)].astype(np.float64)
        return np.where(inside[..., None], v, border)

    if interpolation == 'nearest':
        out = fetch(np.floor(sy + 0.5).astype(np.int64), np.floor(sx + 0.5).astype(np.int64))
    else:
        x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
        fx, fy = (sx - x0)[..., None], (sy - y0)[..., None]
        out = (fetch(y0, x0) * (1 - fx) + fetch(y0, x0 + 1) * fx) * (1 - fy) + \
              (fetch(y0 + 1, x0) * (1 - fx) + fetch(y0 + 1, x0 + 1) * fx) * fy
    if img.dtype == np.uint8:
        out = np.clip(np.rint(out), 0, 255)
    out = out.astype(img.dtype)
    return out[..., 0] if squeeze else out


def _area_weights(n_in, n_out):
    """(n_out, n_in) row-stochastic matrix of cv2.INTER_AREA when shrinking: output cell j averages the source interval
    [j * s, (j + 1) * s), s = n_in / n_out, each source pixel weighted by its overlap."""
    s = n_in / n_out
    w = np.zeros((n_out, n_in), dtype=np.float64)
    for j in range(n_out):
        lo, hi = j * s, (j + 1) * s
        i0, i1 = int(np.floor(lo)), min(int(np.ceil(hi)), n_in)
        for i in range(i0, i1):
            w[j, i] = min(hi, i + 1) - max(lo, i)
    return w / w.sum(1, keepdims=True)


def imresize_area(img, size):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_AREA) for down-scaling (pixel-area averaging); up-scaling falls
    back to bilinear like OpenCV's INTER_AREA does for factors < 1 per axis."""
    h, w = img.shape[:2]
    ow, oh = int(size[0]), int(size[1])
    if ow > w or oh > h:
        return imresize(img, size, interpolation='bilinear')
    src = img.astype(np.float64)
    out = np.tensordot(_area_weights(h, oh), src, axes=(1, 0))                    # (oh, w, ...)
    out = np.moveaxis(np.tensordot(_area_weights(w, ow), out, axes=(1, 1)), 0, 1)  # (oh, ow, ...)
    if img.dtype == np.uint8:
        return np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return out.astype(img.dtype)
