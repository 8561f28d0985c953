"""The cv2-free image operations of the KITTI / DDAD pipelines (gedepth_amd/depth/datasets/pipelines/imageops.py: restatements of
mmcv.imresize / imrescale / imrotate / INTER_AREA resize, reference depth/datasets/pipelines/transforms.py:485-783) against
implementations that are NOT this repository: scipy.ndimage (map_coordinates / affine_transform: generic samplers driven by OpenCV's
documented coordinate rules), PIL (Image.resize / Image.rotate in float mode) and brute-force integration for the area filter.
cv2 itself is not in the image; what it does beyond these rules is fixed-point arithmetic: warpAffine quantises source coordinates to
1/32 pixel (bound stated in test_rotate_fixed_point_bound) and uint8 resize uses 11-bit weights (at most one grey level)."""
import numpy as np
import pytest
from PIL import Image
from scipy import ndimage

from gedepth_amd.depth.datasets.pipelines import imageops as I


def _img(h, w, c=None, seed=0, dtype=np.float32):
    rs = np.random.RandomState(seed)
    a = rs.rand(h, w) if c is None else rs.rand(h, w, c)
    return (a * 255).astype(dtype)


def _cv_linear_coords(n_in, n_out):
    """cv2.INTER_LINEAR source coordinate of destination index d: (d + 0.5) * in / out - 0.5, taps clamped to the image."""
    return (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5


@pytest.mark.parametrize('hw,size', [((11, 17), (34, 22)), ((37, 53), (20, 13)), ((375, 1242), (1216, 352)), ((9, 9), (9, 14))])
def test_bilinear_resize_vs_scipy(hw, size):
    """map_coordinates(order=1, mode='nearest') evaluates the piecewise-linear interpolant with edge replication at arbitrary
    coordinates: fed with OpenCV's half-pixel-centre rule it IS cv2.INTER_LINEAR (no antialiasing), up- and down-scaling."""
    img = _img(*hw, seed=1)
    ow, oh = size
    ys, xs = np.meshgrid(_cv_linear_coords(hw[0], oh), _cv_linear_coords(hw[1], ow), indexing='ij')
    ref = ndimage.map_coordinates(img.astype(np.float64), [ys, xs], order=1, mode='nearest')
    got = I.imresize(img, size, interpolation='bilinear')
    # source coordinates are fp32 inside F.interpolate: 1e-4 px at x ~ 1200, times a grey-level slope of up to 255 per pixel
    assert got.shape == (oh, ow) and np.abs(got - ref).max() <= (5e-2 if max(hw) > 1000 else 3e-3)
    got3 = I.imresize(np.stack([img, img[::-1], img.T[:hw[0], :hw[1]] if hw[0] == hw[1] else img * 0.5], -1), size)
    assert np.abs(got3[..., 0] - ref).max() <= (5e-2 if max(hw) > 1000 else 3e-3)


@pytest.mark.parametrize('hw,size', [((11, 17), (34, 22)), ((20, 30), (61, 47))])
def test_bilinear_upscale_vs_pil(hw, size):
    """PIL's BILINEAR in float mode is the same triangle filter on half-pixel centres when ENLARGING (it widens the filter when
    shrinking, unlike cv2 — so only enlargements are compared)."""
    img = _img(*hw, seed=2)
    ref = np.asarray(Image.fromarray(img, mode='F').resize(size, Image.BILINEAR))
    assert np.abs(I.imresize(img, size) - ref).max() <= 2e-3


def test_uint8_resize_rounding_and_rescale_size():
    img = _img(23, 31, 3, seed=3, dtype=np.uint8)
    ys, xs = np.meshgrid(_cv_linear_coords(23, 40), _cv_linear_coords(31, 50), indexing='ij')
    ref = np.stack([ndimage.map_coordinates(img[..., c].astype(np.float64), [ys, xs], order=1, mode='nearest') for c in range(3)], -1)
    got = I.imresize(img, (50, 40))
    assert got.dtype == np.uint8 and np.abs(got.astype(np.float64) - ref).max() <= 0.5 + 1e-6          # round-to-nearest of the exact value
    assert I.rescale_size((1242, 375), 0.5) == (621, 188) and I.rescale_size((100, 50), (200, 80)) == (160, 80)


@pytest.mark.parametrize('angle', [2.5, -2.5, 17.0, 90.0])
@pytest.mark.parametrize('hw', [(24, 40), (37, 53)])
def test_rotate_vs_scipy_affine_and_pil(angle, hw):
    """mmcv.imrotate(img, angle) = cv2.warpAffine(img, getRotationMatrix2D(centre, -angle, 1), ...): clockwise by `angle` about
    ((w-1)/2, (h-1)/2), inverse mapping, bilinear, constant border.  scipy: affine_transform with the explicit inverse matrix in
    (row, col) order; PIL: Image.rotate(-angle) about its default centre (w/2, h/2 in corner coordinates = the same point)."""
    h, w = hw
    img = _img(h, w, seed=4)
    got = I.imrotate(img, angle, border_value=0)
    a = np.deg2rad(-angle)
    # getRotationMatrix2D(center, ang = -angle): M = [[cos, sin, ...], [-sin, cos, ...]] maps src -> dst; dst -> src is its inverse
    cx, cy = (w - 1) / 2, (h - 1) / 2
    M = np.array([[np.cos(a), np.sin(a)], [-np.sin(a), np.cos(a)]])
    Minv = np.linalg.inv(M)                                       # (x, y) order
    A = np.array([[Minv[1, 1], Minv[1, 0]], [Minv[0, 1], Minv[0, 0]]])      # (row, col) order for scipy
    off = np.array([cy, cx]) - A @ np.array([cy, cx])
    # 'grid-constant': taps outside the image take the border value and ARE interpolated with (cv2's BORDER_CONSTANT; scipy's plain
    # 'constant' returns cval for every sample beyond the edge instead)
    ref = ndimage.affine_transform(img.astype(np.float64), A, offset=off, order=1, mode='grid-constant', cval=0.0)
    assert np.abs(got - ref).max() <= 2e-2, np.abs(got - ref).max()
    pil = np.asarray(Image.fromarray(img, mode='F').rotate(-angle, resample=Image.BILINEAR))
    covered = ndimage.affine_transform(np.ones((h, w)), A, offset=off, order=1, mode='constant', cval=0.0) > 0.999
    assert np.abs(got - pil)[covered & (np.arange(h)[:, None] > 1) & (np.arange(h)[:, None] < h - 2)].max() <= 0.5      # PIL: fixed-point coordinates (1/65536), float taps


def test_rotate_fixed_point_bound():
    """What cv2.warpAffine does beyond the rule above: source coordinates are rounded to 1/32 pixel (INTER_BITS = 5).  The resulting
    difference is bounded by max|gradient| / 32 per axis; stated here for the record on a smooth ramp where it is exact."""
    h, w = 32, 48
    img = (np.arange(w, dtype=np.float32)[None, :] * 2.0 + np.arange(h, dtype=np.float32)[:, None] * 3.0)
    got = I.imrotate(img, 2.5, border_value=0)
    sx, sy = I._rotation_source_coords(h, w, 2.5, None, 1.0)
    q = lambda t: np.round(t * 32) / 32
    inside = (sx > 0) & (sx < w - 1) & (sy > 0) & (sy < h - 1)
    exact = 2.0 * sx + 3.0 * sy
    quant = 2.0 * q(sx) + 3.0 * q(sy)
    assert np.abs(got - exact)[inside].max() <= 1e-3
    assert np.abs(quant - exact)[inside].max() <= (2.0 + 3.0) / 64 + 1e-9


@pytest.mark.parametrize('hw,size', [((12, 18), (6, 4)), ((13, 21), (7, 5)), ((1216, 1936), (640, 384))])
def test_area_resize_vs_supersampling_and_pil_box(hw, size):
    """cv2.INTER_AREA when shrinking = the average of the piecewise-constant image over each destination cell.  Brute force: repeat
    every source pixel n_out times per axis, then average blocks of n_in — exact for any ratio; PIL's BOX filter for integer ratios."""
    h, w = hw
    ow, oh = size
    img = _img(h, w, seed=5, dtype=np.float64) if h < 100 else _img(h, w, seed=5, dtype=np.float32)
    got = I.imresize_area(img, size)
    if h < 100:
        fine = np.repeat(np.repeat(img, oh, axis=0), ow, axis=1)                 # (h * oh, w * ow)
        ref = fine.reshape(oh, h, ow, w).mean(axis=(1, 3))
        assert np.abs(got - ref).max() <= 1e-9
    if h % oh == 0 and w % ow == 0:
        box = np.asarray(Image.fromarray(img.astype(np.float32), mode='F').resize(size, Image.BOX))
        assert np.abs(got - box).max() <= 1e-3
    else:                                                                           # DDAD 1216 x 1936 -> 384 x 640: row sums of the weights
        assert abs(got.mean() - img.mean()) <= 1e-3 * img.mean()


def test_nearest_rule():
    """cv2.INTER_NEAREST: src = min(floor(dst * in / out), in - 1) (NOT centre-based, unlike PIL / scipy): spelled out."""
    img = np.arange(7 * 5).reshape(7, 5).astype(np.float32)
    got = I.imresize(img, (12, 16), interpolation='nearest')
    for y in range(16):
        for x in range(12):
            assert got[y, x] == img[min(int(np.floor(y * 7 / 16)), 6), min(int(np.floor(x * 5 / 12)), 4)]
