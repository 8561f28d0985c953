"""KITTI loading transforms: RGB + ground-embedding channels, LiDAR depth + slope classes, intrinsics.

Restates depth/datasets/pipelines/loading.py:27-96 (LoadKITTICamIntrinsic), :98-158 (DepthLoadAnnotations) and
:230-560 (LoadImageFromFile, the USEPE branch the GEDepth configs enable; the commented-out / unused mask options of
the reference are not carried).  Images are decoded with Pillow and converted to the BGR order cv2 would return.
"""
import os.path as osp

import numpy as np
from PIL import Image

from ..builder import PIPELINES
from .imageops import imresize


@PIPELINES.register_module()
class LoadKITTICamIntrinsic:

    def __init__(self, load_surface_normals=False):
        if load_surface_normals:
            raise NotImplementedError('surface normals are not on the GEDepth path')

    def __call__(self, results):
        if 'input' in results['img_prefix']:                  # raw KITTI: the drive date selects the calibration
            date = results['filename'].split('/')[-5]
            results['cam_intrinsic'] = results['cam_intrinsic_dict'][date]
            results['cam_intrinsic_for_normal'] = results['cam_intrinsic_dict_for_nromal'][date]
        else:                                                 # benchmark test images ship a per-image file
            cam_file = results['filename'].replace('benchmark_test', 'benchmark_test_cam').replace('png', 'txt')
            results['cam_intrinsic'] = np.loadtxt(cam_file).reshape(3, 3).tolist()
        return results

    def __repr__(self):
        return self.__class__.__name__


@PIPELINES.register_module()
class DepthLoadAnnotations:
    """depth_gt = uint16 PNG / depth_scale (metres); pe_k_gt = slope class map ``k_img + 5`` (0..10, 255 = ignore)
    resized with nearest neighbour to the depth map (loading.py:136-150)."""

    def __init__(self, LOAD_DYNAMIC_PE=False, file_client_args=None, imdecode_backend='pillow'):
        self.LOAD_DYNAMIC_PE = LOAD_DYNAMIC_PE
        self.imdecode_backend = imdecode_backend

    def __call__(self, results):
        name = results['ann_info']['depth_map']
        filename = osp.join(results['depth_prefix'], name) if results.get('depth_prefix') is not None else name
        depth_gt = np.asarray(Image.open(filename), dtype=np.float32) / results['depth_scale']
        results['depth_gt'] = depth_gt
        results['depth_ori_shape'] = depth_gt.shape
        results['depth_fields'].append('depth_gt')
        if not self.LOAD_DYNAMIC_PE:
            k_file = filename.replace('.png', '.npz').replace('gt_depth', 'slope_range_5_5_interval_1')
            pe_k_gt = np.load(k_file)['k_img'].astype(np.float32) + 5
            pe_k_gt[pe_k_gt == 260] = 255
            pe_k_gt = imresize(pe_k_gt, (depth_gt.shape[1], depth_gt.shape[0]), interpolation='nearest')
            results['pe_k_gt'] = pe_k_gt.astype(np.float32)
            results['depth_fields'].append('pe_k_gt')
        return results

    def __repr__(self):
        return f"{self.__class__.__name__}(imdecode_backend='{self.imdecode_backend}')"


@PIPELINES.register_module()
class LoadImageFromFile:
    """BGR uint8 image; with ``USEPE`` two float channels are appended (loading.py:366-403,470-528): channel 3 = ground
    depth ``pe_165.npy`` of the drive with values outside (0, 200] zeroed, channel 4 = the unfiltered map, and
    ``pe_ori_point`` = its bottom-right value.  ``pe_root`` (default: the image prefix) replaces the reference's hard-coded
    ``data/kitti/input/``."""

    def __init__(self, to_float32=False, color_type='color', file_client_args=None, imdecode_backend='cv2', USEPE=False,
                 LOAD_DYNAMIC_PE=False, pe_root=None, **unused_mask_options):
        self.to_float32, self.color_type, self.imdecode_backend = to_float32, color_type, imdecode_backend
        self.USEPE, self.LOAD_DYNAMIC_PE, self.pe_root = USEPE, LOAD_DYNAMIC_PE, pe_root
        if LOAD_DYNAMIC_PE:
            raise NotImplementedError('LOAD_DYNAMIC_PE (ground depth from GT slopes) is not used by the released configs')

    def _pe(self, results):
        # the reference hard-codes 'data/kitti/input/' (loading.py:375,397) = its img_prefix; follow the configured prefix
        root = self.pe_root if self.pe_root is not None else (results.get('img_prefix') or osp.join('data', 'kitti', 'input'))
        return np.load(osp.join(root, results['ori_filename'].split('/')[0], 'pe', 'pe_165.npy')).astype(np.float32)

    def __call__(self, results):
        name = results['img_info']['filename']
        filename = osp.join(results['img_prefix'], name) if results.get('img_prefix') is not None else name
        img = np.asarray(Image.open(filename).convert('RGB'))[..., ::-1]          # BGR, like cv2.imdecode
        img = np.ascontiguousarray(img)
        if self.to_float32:
            img = img.astype(np.float32)
        results['filename'] = filename
        results['ori_filename'] = name
        if self.USEPE:
            pe_comput = self._pe(results)
            pe = pe_comput.copy()
            pe[pe > 200] = 0
            pe[pe < 0] = 0
            img = np.concatenate((img, pe[..., None], pe_comput[..., None]), axis=-1)     # float32 (H, W, 5)
            results['pe_ori_point'] = pe_comput[-1, -1]
        results['img'] = img
        results['img_shape'] = results['ori_shape'] = results['pad_shape'] = img.shape
        results['scale_factor'] = 1.0
        c = 1 if img.ndim < 3 else img.shape[2]
        results['img_norm_cfg'] = dict(mean=np.zeros(c, dtype=np.float32), std=np.ones(c, dtype=np.float32), to_rgb=False)
        return results

    def __repr__(self):
        return f"{self.__class__.__name__}(to_float32={self.to_float32}, color_type='{self.color_type}', USEPE={self.USEPE})"


_DDAD_CAMERA_HEIGHT = {'CAMERA_01': 1.56, 'CAMERA_05': 1.57, 'CAMERA_06': 1.53, 'CAMERA_09': 1.53}


@PIPELINES.register_module()
class LoadDDADCamIntrinsic:
    """loading.py:958-978: the camera name is the parent directory of the image."""

    def __call__(self, results):
        results['cam_intrinsic'] = results['cam_intrinsic_dict'][results['filename'].split('/')[-2]]
        return results

    def __repr__(self):
        return self.__class__.__name__


@PIPELINES.register_module()
class DDADDepthLoadAnnotations:
    """loading.py:743-801: depth from ``<...>.npz['depth']``; with ``USE_DYNAMIC_PE`` the slope classes from the sibling
    ``*_slope_public_debug.npz`` (``k_img + 5``, 255 kept as ignore)."""

    def __init__(self, USE_DYNAMIC_PE=False, file_client_args=None, imdecode_backend='pillow'):
        self.USE_DYNAMIC_PE, self.imdecode_backend = USE_DYNAMIC_PE, imdecode_backend

    def __call__(self, results):
        filename = results['ann_info']['depth_map']
        depth_gt = np.load(filename)['depth']
        results['depth_gt'] = depth_gt
        results['depth_ori_shape'] = depth_gt.shape
        results['depth_fields'].append('depth_gt')
        if self.USE_DYNAMIC_PE:
            k = np.load(filename.replace('depth_val', 'depth').replace('.npz', '_slope_public_debug.npz'))['k_img'].astype(np.float32)
            ignore = k == 255
            k = k + 5
            k[ignore] = 255
            results['pe_k_gt'] = k.astype(np.float32)
            results['depth_fields'].append('pe_k_gt')
        return results

    def __repr__(self):
        return f"{self.__class__.__name__}(imdecode_backend='{self.imdecode_backend}')"


@PIPELINES.register_module()
class LoadDDADImageFromFile:
    """loading.py:804-955: BGR image + per-camera ground depth ``<pe_root>/<camera>/ddad_pe.npz['pe']`` (channel 3:
    values outside [0, 250] zeroed; channel 4, with ``USE_DYNAMIC_PE``: raw) and the camera height / ``test`` flag the
    adaptive ground embedding needs (encoder_decoder.py:88-94).  ``pe_root`` replaces ``data/DDAD/pe_public_debug``."""

    def __init__(self, to_float32=False, color_type='color', file_client_args=None, imdecode_backend='cv2', USEPE=False,
                 USE_DYNAMIC_PE=False, pe_root=None):
        self.to_float32, self.color_type, self.imdecode_backend = to_float32, color_type, imdecode_backend
        self.USEPE, self.USE_DYNAMIC_PE = USEPE, USE_DYNAMIC_PE
        self.pe_root = pe_root if pe_root is not None else osp.join('data', 'DDAD', 'pe_public_debug')

    def __call__(self, results):
        name = results['img_info']['filename']
        filename = osp.join(results['img_prefix'], name) if results.get('img_prefix') is not None else name
        img = np.ascontiguousarray(np.asarray(Image.open(filename).convert('RGB'))[..., ::-1])
        if self.to_float32:
            img = img.astype(np.float32)
        results['filename'] = filename
        results['ori_filename'] = name
        if self.USEPE:
            camera = results['ann_info']['depth_map'].split('/')[-2]
            pe_raw = np.load(osp.join(self.pe_root, camera, 'ddad_pe.npz'))['pe']
            pe = pe_raw.copy()
            pe[pe > 250] = 0
            pe[pe < 0] = 0
            img = np.concatenate((img, pe[..., None]), axis=-1)
            if self.USE_DYNAMIC_PE:
                img = np.concatenate((img, pe_raw[..., None]), axis=-1)
                results['height'] = next(h for cam, h in _DDAD_CAMERA_HEIGHT.items() if cam in filename)
                results['test'] = 0
        results['img'] = img
        results['img_shape'] = results['ori_shape'] = results['pad_shape'] = img.shape
        results['scale_factor'] = 1.0
        c = 1 if img.ndim < 3 else img.shape[2]
        results['img_norm_cfg'] = dict(mean=np.zeros(c, dtype=np.float32), std=np.ones(c, dtype=np.float32), to_rgb=False)
        return results

    def __repr__(self):
        return f"{self.__class__.__name__}(to_float32={self.to_float32}, color_type='{self.color_type}', USEPE={self.USEPE})"
