"""KITTI Eigen-split dataset for monocular depth with the ground-embedding channels.

Restates depth/datasets/kitti.py:102-620 for the path the GEDepth configs use: split-file parsing (pairs with depth
``None`` are dropped, entries sorted by file name), the per-date calibration tables handed to the pipeline, train / test
sample preparation, and the evaluation protocol (KB crop of the ground truth, Garg / Eigen crop mask, per-image metrics,
nan-mean summary).  The reference's experimental ``mask_pe`` / ``mask_pe_gt`` evaluation variants are not carried.
"""
import os.path as osp
from collections import OrderedDict

import numpy as np
from PIL import Image
from torch.utils.data import Dataset

from ..core.evaluation import METRIC_NAMES, metrics, pre_eval_to_metrics
from .builder import DATASETS
from .pipelines import Compose

# P_rect_02 of the five recording days (kitti.py:171-195 / :260-293); 3x4, the 3x3 left block is K
_P_RECT = {
    '2011_09_26': [[7.215377e+02, 0.0, 6.095593e+02, 4.485728e+01], [0.0, 7.215377e+02, 1.728540e+02, 2.163791e-01],
                   [0.0, 0.0, 1.0, 2.745884e-03]],
    '2011_09_28': [[7.070493e+02, 0.0, 6.040814e+02, 4.575831e+01], [0.0, 7.070493e+02, 1.805066e+02, -3.454157e-01],
                   [0.0, 0.0, 1.0, 4.981016e-03]],
    '2011_09_29': [[7.183351e+02, 0.0, 6.003891e+02, 4.450382e+01], [0.0, 7.183351e+02, 1.815122e+02, -5.951107e-01],
                   [0.0, 0.0, 1.0, 2.616315e-03]],
    '2011_09_30': [[7.070912e+02, 0.0, 6.018873e+02, 4.688783e+01], [0.0, 7.070912e+02, 1.831104e+02, 1.178601e-01],
                   [0.0, 0.0, 1.0, 6.203223e-03]],
    '2011_10_03': [[7.188560e+02, 0.0, 6.071928e+02, 4.538225e+01], [0.0, 7.188560e+02, 1.852157e+02, -1.130887e-01],
                   [0.0, 0.0, 1.0, 3.779761e-03]],
}


@DATASETS.register_module()
class KITTIDataset(Dataset):
    """Layout (reference docstring, kitti.py:103-134): ``data_root/input/<date>/<drive>/image_02/data/*.png``,
    ``data_root/gt_depth/<drive>/proj_depth/groundtruth/image_02/*.png`` (uint16, metres * 256), split lines
    ``<image> <depth|None> <focal>``; plus ``input/<date>/pe/pe_165.npy`` and ``slope_range_5_5_interval_1/...npz``."""

    def __init__(self, pipeline, img_dir, ann_dir=None, split=None, data_root=None, test_mode=False, depth_scale=256,
                 garg_crop=True, eigen_crop=False, min_depth=1e-3, max_depth=80, mask_pe=False, mask_pe_gt=False):
        if mask_pe or mask_pe_gt:
            raise NotImplementedError('mask_pe / mask_pe_gt evaluation variants are outside the GEDepth hot path')
        self.pipeline = Compose(pipeline)
        self.img_dir, self.ann_dir, self.split, self.data_root = img_dir, ann_dir, split, data_root
        self.test_mode, self.depth_scale = test_mode, depth_scale
        self.garg_crop, self.eigen_crop, self.min_depth, self.max_depth = garg_crop, eigen_crop, min_depth, max_depth
        if self.data_root is not None:
            if not (self.img_dir is None or osp.isabs(self.img_dir)):
                self.img_dir = osp.join(self.data_root, self.img_dir)
            if not (self.ann_dir is None or osp.isabs(self.ann_dir)):
                self.ann_dir = osp.join(self.data_root, self.ann_dir)
        self.img_infos = self.load_annotations(self.img_dir, self.ann_dir, self.split)

    def __len__(self):
        return len(self.img_infos)

    def load_annotations(self, img_dir, ann_dir, split):
        if split is None:
            raise NotImplementedError('Split should be specified')
        self.invalid_depth_num = 0
        infos = []
        with open(split) as f:
            for line in f:
                parts = line.strip().split(' ')
                if not parts or not parts[0]:
                    continue
                info = dict()
                if ann_dir is not None:
                    if parts[1] == 'None':
                        self.invalid_depth_num += 1
                        continue
                    info['ann'] = dict(depth_map=parts[1])
                info['filename'] = parts[0]
                infos.append(info)
        return sorted(infos, key=lambda x: x['filename'])

    def get_ann_info(self, idx):
        return self.img_infos[idx]['ann']

    def pre_pipeline(self, results):
        results['depth_fields'] = []
        results['img_prefix'] = self.img_dir
        results['depth_prefix'] = self.ann_dir
        results['depth_scale'] = self.depth_scale
        results['cam_intrinsic_dict_for_nromal'] = {d: np.array(p)[:, :3] for d, p in _P_RECT.items()}
        results['cam_intrinsic_dict'] = {d: [list(r) for r in p] for d, p in _P_RECT.items()}

    def __getitem__(self, idx):
        results = dict(img_info=self.img_infos[idx], ann_info=self.get_ann_info(idx))
        self.pre_pipeline(results)
        return self.pipeline(results)

    prepare_train_img = prepare_test_img = __getitem__

    def format_results(self, results, imgfile_prefix=None, indices=None, **kwargs):
        results[0] = (results[0] * self.depth_scale).astype(np.uint16)
        return results

    def _gt(self, index):
        path = osp.join(self.ann_dir, self.img_infos[index]['ann']['depth_map'])
        return np.asarray(Image.open(path), dtype=np.float32) / self.depth_scale

    def get_gt_depth_maps(self):
        for i in range(len(self)):
            yield self._gt(i)

    @staticmethod
    def eval_kb_crop(depth_gt):
        h, w = depth_gt.shape
        top, left = int(h - 352), int((w - 1216) / 2)
        return depth_gt[top:top + 352, left:left + 1216][None]

    def eval_mask(self, depth_gt):
        depth_gt = np.squeeze(depth_gt)
        valid = np.logical_and(depth_gt > self.min_depth, depth_gt < self.max_depth)
        if self.garg_crop or self.eigen_crop:
            gh, gw = depth_gt.shape
            crop = np.zeros(valid.shape)
            if self.garg_crop:
                crop[int(0.40810811 * gh):int(0.99189189 * gh), int(0.03594771 * gw):int(0.96405229 * gw)] = 1
            else:
                crop[int(0.3324324 * gh):int(0.91351351 * gh), int(0.0359477 * gw):int(0.96405229 * gw)] = 1
            valid = np.logical_and(valid, crop)
        return valid[None]

    def pre_eval(self, preds, indices):
        """Per-image metric tuples for predictions ``(1, 352, 1216)`` (kitti.py:502-552)."""
        if not isinstance(indices, list):
            indices = [indices]
        if not isinstance(preds, list):
            preds = [preds]
        out_metrics, out_preds = [], []
        for pred, index in zip(preds, indices):
            gt = self.eval_kb_crop(self._gt(index))
            mask = self.eval_mask(gt)
            out_metrics.append(metrics(gt[mask], pred[mask], min_depth=self.min_depth, max_depth=self.max_depth))
            out_preds.append(pred)
        return out_metrics, out_preds

    def evaluate(self, results, metric='eigen', logger=None, **kwargs):
        """results: per-image metric tuples from ``pre_eval`` or a list of predicted depth maps."""
        if len(results) and isinstance(results[0], np.ndarray):
            pre = []
            for gt, pred in zip(self.get_gt_depth_maps(), results):
                gt = self.eval_kb_crop(gt)
                mask = self.eval_mask(gt)
                pre.append(metrics(gt[mask], pred[mask], min_depth=self.min_depth, max_depth=self.max_depth))
            results = pre
        ret = pre_eval_to_metrics(results)
        summary = OrderedDict((k, np.round(np.nanmean(v), 4)) for k, v in ret.items())
        text = 'Summary:\n' + ' | '.join(f'{k:>8s}' for k in METRIC_NAMES) + '\n' + ' | '.join(f'{summary[k]:8.4f}' for k in METRIC_NAMES)
        (logger.info if logger is not None and hasattr(logger, 'info') else print)(text)
        return dict(ret)
