"""DDAD (front + side cameras) dataset with the ground-embedding channels — restates depth/datasets/ddad.py:31-310:
split lines ``<image> <depth .npz>`` filtered by camera, per-camera intrinsics, and the evaluation protocol (prediction
resized bilinearly, align_corners=True, to the full-resolution ground truth; valid where min_depth < gt < max_depth)."""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import Dataset

from ..core.evaluation import METRIC_NAMES, metrics, pre_eval_to_metrics
from .builder import DATASETS
from .pipelines import Compose

_CAM_K = {
    'CAMERA_01': [[2.1815303e+03, 0.0, 9.2802191e+02, 0], [0.0, 2.1816035e+03, 6.1595679e+02, 0], [0.0, 0.0, 1.0, 0]],
    'CAMERA_05': [[1.0570685e+03, 0.0, 9.6468347e+02, 0], [0.0, 1.0559746e+03, 5.8866125e+02, 0], [0.0, 0.0, 1.0, 0]],
    'CAMERA_06': [[1.0607557e+03, 0.0, 9.4655847e+02, 0], [0.0, 1.0592549e+03, 6.1140710e+02, 0], [0.0, 0.0, 1.0, 0]],
    'CAMERA_09': [[1.0634580e+03, 0.0, 9.4466577e+02, 0], [0.0, 1.0652224e+03, 6.1269843e+02, 0], [0.0, 0.0, 1.0, 0]],
}


@DATASETS.register_module()
class DDADDataset(Dataset):

    def __init__(self, pipeline, cameras=('CAMERA_01', 'CAMERA_05', 'CAMERA_06', 'CAMERA_07', 'CAMERA_08', 'CAMERA_09'),
                 split=None, test_mode=False, garg_crop=False, eigen_crop=False, min_depth=1e-3, max_depth=200):
        self.garg_crop, self.eigen_crop, self.cameras = garg_crop, eigen_crop, list(cameras)
        self.pipeline = Compose(pipeline)
        self.split, self.test_mode, self.min_depth, self.max_depth = split, test_mode, min_depth, max_depth
        self.img_infos = self.load_annotations(split)

    def __len__(self):
        return len(self.img_infos)

    def load_annotations(self, split):
        if split is None:
            raise NotImplementedError('Split should be specified')
        infos = []
        with open(split) as f:
            for line in f:
                parts = line.strip().split(' ')
                if len(parts) < 2:
                    continue
                if parts[1].split('/')[-2] in self.cameras:
                    infos.append(dict(filename=parts[0], ann=dict(depth_map=parts[1].replace('depth_val', 'depth'))))
        return sorted(infos, key=lambda x: x['filename'])

    def get_ann_info(self, idx):
        return self.img_infos[idx]['ann']

    def pre_pipeline(self, results):
        results['depth_fields'] = []
        results['cam_intrinsic_dict'] = {c: [list(r) for r in k] for c, k in _CAM_K.items()}

    def __getitem__(self, idx):
        results = dict(img_info=self.img_infos[idx], ann_info=self.get_ann_info(idx))
        self.pre_pipeline(results)
        return self.pipeline(results)

    prepare_train_img = prepare_test_img = __getitem__

    def format_results(self, results, imgfile_prefix=None, indices=None, **kwargs):
        results[0] = results[0].astype(np.uint16)
        return results

    def get_gt_depth_maps(self):
        for info in self.img_infos:
            yield np.load(info['ann']['depth_map'])['depth']

    def eval_mask(self, depth_gt):
        depth_gt = np.squeeze(depth_gt)
        return np.logical_and(depth_gt > self.min_depth, depth_gt < self.max_depth)[None]

    def pre_eval(self, preds, indices):
        if not isinstance(indices, list):
            indices = [indices]
        if not isinstance(preds, list):
            preds = [preds]
        out_metrics, out_preds = [], []
        for pred, index in zip(preds, indices):
            gt = np.load(self.img_infos[index]['ann']['depth_map'])['depth'].astype(np.float32)
            pred = F.interpolate(torch.from_numpy(np.asarray(pred, dtype=np.float32))[None], size=gt.shape, mode='bilinear',
                                 align_corners=True)[0].numpy()
            gt = gt[None]
            mask = self.eval_mask(gt)
            out_metrics.append(metrics(gt[mask], pred[mask], min_depth=self.min_depth, max_depth=self.max_depth))
            out_preds.append(pred)
        return out_metrics, out_preds

    def evaluate(self, results, metric='eigen', logger=None, **kwargs):
        if len(results) and isinstance(results[0], np.ndarray):
            results = self.pre_eval(list(results), list(range(len(results))))[0]
        ret = pre_eval_to_metrics(results)
        summary = OrderedDict((k, np.round(np.nanmean(v), 4)) for k, v in ret.items())
        text = 'Summary:\n' + ' | '.join(f'{k:>8s}' for k in METRIC_NAMES) + '\n' + ' | '.join(f'{summary[k]:8.4f}' for k in METRIC_NAMES)
        (logger.info if logger is not None and hasattr(logger, 'info') else print)(text)
        return dict(ret)
