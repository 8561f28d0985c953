"""Depth metrics of the evaluation protocol (mirror of depth/core/evaluation/metrics.py:8-100).
Host-side numpy on per-image arrays, exactly as the reference evaluates them (nan-mean over images)."""
from collections import OrderedDict

import numpy as np

METRIC_NAMES = ('a1', 'a2', 'a3', 'abs_rel', 'rmse', 'log_10', 'rmse_log', 'silog', 'sq_rel')


def calculate(gt, pred):
    if gt.shape[0] == 0:
        return (np.nan,) * 9
    ratio = np.maximum(gt / pred, pred / gt)
    a1, a2, a3 = [(ratio < 1.25 ** p).mean() for p in (1, 2, 3)]
    diff = gt - pred
    abs_rel = np.mean(np.abs(diff) / gt)
    sq_rel = np.mean(diff ** 2 / gt)
    rmse = np.sqrt(np.mean(diff ** 2))
    log_diff = np.log(pred) - np.log(gt)
    rmse_log = np.sqrt(np.mean(log_diff ** 2))
    silog = np.sqrt(np.mean(log_diff ** 2) - np.mean(log_diff) ** 2) * 100
    if np.isnan(silog):
        silog = 0
    log_10 = np.mean(np.abs(np.log10(gt) - np.log10(pred)))
    return a1, a2, a3, abs_rel, rmse, log_10, rmse_log, silog, sq_rel


def metrics(gt, pred, min_depth=1e-3, max_depth=80):
    mask = np.logical_and(gt > min_depth, gt < max_depth)
    return calculate(gt[mask], pred[mask])


def eval_metrics(gt, pred, min_depth=1e-3, max_depth=80):
    return dict(zip(METRIC_NAMES, metrics(gt, pred, min_depth, max_depth)))


def pre_eval_to_metrics(pre_eval_results):
    cols = tuple(zip(*pre_eval_results))
    return OrderedDict((name, np.nanmean(cols[i])) for i, name in enumerate(METRIC_NAMES))
