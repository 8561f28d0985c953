"""Evaluation loops (depth/apis/test.py:32-232): run the model with ``return_loss=False`` over a data loader and either
keep the predictions or reduce them to per-image metric tuples on the fly (``pre_eval``)."""
import torch
import torch.distributed as dist

from ...mmrt.runner import get_dist_info


def _to_device(data, device):
    out = {}
    for k, v in data.items():
        if k == 'img_metas':
            out[k] = v
        elif isinstance(v, list):
            out[k] = [t.to(device, non_blocking=True) if torch.is_tensor(t) else t for t in v]
        else:
            out[k] = v.to(device, non_blocking=True) if torch.is_tensor(v) else v
    return out


def single_gpu_test(model, data_loader, pre_eval=False, format_only=False, format_args=None, device=None):
    """Returns a list with one entry per image: the metric tuple (``pre_eval``) or the ``(1, H, W)`` depth map."""
    model.eval()
    dataset = data_loader.dataset
    device = device or next(model.parameters()).device
    results, idx = [], 0
    loader_indices = data_loader.batch_sampler
    for batch_indices, data in zip(loader_indices, data_loader):
        with torch.no_grad():
            result = model(return_loss=False, **_to_device(data, device))
        if format_only:
            result = dataset.format_results(result, indices=batch_indices, **(format_args or {}))
        if pre_eval:
            result, _ = dataset.pre_eval(result, indices=list(batch_indices))
        results.extend(result)
        idx += len(result)
    return results


def multi_gpu_test(model, data_loader, pre_eval=False, format_only=False, format_args=None, device=None):
    """Each rank evaluates its shard of a non-shuffled DistributedSampler; rank 0 receives the results in dataset order."""
    part = single_gpu_test(model, data_loader, pre_eval, format_only, format_args, device)
    rank, world = get_dist_info()
    if world == 1:
        return part
    gathered = [None] * world
    dist.all_gather_object(gathered, part)
    if rank != 0:
        return None
    ordered = []
    for i in range(max(len(g) for g in gathered)):
        for g in gathered:                          # DistributedSampler deals indices round-robin
            if i < len(g):
                ordered.append(g[i])
    return ordered[:len(data_loader.dataset)]
