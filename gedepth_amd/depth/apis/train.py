"""``train_depther`` — mirror of depth/apis/train.py:28-121: data loader(s) -> DDP wrap -> optimizer ->
IterBasedRunner + hooks -> (resume/load) -> run."""
import random

import numpy as np
import torch

from ...mmrt.ddp import FlatDDP
from ...mmrt.optim import build_optimizer
from ...mmrt.runner import EvalHook, IterBasedRunner, get_dist_info


def set_random_seed(seed, deterministic=False):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def train_depther(model, dataset, cfg, distributed=False, validate=False, timestamp=None, meta=None, logger=print,
                  evaluate_fn=None, data_loaders=None, device=None, batch_transform=None, channels_last=False):
    """``dataset``: a map-style dataset (or list of them) yielding dict samples; alternatively pass ready
    ``data_loaders``.  ``evaluate_fn(runner) -> dict`` plays DistEvalHook's role when ``validate``."""
    device = device or torch.device('cuda', torch.cuda.current_device())
    if data_loaders is None:
        from ..datasets.loader import build_dataloader
        datasets = dataset if isinstance(dataset, (list, tuple)) else [dataset]
        data_loaders = [build_dataloader(ds, cfg.data.samples_per_gpu, cfg.data.workers_per_gpu, dist=distributed,
                                         seed=cfg.get('seed'), drop_last=True) for ds in datasets]
    model = model.to(device)
    if channels_last:                                           # before the optimizer: the parameter arena keeps the NHWC weight order
        from ..models.utils import to_channels_last
        to_channels_last(model)
    optimizer = build_optimizer(model, cfg.optimizer, cfg.get('optimizer_config', {}).get('grad_clip'))
    wrapped = FlatDDP(model, optimizer.arena) if distributed else model
    if cfg.get('runner') is None:
        cfg.runner = dict(type='IterBasedRunner', max_iters=cfg.total_iters)
    assert cfg.runner['type'] == 'IterBasedRunner'
    # numerics follow the config: the reference trains in fp32 (fp16_enabled = False, depth/models/depther/base.py:20), so a
    # reference config run through this drop-in stays fp32; bf16 autocast is an explicit choice (cfg.amp = 'bf16' / --bf16)
    amp_name = cfg.get('amp', 'fp32')
    if amp_name not in ('fp32', 'bf16'):
        raise ValueError(f"cfg.amp must be 'fp32' (the reference's precision, default) or 'bf16' (autocast + bf16 shadow weights), got {amp_name!r}")
    logger(f'precision: {amp_name}' + (' (bf16 autocast, fp32 master weights)' if amp_name == 'bf16' else ' (reference precision)'))
    amp = torch.bfloat16 if amp_name == 'bf16' else None
    runner = IterBasedRunner(wrapped, optimizer, work_dir=cfg.get('work_dir'), logger=logger, meta=meta,
                             max_iters=cfg.runner['max_iters'], amp_dtype=amp, hip_graph=bool(cfg.get('hip_graph', False)))
    runner.batch_transform = batch_transform
    if timestamp:
        runner.timestamp = timestamp
    runner.register_training_hooks(cfg.lr_config, dict(grad_clip=cfg.get('optimizer_config', {}).get('grad_clip')),
                                   cfg.get('checkpoint_config'), cfg.get('log_config'))
    if validate and evaluate_fn is not None:
        runner.register_hook(EvalHook(evaluate_fn, **dict(cfg.get('evaluation', {}))))
    if cfg.get('resume_from'):
        runner.resume(cfg.resume_from)
    elif cfg.get('load_from'):
        runner.load_checkpoint(cfg.load_from)
    runner.run(data_loaders, cfg.get('workflow', [('train', 1)]))
    return runner
