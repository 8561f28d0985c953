"""Iteration-based training loop and hooks.

Restates the parts of ``mmcv.runner`` the reference drives through depth/apis/train.py:81-121 with the configs'
settings (configs/depthformer/depthformer_v.py:141-166): ``IterBasedRunner`` with the hot loop
``next(loader) -> model.train_step -> zero_grad/backward/clip/step -> lr update -> log/ckpt/eval hooks``.
The optimizer hook's clip + AdamW is one fused launch (mmrt/optim.py), the gradient exchange is
mmrt/ddp.py, and logging reads device scalars only every ``interval`` iterations (no per-step host sync).
"""
import json
import os
import os.path as osp
import time
from collections import OrderedDict

import torch
import torch.distributed as dist

from .checkpoint import load_checkpoint, save_checkpoint
from .optim import CosineAnnealingLr


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class Hook:
    def before_run(self, runner): pass
    def after_run(self, runner): pass
    def before_train_iter(self, runner): pass
    def after_train_iter(self, runner): pass

    def every_n_iters(self, runner, n):
        return (runner.iter + 1) % n == 0 if n > 0 else False


class LrUpdaterHook(Hook):
    """Sets the learning rate before every iteration (policy 'CosineAnnealing', linear warm-up)."""

    def __init__(self, policy='CosineAnnealing', **kwargs):
        if policy != 'CosineAnnealing':
            raise NotImplementedError(f'lr policy {policy}: the GEDepth configs use CosineAnnealing')
        self.kwargs = kwargs
        self.schedule = None

    def before_run(self, runner):
        base_lr = runner.optimizer.defaults['lr']
        self.schedule = CosineAnnealingLr(base_lr, runner.max_iters, **self.kwargs)

    def before_train_iter(self, runner):
        runner.current_lr = self.schedule.apply(runner.optimizer, runner.iter)


class OptimizerHook(Hook):
    """zero_grad -> backward -> (all-reduce) -> clip + step.  ``grad_clip`` is applied inside the fused AdamW
    kernel (mmrt/optim.py); autocast(bf16) wraps the forward in the runner."""

    def __init__(self, grad_clip=None):
        self.grad_clip = grad_clip

    def after_train_iter(self, runner):
        if getattr(runner, 'graphed', None) is not None:       # backward, exchange and optimizer ran inside the captured step
            return
        runner.outputs['loss'].backward()
        if hasattr(runner.model, 'finish'):
            runner.model.finish()
        runner.optimizer.step()


class CheckpointHook(Hook):

    def __init__(self, interval=-1, by_epoch=False, max_keep_ckpts=-1, out_dir=None, save_optimizer=True, **kwargs):
        assert not by_epoch
        self.interval, self.max_keep, self.out_dir, self.save_optimizer = interval, max_keep_ckpts, out_dir, save_optimizer

    def after_train_iter(self, runner):
        if not self.every_n_iters(runner, self.interval):
            return
        rank, _ = get_dist_info()
        if rank == 0:
            out_dir = self.out_dir or runner.work_dir
            path = osp.join(out_dir, f'iter_{runner.iter + 1}.pth')
            runner.save_checkpoint(path, save_optimizer=self.save_optimizer)
            if self.max_keep > 0:
                old = runner.iter + 1 - self.max_keep * self.interval
                while old > 0:
                    p = osp.join(out_dir, f'iter_{old}.pth')
                    if not osp.exists(p):
                        break
                    os.remove(p)
                    old -= self.interval
        if dist.is_available() and dist.is_initialized():
            dist.barrier()


class TextLoggerHook(Hook):
    """Every ``interval`` iterations: averaged log vars + lr + time -> logger and ``<work_dir>/<ts>.log.json``."""

    def __init__(self, by_epoch=False, interval=10, **kwargs):
        self.interval = interval
        self.buffer = []
        self.t0 = None
        self.json_path = None

    def before_run(self, runner):
        self.t0 = time.time()
        if runner.work_dir and runner.rank == 0:
            self.json_path = osp.join(runner.work_dir, f'{runner.timestamp}.log.json')

    def after_train_iter(self, runner):
        self.buffer.append(runner.outputs['log_vars'])
        if not self.every_n_iters(runner, self.interval):
            return
        keys = list(self.buffer[0].keys())
        avg = OrderedDict((k, sum(b[k] for b in self.buffer) / len(self.buffer)) for k in keys)   # host sync happens here only
        now = time.time()
        rec = OrderedDict(mode='train', iter=runner.iter + 1, lr=runner.current_lr,
                          time=(now - self.t0) / len(self.buffer), **avg)
        self.buffer, self.t0 = [], now
        if runner.rank == 0:
            msg = ', '.join(f'{k}: {v:.5g}' if isinstance(v, float) else f'{k}: {v}' for k, v in rec.items())
            runner.logger(f'Iter [{runner.iter + 1}/{runner.max_iters}] {msg}')
            if self.json_path:
                with open(self.json_path, 'a') as f:
                    f.write(json.dumps(rec) + '\n')
        runner.log_buffer = rec


class TensorboardLoggerHook(Hook):
    """Accepted for config compatibility; scalars go to the json log (no tensorboard dependency in the image)."""

    def __init__(self, **kwargs):
        pass


class EvalHook(Hook):
    """Every ``interval`` iterations run ``evaluate_fn(runner) -> dict`` and keep the best checkpoint by
    ``save_best`` with rule less/greater (depth/core/evaluation/eval_hooks.py:9-118, configs :152-159)."""

    def __init__(self, evaluate_fn, interval=800, by_epoch=False, save_best=None, rule='less', start=0, broadcast_bn_buffer=True,
                 **kwargs):
        self.evaluate_fn, self.interval, self.save_best, self.rule, self.start = evaluate_fn, interval, save_best, rule, start
        self.broadcast_bn_buffer = broadcast_bn_buffer
        self.best = None

    @staticmethod
    def broadcast_bn_buffers(model):
        """BatchNorm statistics stay per GPU during training (no SyncBN, like the reference); before a distributed evaluation
        rank 0's running_var / running_mean are broadcast so that every rank evaluates the same model
        (depth/core/evaluation/eval_hooks.py:75-87).  Returns the number of BatchNorm modules touched."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return 0
        n = 0
        for module in model.modules():
            if isinstance(module, torch.nn.modules.batchnorm._BatchNorm) and module.track_running_stats:
                dist.broadcast(module.running_var, 0)
                dist.broadcast(module.running_mean, 0)
                n += 1
        return n

    def after_train_iter(self, runner):
        if not self.every_n_iters(runner, self.interval) or runner.iter + 1 < self.start:
            return
        if self.broadcast_bn_buffer:
            self.broadcast_bn_buffers(runner.model)
        metrics = self.evaluate_fn(runner)
        if runner.rank == 0 and metrics:
            runner.logger('Eval ' + ', '.join(f'{k}: {v:.4f}' for k, v in metrics.items()))
            if self.save_best and self.save_best in metrics:
                v = metrics[self.save_best]
                better = self.best is None or (v < self.best if self.rule == 'less' else v > self.best)
                if better:
                    self.best = v
                    runner.save_checkpoint(osp.join(runner.work_dir, f'best_{self.save_best}_iter_{runner.iter + 1}.pth'))


HOOKS = dict(TextLoggerHook=TextLoggerHook, TensorboardLoggerHook=TensorboardLoggerHook)


class IterBasedRunner:

    def __init__(self, model, optimizer, work_dir=None, logger=print, meta=None, max_iters=None, amp_dtype=None, hip_graph=False):
        """``hip_graph``: capture forward + losses + backward + gradient exchange + clip + AdamW in one hipGraph after three eager
        iterations and replay it (mmrt/graph.py) — one launch per iteration instead of 1250 - 2400; needs batches of a fixed shape
        (the reference's loaders use drop_last=True)."""
        self.hip_graph, self.graphed = bool(hip_graph), None
        self.model, self.optimizer, self.work_dir, self.logger, self.meta = model, optimizer, work_dir, logger, meta or {}
        self.max_iters = max_iters
        self.iter = 0
        self.epoch = 0
        self.batch_transform = None
        self.hooks = []
        self.outputs = None
        self.current_lr = optimizer.defaults['lr']
        self.rank, self.world_size = get_dist_info()
        self.timestamp = time.strftime('%Y%m%d_%H%M%S', time.localtime())
        self.amp_dtype = amp_dtype
        self.log_buffer = {}
        if work_dir and self.rank == 0:
            os.makedirs(work_dir, exist_ok=True)

    def register_hook(self, hook):
        self.hooks.append(hook)

    def register_training_hooks(self, lr_config, optimizer_config=None, checkpoint_config=None, log_config=None):
        lr_cfg = dict(lr_config)
        self.register_hook(LrUpdaterHook(**lr_cfg))
        self.register_hook(OptimizerHook(**(optimizer_config or {})))
        if checkpoint_config:
            self.register_hook(CheckpointHook(**checkpoint_config))
        if log_config:
            for h in log_config.get('hooks', []):
                h = dict(h)
                typ = h.pop('type')
                if typ not in HOOKS:
                    raise KeyError(f'{typ} is not a known logger hook')
                self.register_hook(HOOKS[typ](interval=log_config.get('interval', 10), **h))

    def call_hook(self, name):
        for h in self.hooks:
            getattr(h, name)(self)

    @staticmethod
    def _set_epoch(loader, epoch):
        sampler = getattr(loader, 'sampler', None)
        if hasattr(sampler, 'set_epoch'):
            sampler.set_epoch(epoch)
        batch_sampler = getattr(loader, 'batch_sampler', None)
        if hasattr(getattr(batch_sampler, 'sampler', None), 'set_epoch') and batch_sampler.sampler is not sampler:
            batch_sampler.sampler.set_epoch(epoch)

    def _to_device(self, batch, device):
        return {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}

    def run(self, data_loaders, workflow=(('train', 1),), **kwargs):
        assert len(data_loaders) == 1 and workflow[0][0] == 'train'
        loader = data_loaders[0]
        device = next(self.model.parameters()).device
        self.model.train()
        self._refresh_shadow()                  # parameters may have been written out of band since build_optimizer (init, load_from)
        self.call_hook('before_run')
        # mmcv IterLoader semantics: an epoch counter that re-seeds the DistributedSampler on every restart of the loader
        # (without set_epoch every epoch replays the same permutation and per-rank shard); on resume it is derived from
        # the iteration count
        per_epoch = max(1, len(loader)) if hasattr(loader, '__len__') else 1
        self.epoch = self.iter // per_epoch
        self._set_epoch(loader, self.epoch)
        it = iter(loader)
        while self.iter < self.max_iters:
            try:
                batch = next(it)
            except StopIteration:
                self.epoch += 1
                self._set_epoch(loader, self.epoch)
                it = iter(loader)
                batch = next(it)
            # batch_transform: a device-side data pipeline (depth/datasets/gpu_pipeline.py) turns the loader's raw samples into
            # the batch dict; otherwise the collated host batch is staged with non-blocking copies
            batch = self.batch_transform(batch) if self.batch_transform is not None else self._to_device(batch, device)
            self.call_hook('before_train_iter')
            if self.hip_graph:
                if self.graphed is None:
                    from .graph import GraphedTrainStep
                    inner = self.model.module if hasattr(self.model, 'finish') else self.model
                    static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
                    self.graphed = GraphedTrainStep(inner, self.optimizer, static, amp_dtype=self.amp_dtype,
                                                    ddp=self.model if hasattr(self.model, 'finish') else None, warmup=3)
                self.outputs = self.graphed(batch)
                self.call_hook('after_train_iter')
                self.iter += 1
                continue
            self.optimizer.zero_grad()
            if self.amp_dtype is not None:
                with torch.autocast(device.type, dtype=self.amp_dtype):
                    self.outputs = self.model.train_step(batch, self.optimizer, **kwargs)
            else:
                self.outputs = self.model.train_step(batch, self.optimizer, **kwargs)
            self.call_hook('after_train_iter')
            self.iter += 1
        if self.graphed is not None:
            # what the run did with its graph, then the graph's private memory pool goes back (before evaluation / the next run)
            self.graph_stats = dict(captured=self.graphed.graph is not None, replays=self.graphed.replays, disabled=self.graphed.disabled)
            self.graphed.release()
            self.graphed = None
        self.call_hook('after_run')

    # ---------------------------------------------------------------- checkpoints (mmcv layout)
    def save_checkpoint(self, path, save_optimizer=True):
        meta = dict(self.meta, iter=self.iter + 1, time=time.asctime())
        model = self.model.module if hasattr(self.model, 'module') else self.model
        save_checkpoint(model, path, optimizer=self.optimizer if save_optimizer else None, meta=meta)

    def load_checkpoint(self, path, map_location='cpu', strict=False):
        model = self.model.module if hasattr(self.model, 'module') else self.model
        return load_checkpoint(model, path, map_location, strict, logger=self.logger)

    def _refresh_shadow(self):
        """Rebuild the optimizer's bf16 parameter shadow from the fp32 arena (GradArena.refresh_shadow's invariant)."""
        arena = getattr(self.optimizer, 'arena', None)
        if arena is not None and hasattr(arena, 'refresh_shadow'):
            arena.refresh_shadow(copy=True)

    def resume(self, path, map_location='cpu'):
        ckpt = self.load_checkpoint(path, map_location)
        self.iter = ckpt.get('meta', {}).get('iter', 0)
        if 'optimizer' in ckpt:
            self.optimizer.load_state_dict(ckpt['optimizer'])
        self._refresh_shadow()
        self.logger(f'resumed from {path} at iter {self.iter}')
