"""Optimizer side of the training step: flat fp32 arenas + one fused AdamW launch.

Replaces mmcv's ``DefaultOptimizerConstructor`` + ``OptimizerHook(grad_clip)`` + ``torch.optim.AdamW``
as configured by configs/depthformer/depthformer_v.py:128-148 (reference).  MI355X-first design: all
parameters live in ONE contiguous fp32 buffer and all gradients in another (autograd accumulates straight into
views of it), so
  * the L2 gradient norm is one streaming reduction (ge_sumsq),
  * clip + AdamW is one launch over the arena (ge_adamw_step) instead of ~480 tensors x ~10 ATen kernels,
  * data-parallel all-reduce operates on large contiguous slices with no flatten/unflatten copies
    (gedepth_amd/mmrt/ddp.py).
"""
import math

import torch

from .. import hip


class GradArena:
    """Flat parameter / gradient storage; ``param.data`` and ``param.grad`` become views into it."""

    def __init__(self, params, align=64, adopt=None):
        """``adopt``: let autograd hand over the gradient tensors it produces (``p.grad = None`` at ``zero_grad``; the engine then
        *assigns* instead of launching one fp32 add per parameter into a zeroed slice — 297 launches per step for Swin-T) and
        bring them into the arena with a few multi-tensor copies (``collect``): per bucket as soon as it is complete under
        FlatDDP, otherwise at the optimizer step.  Default: on for HIP tensors."""
        self.params = [p for p in params if p.requires_grad]
        assert self.params, 'no trainable parameters'
        dev, dt = self.params[0].device, torch.float32
        assert all(p.dtype == dt and p.device == dev for p in self.params), 'fp32 master parameters on one device'
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + align - 1) // align * align
        self.numel = n
        self.flat_param = torch.zeros(n, device=dev, dtype=dt)
        self.flat_grad = torch.zeros(n, device=dev, dtype=dt)
        self.views = []
        for p, off in zip(self.params, self.offsets):
            view = self._view(self.flat_param, off, p)
            view.copy_(p.data)
            p.data = view
            p.grad = self._view(self.flat_grad, off, p)
            self.views.append(p.grad)
        self.adopt = (dev.type == 'cuda') if adopt is None else bool(adopt)
        self._tag_views()

    def _tag_views(self):
        """``p._ge_grad_view``: the parameter's gradient slice, for producers that can write their result straight into the arena
        (``grad_target``) instead of handing autograd a tensor that ``collect`` then copies in."""
        for p, v in zip(self.params, self.views):
            p._ge_grad_view = v if self.adopt else None

    @staticmethod
    def _view(flat, off, p):
        """A view of the arena slice shaped (and, for channels-last conv weights, strided) like ``p``: the optimizer is
        element-wise, so the storage order inside a slice is free and convolution weights can stay in MIOpen's NHWC order."""
        chunk = flat[off:off + p.numel()]
        if p.dim() == 4 and p.shape[1] > 1 and not p.is_contiguous() and p.is_contiguous(memory_format=torch.channels_last):
            O, I, H, W = p.shape
            return chunk.view(O, H, W, I).permute(0, 3, 1, 2)
        return chunk.view_as(p)

    def zero_grad(self):
        if self.adopt:
            for p in self.params:
                p.grad = None
                p._ge_grad_claimed = False
        else:
            self.flat_grad.zero_()

    def collect(self, indices=None):
        """Make the arena hold the current gradients of the parameters ``indices`` (all by default) and point ``p.grad`` back at
        its slice: tensors autograd assigned are copied in (multi-tensor copy), slices of parameters without a gradient are
        zeroed, slices autograd accumulated into directly are left alone.  Idempotent."""
        src, dst, empty = [], [], []
        for i in (range(len(self.params)) if indices is None else indices):
            p, v = self.params[i], self.views[i]
            g = p.grad
            if g is None:
                empty.append(v)
            elif g is not v and g.data_ptr() != v.data_ptr():
                src.append(g.detach())
                dst.append(v)
            p.grad = v
            p._ge_grad_claimed = False
        with torch.no_grad():
            if empty:
                torch._foreach_zero_(empty)
            if src:
                torch._foreach_copy_(dst, src)

    # ---- bf16 shadow of the parameters (what the autocast forward reads) ----
    def enable_shadow(self):
        """Allocate a bf16 copy of the parameter arena and hang a view of it on every parameter (``p._ge_lp``).  The fused
        optimizer keeps it current; consumers go through ``lowp(p, dtype)``, which checks the tensor version so that any
        other write to the parameter (checkpoint load, broadcast, init) falls back to a cast until the next refresh."""
        if getattr(self, 'flat_shadow', None) is None:
            self.flat_shadow = torch.zeros(self.numel, device=self.flat_param.device, dtype=torch.bfloat16)
            for p, off in zip(self.params, self.offsets):
                p._ge_lp = self._view(self.flat_shadow, off, p)
                p._ge_lp_version = -1
        return self.flat_shadow

    def refresh_shadow(self, copy=True):
        """Mark the shadow current (after the optimizer kernel wrote it) or rebuild it from the fp32 arena (``copy``).

        INVARIANT: ``lowp`` trusts the shadow while ``p._version`` is unchanged.  Writes that do not bump that counter —
        ``p.data.copy_ / mul_``, writes through ``arena.flat_param`` (EMA, a manual broadcast, initialisation after
        ``build_optimizer``), raw-pointer kernels — leave the shadow stale until the next optimizer step: call
        ``refresh_shadow(copy=True)`` after any such out-of-band parameter write.  The runner does so at ``before_run`` and after a
        resume, ``FlatDDP`` after its parameter broadcast."""
        if getattr(self, 'flat_shadow', None) is None:
            return
        if copy:
            self.flat_shadow.copy_(self.flat_param)
        for p in self.params:
            p._ge_lp_version = p._version

    def reattach(self):
        """Copy externally-assigned ``.grad`` tensors into the arena and re-point them (tests / foreign code)."""
        self.collect()

    def slices(self):
        return [(off, p.numel()) for p, off in zip(self.params, self.offsets)]

    def on_permute(self, callback):
        """Register ``callback(remap)``, called after ``permute``: ``remap(buf)`` returns the re-ordered copy of any flat buffer
        laid out like this arena before the permutation (optimizer moments, weight-decay mask)."""
        self._permute_callbacks = getattr(self, '_permute_callbacks', []) + [callback]

    @torch.no_grad()
    def permute(self, order):
        """Re-order the slices of the arena: ``order[i]`` = current index of the parameter that comes i-th afterwards.  Parameter,
        gradient and shadow values move with their slices; ``p.data`` / ``p.grad`` / ``p._ge_lp`` are re-pointed.  Used once by
        FlatDDP after the first step, when the order in which backward delivers the gradients is known: buckets are contiguous
        slices, so the arena must be in ARRIVAL order for the first bucket to be complete early (gedepth_amd/mmrt/ddp.py)."""
        n = len(self.params)
        assert sorted(order) == list(range(n)), 'permute() needs a permutation of the parameter indices'
        if list(order) == list(range(n)):
            return
        align_sizes = [(self.offsets[i + 1] if i + 1 < n else self.numel) - self.offsets[i] for i in range(n)]
        old_off = list(self.offsets)
        new_off, pos = [0] * n, 0
        for j, i in enumerate(order):
            new_off[j] = pos
            pos += align_sizes[i]
        assert pos == self.numel
        # one gather index instead of ~500 slice copies: src[k] = old position of the element that lands at k
        src = torch.empty(self.numel, dtype=torch.int64)
        for j, i in enumerate(order):
            src[new_off[j]:new_off[j] + align_sizes[i]] = torch.arange(old_off[i], old_off[i] + align_sizes[i])
        src = src.to(self.flat_param.device)

        def remap(buf):
            assert buf.numel() == self.numel
            return buf.index_select(0, src)
        has_grad = [p.grad is not None for p in self.params]
        self.collect()                                   # every live gradient into its (old) slice before the slices move
        self.flat_param = remap(self.flat_param)
        self.flat_grad = remap(self.flat_grad)
        shadow = getattr(self, 'flat_shadow', None)
        if shadow is not None:
            self.flat_shadow = remap(shadow)
        self.params = [self.params[i] for i in order]
        has_grad = [has_grad[i] for i in order]
        self.offsets = new_off
        self.views = []
        for p, off, hg in zip(self.params, self.offsets, has_grad):
            current = getattr(p, '_ge_lp_version', None) == p._version
            p.data = self._view(self.flat_param, off, p)
            view = self._view(self.flat_grad, off, p)
            self.views.append(view)
            p.grad = view if (hg or not self.adopt) else None
            if shadow is not None:
                p._ge_lp = self._view(self.flat_shadow, off, p)
                p._ge_lp_version = p._version if current else -1
        self._tag_views()
        for cb in getattr(self, '_permute_callbacks', []):
            cb(remap)


def _claim(p):
    """At most ONE producer per parameter and step may write into the arena slice.  ``p.grad is None`` alone does not say "first gradient":
    autograd runs a leaf's AccumulateGrad only after ALL of its uses have produced their gradient, so during the second producer's backward
    of a weight used twice in one forward (tied weights, one module applied to several inputs) ``p.grad`` is still None — both would get the
    same storage, the second would overwrite the first and the engine would then sum two aliases of one buffer.  The claim is dropped by
    ``GradArena.zero_grad`` / ``collect``; a later producer gets None and hands autograd a tensor of its own, which the engine sums with the
    alias as usual."""
    if getattr(p, '_ge_grad_claimed', False):
        return False
    p._ge_grad_claimed = True
    return True


def grad_target(p, dtype=torch.float32):
    """Where a backward may WRITE the gradient of parameter ``p``: a fresh alias of its arena slice, or None.  Only for the first
    producer of a step (``p.grad is None`` after ``zero_grad`` in adopt mode and the slice not yet handed out, see ``_claim``: a second use of
    the same parameter must accumulate, which autograd does on tensors of its own) and only when the slice has the dtype the producer writes.
    Autograd then adopts the alias as ``p.grad`` — same storage as the arena — and ``GradArena.collect`` has nothing to copy (Swin-L: 1.1 GB of
    gradients per step went through a multi-tensor copy, 0.5 ms at 2 images per GPU)."""
    v = getattr(p, '_ge_grad_view', None)
    if v is None or p.grad is not None or v.dtype != dtype or not v.is_contiguous() or not _claim(p):
        return None
    return v.detach()


def grad_target_ohwi(p):
    """``grad_target`` for a channels-last convolution weight: its slice as the contiguous (O, H, W, I) tensor the weight-gradient kernels
    accumulate into; ``.permute(0, 3, 1, 2)`` of it is the (O, I, H, W) gradient with the parameter's own strides."""
    v = getattr(p, '_ge_grad_view', None)
    if v is None or p.grad is not None or v.dtype != torch.float32 or v.dim() != 4:
        return None
    t = v.detach().permute(0, 2, 3, 1)
    if not t.is_contiguous() or not _claim(p):
        return None
    return t


def grad_into_arena(p, src, dtype=None):
    """A finished gradient ``src`` (any dtype / layout) -> the parameter's arena slice in one copy (the widening cast that was needed anyway),
    returned as the alias autograd adopts; falls back to ``src.to(dtype)`` (no slice, a gradient already present, or the slice already
    handed to another producer this step)."""
    v = getattr(p, '_ge_grad_view', None) if p is not None else None
    if v is None or p.grad is not None or v.shape != src.shape or not _claim(p):
        return src.to(dtype or (p.dtype if p is not None else src.dtype))
    t = v.detach()
    t.copy_(src)
    return t


def lowp(p, dtype):
    """``p.to(dtype)`` served from the optimizer's bf16 shadow arena when it is current (no kernel), else a cast."""
    s = getattr(p, '_ge_lp', None)
    if s is not None and s.dtype == dtype and p._ge_lp_version == p._version:
        return s
    return p.to(dtype)


class FusedAdamW(torch.optim.Optimizer):
    """AdamW (torch.optim.AdamW arithmetic) with clip_grad_norm_ folded in, one HIP launch per step.

    All groups must share lr / betas / eps; ``weight_decay`` may be the base value or 0 per group
    (that is what ``paramwise_cfg.custom_keys`` with ``decay_mult=0`` produces)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=0.0, bf16_shadow=True):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        wds = sorted({g['weight_decay'] for g in self.param_groups})
        nz = [w for w in wds if w != 0]
        if len(nz) > 1:
            raise NotImplementedError('FusedAdamW supports one non-zero weight decay (decay_mult in {0, 1})')
        self.base_wd = nz[0] if nz else 0.0
        all_params = [p for g in self.param_groups for p in g['params'] if p.requires_grad]
        self.arena = GradArena(all_params)
        dev = self.arena.flat_param.device
        if dev.type != 'cuda':
            raise RuntimeError('FusedAdamW runs on MI355X only (no CPU fallback)')
        self.exp_avg = torch.zeros_like(self.arena.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.arena.flat_param)
        self.wd_mask = torch.zeros(self.arena.numel, device=dev, dtype=torch.uint8)
        decay = {id(p) for g in self.param_groups if g['weight_decay'] != 0 for p in g['params']}
        for p, (off, n) in zip(self.arena.params, self.arena.slices()):
            if id(p) in decay:
                self.wd_mask[off:off + n] = 1
        self.max_grad_norm = float(max_grad_norm)
        self.step_count = 0
        self.hyper = torch.zeros(10, device=dev, dtype=torch.float32)
        # ring of pinned staging buffers: the H2D copy of the step scalars never blocks the host
        self._ring = [(torch.zeros(10, dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(8)]
        self.gnorm_sq = torch.zeros(1, device=dev, dtype=torch.float64)
        if bf16_shadow:                            # 2 B / parameter: the autocast forward reads weights from here, no per-tensor casts
            self.arena.enable_shadow()
            self.arena.refresh_shadow(copy=True)
        self.arena.on_permute(self._arena_permuted)

    def _arena_permuted(self, remap):
        """The arena changed its slice order (FlatDDP: arrival order): the moments and the decay mask follow their parameters."""
        self.exp_avg, self.exp_avg_sq, self.wd_mask = remap(self.exp_avg), remap(self.exp_avg_sq), remap(self.wd_mask)

    def sync_grads_from_params(self):
        self.arena.reattach()

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    @property
    def last_grad_norm(self):
        """Device scalar: L2 norm of the (un-clipped) gradient of the latest step."""
        return self.gnorm_sq.sqrt()

    @torch.no_grad()
    def prepare(self):
        """Host half of a step: advance the step counter and put this step's scalars (lr, betas, eps, weight decay, bias corrections,
        clip norm) into the device buffer the kernels read.  Kept apart from ``launch`` so that a captured training step
        (gedepth_amd/mmrt/graph.py) holds only the kernels and the scalars are refreshed before every replay."""
        g0 = self.param_groups[0]
        assert all(g['lr'] == g0['lr'] for g in self.param_groups), 'per-group learning rates are not supported'
        self.step_count += 1
        b1, b2 = g0['betas']
        t = self.step_count
        host, ev = self._ring[t % len(self._ring)]
        ev.synchronize()                       # slot was used 8 steps ago: already complete in practice
        host.copy_(torch.tensor([g0['lr'], b1, b2, g0['eps'], self.base_wd, 1 - b1 ** t, 1 - b2 ** t,
                                 self.max_grad_norm, 1 - b1, 1 - b2], dtype=torch.float32))      # 1 - beta in double, like torch
        self.hyper.copy_(host, non_blocking=True)
        ev.record()

    @torch.no_grad()
    def launch(self):
        """Device half of a step: gradient norm + clip + AdamW (+ bf16 shadow) over the arenas.  Capturable."""
        lib, a = hip.lib(), self.arena
        a.collect()                            # gradients autograd handed over (not yet brought in by FlatDDP) -> arena
        self.gnorm_sq.zero_()
        hip.check(lib.ge_sumsq(hip.ptr(a.flat_grad), a.numel, hip.ptr(self.gnorm_sq), hip.stream()), 'ge_sumsq')
        shadow = getattr(a, 'flat_shadow', None)
        if shadow is None:
            hip.check(lib.ge_adamw_step(hip.ptr(a.flat_param), hip.ptr(a.flat_grad), hip.ptr(self.exp_avg), hip.ptr(self.exp_avg_sq),
                                        hip.ptr(self.wd_mask), hip.ptr(self.hyper), hip.ptr(self.gnorm_sq), a.numel,
                                        hip.stream()), 'ge_adamw_step')
        else:
            hip.check(lib.ge_adamw_step_shadow(hip.ptr(a.flat_param), hip.ptr(a.flat_grad), hip.ptr(self.exp_avg),
                                               hip.ptr(self.exp_avg_sq), hip.ptr(self.wd_mask), hip.ptr(self.hyper),
                                               hip.ptr(self.gnorm_sq), a.numel, hip.ptr(shadow), hip.stream()), 'ge_adamw_step_shadow')
            a.refresh_shadow(copy=False)

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        self.prepare()
        self.launch()

    def state_dict(self):
        """``torch.optim.AdamW`` layout (what mmcv's CheckpointHook stores and the reference's checkpoints hold):
        ``state[i] = {step, exp_avg, exp_avg_sq}`` per parameter — views sliced out of the arenas — and ``param_groups``
        with parameter ids, ids running over the groups in order."""
        index = {id(p): i for i, p in enumerate(p for g in self.param_groups for p in g['params'])}
        state = {}
        for p, (off, n) in zip(self.arena.params, self.arena.slices()):
            state[index[id(p)]] = dict(step=torch.tensor(float(self.step_count)),
                                       exp_avg=GradArena._view(self.exp_avg, off, p).clone(),
                                       exp_avg_sq=GradArena._view(self.exp_avg_sq, off, p).clone())
        groups = []
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != 'params'}
            d['params'] = [index[id(p)] for p in g['params']]
            groups.append(d)
        return dict(state=state, param_groups=groups)

    def load_state_dict(self, state_dict):
        """Accepts the torch layout above (ours, mmcv's, the reference's) and the flat-arena layout of round 1
        (``{step, exp_avg, exp_avg_sq, param_groups}``).  Shapes are validated per parameter before anything is copied."""
        if 'state' not in state_dict:                                   # round-1 flat layout
            for k in ('exp_avg', 'exp_avg_sq'):
                if tuple(state_dict[k].shape) != (self.arena.numel,):
                    raise ValueError(f'optimizer {k}: arena of {tuple(state_dict[k].shape)} elements, this model needs {self.arena.numel}')
            self.step_count = int(state_dict['step'])
            # round-1 arenas stored EVERY slice in the parameter's logical (NCHW) order; today a channels-last conv weight keeps
            # its slice in NHWC order (GradArena._view).  Copy slice by slice through the view, so that a moment lands on the
            # element it belongs to whatever the layout of this run (a 1:1 arena copy would permute the moments of every conv
            # weight with more than one input channel, silently).
            for k, buf in (('exp_avg', self.exp_avg), ('exp_avg_sq', self.exp_avg_sq)):
                src = state_dict[k].to(buf.device, torch.float32)
                buf.zero_()
                for p, (off, n) in zip(self.arena.params, self.arena.slices()):
                    GradArena._view(buf, off, p).copy_(src[off:off + n].view(p.shape))
            for g, s in zip(self.param_groups, state_dict['param_groups']):
                g.update({k: v for k, v in s.items() if k != 'params'})
            return
        saved_groups = state_dict['param_groups']
        if len(saved_groups) != len(self.param_groups):
            raise ValueError(f'optimizer state has {len(saved_groups)} parameter groups, this optimizer {len(self.param_groups)}')
        own_ids, id_of = [], {}
        for g, sg in zip(self.param_groups, saved_groups):
            if len(g['params']) != len(sg['params']):
                raise ValueError('optimizer state: a parameter group has a different number of parameters')
            for p, pid in zip(g['params'], sg['params']):
                id_of[id(p)] = pid
        st = state_dict['state']
        st = {int(k): v for k, v in st.items()}
        plan, steps = [], set()
        for p, (off, n) in zip(self.arena.params, self.arena.slices()):
            rec = st.get(id_of[id(p)])
            if rec is None:                                              # parameter that never received a gradient
                plan.append((off, p, None))
                continue
            for k in ('exp_avg', 'exp_avg_sq'):
                if tuple(rec[k].shape) != tuple(p.shape):
                    raise ValueError(f'optimizer state {k} of parameter {id_of[id(p)]}: shape {tuple(rec[k].shape)} != {tuple(p.shape)}')
            steps.add(int(float(rec['step'])))
            plan.append((off, p, rec))
        if len(steps) > 1:
            raise NotImplementedError(f'FusedAdamW keeps one step counter; the checkpoint has {sorted(steps)}')
        for off, p, rec in plan:
            for buf, k in ((self.exp_avg, 'exp_avg'), (self.exp_avg_sq, 'exp_avg_sq')):
                view = GradArena._view(buf, off, p)                         # same element order as the parameter's arena slice
                if rec is None:
                    view.zero_()
                else:
                    view.copy_(rec[k].to(buf.device, torch.float32))
        self.step_count = steps.pop() if steps else 0
        for g, s in zip(self.param_groups, saved_groups):
            g.update({k: v for k, v in s.items() if k != 'params'})


def paramwise_groups(model, base_lr, base_wd, paramwise_cfg=None):
    """mmcv ``DefaultOptimizerConstructor`` for the keys the reference uses: ``custom_keys`` matched as
    substrings of the full parameter name, longest key first; ``lr_mult`` / ``decay_mult`` per key."""
    custom = dict((paramwise_cfg or {}).get('custom_keys', {}))
    keys = sorted(sorted(custom.keys()), key=len, reverse=True)
    groups = []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        g = dict(params=[p], lr=base_lr, weight_decay=base_wd, name=name)
        for k in keys:
            if k in name:
                g['lr'] = base_lr * custom[k].get('lr_mult', 1.)
                g['weight_decay'] = base_wd * custom[k].get('decay_mult', 1.)
                break
        groups.append(g)
    return groups


def build_optimizer(model, cfg, grad_clip=None):
    """``cfg`` = the config's ``optimizer`` dict (type AdamW); ``grad_clip`` = optimizer_config.grad_clip."""
    cfg = dict(cfg)
    typ = cfg.pop('type')
    if typ != 'AdamW':
        raise NotImplementedError(f'optimizer {typ}: the GEDepth configs train with AdamW')
    pw = cfg.pop('paramwise_cfg', None)
    lr, wd = cfg.pop('lr'), cfg.pop('weight_decay', 0.01)
    if hasattr(model, 'module'):
        model = model.module
    groups = paramwise_groups(model, lr, wd, pw)
    max_norm = 0.0
    if grad_clip:
        assert grad_clip.get('norm_type', 2) == 2
        max_norm = grad_clip['max_norm']
    return FusedAdamW(groups, lr=lr, weight_decay=wd, max_grad_norm=max_norm, **cfg)


class CosineAnnealingLr:
    """mmcv CosineAnnealingLrUpdaterHook (by_epoch=False) with linear warm-up
    (configs/depthformer/depthformer_v.py:141-147)."""

    def __init__(self, base_lr, max_iters, min_lr=None, min_lr_ratio=None, warmup=None, warmup_iters=0,
                 warmup_ratio=0.1, by_epoch=False, policy=None):
        assert (min_lr is None) ^ (min_lr_ratio is None)
        assert warmup in (None, 'linear', 'constant', 'exp')
        self.base_lr, self.max_iters = base_lr, max_iters
        self.target = min_lr if min_lr is not None else base_lr * min_lr_ratio
        self.warmup, self.warmup_iters, self.warmup_ratio = warmup, warmup_iters, warmup_ratio

    def lr_at(self, it):
        cos_out = math.cos(math.pi * (it / self.max_iters)) + 1
        lr = self.target + 0.5 * (self.base_lr - self.target) * cos_out
        if self.warmup is not None and it < self.warmup_iters:
            if self.warmup == 'constant':
                lr = lr * self.warmup_ratio
            elif self.warmup == 'linear':
                lr = lr * (1 - (1 - it / self.warmup_iters) * (1 - self.warmup_ratio))
            else:
                lr = lr * self.warmup_ratio ** (1 - it / self.warmup_iters)
        return lr

    def apply(self, optimizer, it):
        lr = self.lr_at(it)
        for g in optimizer.param_groups:
            g['lr'] = lr
        return lr
