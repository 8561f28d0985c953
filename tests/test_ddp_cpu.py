"""Multi-process data-parallel path on CPU (gloo, world_size 2): flat-arena bucketed all-reduce, parameter
broadcast, fused scalar all-reduce of the log vars, and the iteration-based runner.  SURVEY.md §8(e)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gedepth_amd.depth.models.depther.base import BaseDepther, DeferredLogVars  # noqa: E402
from gedepth_amd.mmrt.ddp import FlatDDP  # noqa: E402
from gedepth_amd.mmrt.optim import CosineAnnealingLr, GradArena, paramwise_groups  # noqa: E402


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(8, 32)
        self.norm = nn.LayerNorm(32)
        self.b = nn.Linear(32, 16)
        self.c = nn.Linear(16, 1)

    def forward(self, x):
        return self.c(torch.relu(self.b(self.norm(torch.relu(self.a(x))))))

    def train_step(self, batch, optimizer=None):
        loss = (self(batch['x']) - batch['y']).pow(2).mean()
        l, lv = BaseDepther._parse_losses({'loss_mse': loss, 'aux': loss.detach() * 2})
        return dict(loss=l, log_vars=lv, num_samples=len(batch['x']))


def reg_grad(model):
    """The gradient as one vector in REGISTRATION order, whatever order the arena's slices are in (FlatDDP permutes the arena into
    gradient-arrival order after its first step)."""
    return torch.cat([p.grad.detach().reshape(-1) for p in model.parameters()])


class ArenaSGD:
    """Minimal optimizer over a GradArena for the CPU tests (the product's FusedAdamW is MI355X-only)."""

    def __init__(self, model, lr=0.1, adopt=None):
        self.arena = GradArena(model.parameters(), adopt=adopt)
        self.defaults = dict(lr=lr)
        self.param_groups = [dict(lr=lr, params=self.arena.params)]

    def zero_grad(self):
        self.arena.zero_grad()

    def step(self):
        self.arena.collect()
        self.arena.flat_param.add_(self.arena.flat_grad, alpha=-self.param_groups[0]['lr'])

    def state_dict(self):
        return {}

    def load_state_dict(self, state):
        pass


def _worker(rank, world, port, tmp, adopt=None):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                      # different init per rank: broadcast must fix it
    model = Toy()
    opt = ArenaSGD(model, adopt=adopt)
    ddp = FlatDDP(model, opt.arena, bucket_mb=0.0005)  # ~130 floats per bucket -> several buckets
    assert len(ddp.buckets) >= 3
    g = torch.Generator().manual_seed(0)
    X, Y = torch.randn(8, 8, generator=g), torch.randn(8, 1, generator=g)
    batch = dict(x=X[rank * 4:(rank + 1) * 4], y=Y[rank * 4:(rank + 1) * 4])
    opt.zero_grad()
    out = ddp.train_step(batch)
    out['loss'].backward()
    ddp.finish()
    res = dict(params={k: v.clone() for k, v in model.state_dict().items()},
               grad=reg_grad(model).clone(), log=dict(out['log_vars']), local_loss=out['loss'].item())
    # second iteration exercises the bucket state reset
    opt.step()
    opt.zero_grad()
    out = ddp.train_step(batch)
    out['loss'].backward()
    ddp.finish()
    res['grad2'] = reg_grad(model).clone()
    assert ddp._order_learned and ddp.describe()['launch_order'] == list(range(len(ddp.buckets)))
    torch.save(res, os.path.join(tmp, f'r{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('adopt', [False, True])
def test_flat_ddp_gloo_world2(tmp_path, adopt):
    """``adopt``: gradients handed over by autograd and copied into the arena per bucket (the MI355X default) vs accumulated
    straight into the arena slices — the exchanged result must be the same."""
    port = free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), adopt), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f'r{i}.pt', weights_only=False) for i in range(2))
    # broadcast: both ranks hold rank 0's initial parameters
    for k in r0['params']:
        assert torch.equal(r0['params'][k], r1['params'][k]), k
    # gradients: identical across ranks and equal to the full-batch gradient of a single process
    assert torch.equal(r0['grad'], r1['grad'])
    assert torch.equal(r0['grad2'], r1['grad2'])
    torch.manual_seed(100)
    ref = Toy()
    ref.load_state_dict(r0['params'])
    g = torch.Generator().manual_seed(0)
    X, Y = torch.randn(8, 8, generator=g), torch.randn(8, 1, generator=g)
    (ref(X) - Y).pow(2).mean().backward()
    assert torch.allclose(reg_grad(ref), r0['grad'], rtol=1e-5, atol=1e-7)
    # log vars: mean over ranks, one fused all-reduce
    assert abs(r0['log']['loss'] - 0.5 * (r0['local_loss'] + r1['local_loss'])) < 1e-6
    assert abs(r0['log']['aux'] - 2 * r0['log']['loss_mse']) < 1e-6
    assert r0['log'] == r1['log']


def _worker_bf16(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(7)
    model = Toy()
    opt = ArenaSGD(model)
    ddp = FlatDDP(model, opt.arena, bucket_mb=0.0005, grad_dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(0)
    X, Y = torch.randn(8, 8, generator=g), torch.randn(8, 1, generator=g)
    batch = dict(x=X[rank * 4:(rank + 1) * 4], y=Y[rank * 4:(rank + 1) * 4])
    opt.zero_grad()
    out = ddp.train_step(batch)
    out['loss'].backward()
    opt.arena.collect()
    local = reg_grad(model).clone()
    ddp.finish()
    torch.save(dict(local=local, reduced=reg_grad(model).clone()), os.path.join(tmp, f'b{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_ddp_bf16_gradient_exchange(tmp_path):
    """grad_dtype=bf16 (half the xGMI bytes): every rank ends with mean over ranks of the bf16-rounded
    local gradients, widened back into the fp32 arena — identical on all ranks, within bf16 rounding of the fp32 mean."""
    port = free_port()
    mp.spawn(_worker_bf16, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f'b{i}.pt', weights_only=False) for i in range(2))
    assert torch.equal(r0['reduced'], r1['reduced'])
    exact = 0.5 * (r0['local'] + r1['local'])
    expect = ((r0['local'].bfloat16() + r1['local'].bfloat16()).float()) * 0.5         # the wire format's own arithmetic
    assert torch.allclose(r0['reduced'], expect, rtol=0, atol=1e-6 * expect.abs().max().item() + 1e-12) or \
        (r0['reduced'] - exact).abs().max() <= 2 ** -7 * exact.abs().max()
    assert (r0['reduced'] - exact).abs().max() <= 2 ** -7 * exact.abs().max()
    assert not torch.equal(r0['reduced'], exact)                      # really went through bf16


def test_deferred_log_vars_single_process():
    lv = DeferredLogVars(['a', 'b'], torch.tensor([1.5, 2.5]))
    assert lv.tensor() is not None
    assert lv['a'] == 1.5 and dict(lv.items()) == {'a': 1.5, 'b': 2.5} and lv.tensor() is None
    loss, lv = BaseDepther._parse_losses({'loss_x': torch.tensor([1.0, 3.0]), 'acc': torch.tensor(0.5),
                                          'loss_list': [torch.tensor(1.0), torch.tensor(2.0)]})
    assert loss.item() == 5.0 and lv['loss'] == 5.0 and lv['loss_x'] == 2.0 and lv['acc'] == 0.5
    with pytest.raises(TypeError):
        BaseDepther._parse_losses({'loss': 1.0})


def test_cosine_lr_with_linear_warmup():
    """mmcv CosineAnnealing by iteration + linear warm-up (configs/depthformer/depthformer_v.py:141-147)."""
    s = CosineAnnealingLr(1e-4, 76800, min_lr_ratio=1e-8, warmup='linear', warmup_iters=25600, warmup_ratio=1e-3)
    assert abs(s.lr_at(0) - 1e-4 * 1e-3) < 1e-12
    import math
    reg = lambda it: 1e-12 + 0.5 * (1e-4 - 1e-12) * (math.cos(math.pi * it / 76800) + 1)
    it = 12800
    assert abs(s.lr_at(it) - reg(it) * (1 - (1 - it / 25600) * (1 - 1e-3))) < 1e-15
    assert abs(s.lr_at(25600) - reg(25600)) < 1e-15
    assert abs(s.lr_at(76800) - 1e-12) < 1e-15


def test_paramwise_decay_mult():
    from gedepth_amd.depth.models import build_depther
    from gedepth_amd.mmrt.config import Config
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', 'depthformer_swint_v.py'))
    cfg.model.pretrained = None
    cfg.model.backbone.depths = [1, 1, 1, 1]
    m = build_depther(cfg.model)
    groups = {g['name']: g for g in paramwise_groups(m, 1e-4, 0.01, cfg.optimizer.paramwise_cfg)}
    assert groups['backbone.stages.0.blocks.0.attn.w_msa.relative_position_bias_table']['weight_decay'] == 0
    assert groups['backbone.stages.0.blocks.0.norm1.weight']['weight_decay'] == 0
    assert groups['backbone.norm2.bias']['weight_decay'] == 0
    assert groups['backbone.stages.0.downsample.norm.weight']['weight_decay'] == 0
    assert groups['backbone.stages.0.blocks.0.attn.w_msa.qkv.weight']['weight_decay'] == 0.01
    assert groups['neck.lateral_convs.0.bn.weight']['weight_decay'] == 0.01     # 'norm' is not a substring of 'bn'
    assert groups['decode_head.conv_depth.weight']['weight_decay'] == 0.01


def test_runner_hooks_checkpoint_resume(tmp_path):
    from gedepth_amd.mmrt.runner import IterBasedRunner
    torch.manual_seed(0)
    model = Toy()
    opt = ArenaSGD(model, lr=0.05)
    logs = []
    runner = IterBasedRunner(model, opt, work_dir=str(tmp_path), logger=logs.append, max_iters=12)
    runner.register_training_hooks(dict(policy='CosineAnnealing', warmup='linear', warmup_iters=4, warmup_ratio=0.1,
                                        min_lr_ratio=1e-3, by_epoch=False),
                                   dict(grad_clip=None), dict(by_epoch=False, interval=4, max_keep_ckpts=2),
                                   dict(interval=3, hooks=[dict(type='TextLoggerHook', by_epoch=False),
                                                           dict(type='TensorboardLoggerHook')]))
    g = torch.Generator().manual_seed(0)
    data = [dict(x=torch.randn(4, 8, generator=g), y=torch.randn(4, 1, generator=g)) for _ in range(3)]
    runner.run([data])
    assert runner.iter == 12 and len(logs) == 4
    assert sorted(f for f in os.listdir(tmp_path) if f.endswith('.pth')) == ['iter_12.pth', 'iter_8.pth']
    first, last = float(logs[0].split('loss: ')[1].split(',')[0]), float(logs[-1].split('loss: ')[1].split(',')[0])
    assert last < first
    ckpt = torch.load(tmp_path / 'iter_12.pth', weights_only=False)
    assert set(ckpt) >= {'meta', 'state_dict'} and ckpt['meta']['iter'] == 12
    model2 = Toy()
    r2 = IterBasedRunner(model2, ArenaSGD(model2), work_dir=str(tmp_path), logger=logs.append, max_iters=12)
    r2.resume(str(tmp_path / 'iter_12.pth'))
    assert r2.iter == 12
    for k, v in model.state_dict().items():
        assert torch.equal(v, model2.state_dict()[k])


def test_runner_reseeds_sampler_each_epoch_and_on_resume():
    """mmcv IterLoader semantics: ``sampler.set_epoch`` on every restart of the loader (a DistributedSampler otherwise
    replays one permutation for ever), epoch counter derived from the iteration on resume."""
    from torch.utils.data import DataLoader, DistributedSampler
    from gedepth_amd.mmrt.runner import IterBasedRunner

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 6

        def __getitem__(self, i):
            return dict(x=torch.full((8,), float(i)), y=torch.zeros(1), idx=torch.tensor(i))

    class Spy(Toy):
        seen = []

        def train_step(self, batch, optimizer, **kw):
            Spy.seen.append(batch['idx'].tolist())
            return super().train_step(batch, optimizer, **kw)
    torch.manual_seed(0)
    model = Spy()
    sampler = DistributedSampler(DS(), num_replicas=1, rank=0, shuffle=True, seed=3)
    calls = []
    orig = sampler.set_epoch
    sampler.set_epoch = lambda e: (calls.append(e), orig(e))[1]
    loader = DataLoader(DS(), batch_size=3, sampler=sampler)
    runner = IterBasedRunner(model, ArenaSGD(model), logger=lambda m: None, max_iters=6)
    runner.register_training_hooks(dict(policy='CosineAnnealing', min_lr_ratio=1e-3, by_epoch=False), dict(grad_clip=None))
    Spy.seen = []
    runner.run([loader])
    assert calls == [0, 1, 2] and runner.epoch == 2
    epochs = [sum(Spy.seen[2 * e:2 * e + 2], []) for e in range(3)]
    assert all(sorted(e) == list(range(6)) for e in epochs) and len({tuple(e) for e in epochs}) > 1, epochs
    calls.clear()
    r2 = IterBasedRunner(model, ArenaSGD(model), logger=lambda m: None, max_iters=6)
    r2.register_training_hooks(dict(policy='CosineAnnealing', min_lr_ratio=1e-3, by_epoch=False), dict(grad_clip=None))
    r2.iter = 4                                               # as after resume(): 4 iterations = 2 epochs of 2 batches
    r2.run([loader])
    assert calls[0] == 2


def test_loader_workers_get_distinct_reproducible_seeds():
    """depth/datasets/builder.py:152-157: seed = num_workers * rank + worker_id + seed for numpy / random / torch."""
    import numpy as np
    from gedepth_amd.depth.datasets.loader import build_dataloader

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 8

        def __getitem__(self, i):
            info = torch.utils.data.get_worker_info()
            return dict(worker=torch.tensor(info.id), draw=torch.tensor(np.random.randint(0, 2 ** 31 - 1)))

    def draws(seed):
        out = {}
        for b in build_dataloader(DS(), 1, workers_per_gpu=2, dist=False, shuffle=False, seed=seed, pin_memory=False):
            out.setdefault(int(b['worker']), []).append(int(b['draw']))
        return out
    a, b = draws(5), draws(5)
    assert a == b and a[0] != a[1], (a, b)                     # reproducible, and the two workers differ
    exp = np.random.RandomState(2 * 0 + 0 + 5).randint(0, 2 ** 31 - 1)
    assert a[0][0] == exp
    c = draws(None)
    assert c[0] != c[1]


def test_load_checkpoint_reports_and_refuses_mismatches(tmp_path):
    from gedepth_amd.mmrt.checkpoint import load_checkpoint, save_checkpoint
    m = Toy()
    save_checkpoint(m, str(tmp_path / 'a.pth'))
    logs = []
    load_checkpoint(Toy(), str(tmp_path / 'a.pth'), logger=logs.append)
    assert not logs
    sd = m.state_dict()
    torch.save(dict(state_dict={'backbone.' + k: v for k, v in sd.items()}), tmp_path / 'prefixed.pth')
    with pytest.raises(RuntimeError, match='none of the'):
        load_checkpoint(Toy(), str(tmp_path / 'prefixed.pth'))
    first = next(iter(sd))
    part = {k: v for k, v in sd.items() if k != first}
    part['stray.weight'] = torch.zeros(1)
    torch.save(dict(state_dict=part), tmp_path / 'part.pth')
    load_checkpoint(Toy(), str(tmp_path / 'part.pth'), logger=logs.append)
    assert any('missing' in l and first in l for l in logs) and any('unexpected' in l and 'stray.weight' in l for l in logs)


# ---------------------------------------------------------------- distributed evaluation (depth/apis/test.py)
class _IndexDataset(torch.utils.data.Dataset):
    """5 samples (odd on purpose: the DistributedSampler pads rank 1 with a repeated index)."""

    def __len__(self):
        return 5

    def __getitem__(self, i):
        return dict(img=[torch.full((1, 2, 2), float(i))], img_metas=[dict(index=i)])

    def pre_eval(self, preds, indices):
        return [(float(p.mean()), i) for p, i in zip(preds, indices)], preds


class _Echo(nn.Module):
    def __init__(self):
        super().__init__()
        self.p = nn.Parameter(torch.zeros(1))

    def forward(self, img, img_metas, return_loss=True, **kw):
        assert return_loss is False
        return [t.numpy() for t in img[0]]


def _eval_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from gedepth_amd.depth.apis.test import multi_gpu_test
    from gedepth_amd.depth.datasets import build_dataloader
    loader = build_dataloader(_IndexDataset(), 1, 0, dist=True, shuffle=False, pin_memory=False)
    res = multi_gpu_test(_Echo(), loader, pre_eval=True, device='cpu')
    torch.save(res, os.path.join(tmp, f'eval{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_multi_gpu_test_returns_dataset_order_on_rank0(tmp_path):
    port = free_port()
    mp.spawn(_eval_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / 'eval0.pt', weights_only=False)
    r1 = torch.load(tmp_path / 'eval1.pt', weights_only=False)
    assert r1 is None
    assert [v for v, _ in r0] == [0.0, 1.0, 2.0, 3.0, 4.0] and [i for _, i in r0] == [0, 1, 2, 3, 4]


def test_grad_arena_adopts_autograd_gradients():
    """adopt mode: zero_grad drops the .grad references, backward assigns fresh tensors, collect() brings them into the arena
    (zeros for parameters that got no gradient), a second collect is a no-op, and accumulation over two backward passes
    before the step still sums."""
    torch.manual_seed(0)
    model = Toy()
    extra = torch.nn.Parameter(torch.ones(3))                     # never used in the loss: its slice must read zero
    arena = GradArena(list(model.parameters()) + [extra], adopt=True)
    arena.flat_grad.fill_(7.0)                                    # stale content of a previous step
    arena.zero_grad()
    assert all(p.grad is None for p in arena.params)
    x, y = torch.randn(4, 8), torch.randn(4, 1)
    (model(x) - y).pow(2).mean().backward()
    handed = [p.grad for p in arena.params[:-1]]
    assert all(g is not None and g.data_ptr() != v.data_ptr() for g, v in zip(handed, arena.views))
    (model(x) - y).pow(2).mean().backward()                       # accumulates into the handed-over tensors
    arena.collect()
    ref = Toy()
    ref.load_state_dict(model.state_dict())
    (2 * (ref(x) - y).pow(2).mean()).backward()
    for i, (p, q) in enumerate(zip(arena.params[:-1], ref.parameters())):
        assert p.grad.data_ptr() == arena.views[i].data_ptr()
        assert torch.allclose(p.grad, q.grad, rtol=1e-6, atol=1e-8)
    assert bool((arena.views[-1] == 0).all())
    before = arena.flat_grad.clone()
    arena.collect()
    assert torch.equal(before, arena.flat_grad)


def test_arena_slice_is_handed_out_once_per_parameter_and_step():
    """One weight used TWICE in a forward (tied weights / a module applied to two inputs) under ``GradArena(adopt=True)``: during the second
    producer's backward ``p.grad`` is still None (AccumulateGrad runs after all uses), so "p.grad is None" alone would give both producers the
    same arena slice — the second overwrites the first and the engine sums two aliases of one buffer (measured before the fix: max error 488 on
    gradients of magnitude 472).  ``optim._claim`` hands the slice out once; the second producer gets a tensor of its own.  All three producer
    entry points (grad_target via the token Linear, grad_target_ohwi, grad_into_arena), against plain autograd (what the reference does)."""
    from gedepth_amd.mmrt import bricks
    from gedepth_amd.mmrt.optim import grad_into_arena, grad_target, grad_target_ohwi
    torch.manual_seed(0)
    lin = torch.nn.Linear(16, 24, bias=False)                          # (the bias column sum is a HIP kernel: no CPU path by design)
    arena = GradArena(lin.parameters(), adopt=True)
    x1, x2 = torch.randn(5, 40, 16), torch.randn(3, 7, 16)
    ref_w = lin.weight.detach().clone().requires_grad_(True)
    (torch.nn.functional.linear(x1, ref_w).square().sum() + 3 * torch.nn.functional.linear(x2, ref_w).sin().sum()).backward()
    for step in range(2):                                              # twice: zero_grad / collect must drop the claim again
        arena.zero_grad()
        y1 = bricks._LinearTokens.apply(x1, lin.weight, None, 0)
        y2 = bricks._LinearTokens.apply(x2, lin.weight, None, 0)
        (y1.square().sum() + 3 * y2.sin().sum()).backward()
        arena.collect()
        assert torch.allclose(lin.weight.grad, ref_w.grad, rtol=1e-5, atol=1e-5), (step, (lin.weight.grad - ref_w.grad).abs().max().item())
        assert lin.weight.grad.data_ptr() == arena.views[0].data_ptr() and not lin.weight._ge_grad_claimed
    # the three entry points share the claim
    conv = torch.nn.Conv2d(8, 4, 3).to(memory_format=torch.channels_last)
    a2 = GradArena(conv.parameters(), adopt=True)
    w = conv.weight
    for first in (grad_target_ohwi, lambda p: grad_into_arena(p, torch.ones_like(p)), ):
        a2.zero_grad()
        t = first(w)
        assert t is not None and t.data_ptr() == w._ge_grad_view.data_ptr()
        assert grad_target_ohwi(w) is None and grad_target(w) is None
        again = grad_into_arena(w, torch.full_like(w, 2.0))
        assert again.data_ptr() != w._ge_grad_view.data_ptr() and bool((again == 2).all())
        if first is not grad_target_ohwi:
            assert bool((w._ge_grad_view == 1).all())                  # the first producer's values were not overwritten
    a2.collect()
    assert grad_target_ohwi(w) is None                                 # after collect p.grad is the slice: accumulate through autograd
    a2.zero_grad()
    assert grad_target_ohwi(w) is not None


def test_bf16_shadow_serves_only_current_values():
    """``lowp`` hands out the optimizer's bf16 shadow view while it reflects the parameter (refresh after the write) and a plain
    cast after any other in-place write (checkpoint load, broadcast through the parameter, init) until the next refresh."""
    from gedepth_amd.mmrt.optim import lowp
    torch.manual_seed(0)
    model = Toy()
    arena = GradArena(model.parameters())
    p = arena.params[0]
    assert lowp(p, torch.bfloat16).data_ptr() != p.data_ptr() and not hasattr(p, '_ge_lp')       # no shadow: a cast
    arena.enable_shadow()
    assert lowp(p, torch.bfloat16).data_ptr() != p._ge_lp.data_ptr()                             # allocated but not yet filled
    arena.refresh_shadow(copy=True)
    for q in arena.params:
        s = lowp(q, torch.bfloat16)
        assert s.data_ptr() == q._ge_lp.data_ptr() and s.shape == q.shape and torch.equal(s, q.detach().to(torch.bfloat16))
        assert lowp(q, torch.float32) is q
    with torch.no_grad():
        p.mul_(2.0)
    stale = lowp(p, torch.bfloat16)
    assert stale.data_ptr() != p._ge_lp.data_ptr() and torch.equal(stale, p.detach().to(torch.bfloat16))
    arena.flat_param.mul_(0.5)                                    # a write through the arena does not touch tensor versions ...
    arena.refresh_shadow(copy=True)                               # ... so whoever does it refreshes (FusedAdamW, FlatDDP broadcast)
    assert lowp(p, torch.bfloat16).data_ptr() == p._ge_lp.data_ptr()
    assert torch.equal(p._ge_lp, p.detach().to(torch.bfloat16))


# ------------------------------------------------------------------ world size 4: ragged buckets, late / missing gradients, trace
class Branchy(nn.Module):
    """Parameters whose gradients arrive OUT of registration order (the `late` head is registered first but used last in the
    forward, so backward produces it first; `early` is registered last and produced last), one parameter that gets no gradient
    on odd steps, and sizes that leave a ragged tail bucket."""

    def __init__(self):
        super().__init__()
        self.late = nn.Linear(5, 1)                      # first in the arena -> LAST bucket (buckets are built from the end)
        self.body = nn.Linear(7, 5)
        self.skip = nn.Linear(7, 5)                      # unused on odd steps
        self.early = nn.Linear(3, 7)                     # last in the arena -> FIRST bucket, but its gradient arrives last

    def forward(self, x, use_skip):
        h = self.early(x)
        y = torch.tanh(self.body(h))
        if use_skip:
            y = y + self.skip(h)
        return self.late(y)


# which ranks run the `skip` branch in which step: step 0 leaves rank 0 WITHOUT a gradient for `skip` (rank 0's arrival sequence is
# incomplete -> nobody may adopt an order), step 1 leaves rank 1 without it while rank 0 is complete (every rank must adopt rank 0's
# sequence, rank 1 included, although its own hook history is incomplete: the decision is collective, not per rank)
_SKIP_PLAN = [(False, True, True, True), (True, False, True, True), (False, False, False, False), (True, True, True, True)]


def _worker4(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), GE_DDP_TRACE='1')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(3 + rank)
    model = Branchy()
    arena = GradArena(model.parameters(), align=1, adopt=True)       # align 1: slice sizes 5,1,35,5,35,5,21,7 -> ragged buckets
    ddp = FlatDDP(model, arena, bucket_mb=30 * 4 / 2 ** 20)          # 30 floats per bucket
    sizes = [hi - lo for lo, hi, _ in ddp.buckets]
    # built from the end of the arena: 7 + 21 + 5 = 33, 35, 5 + 35 = 40 and a 6-float sliver (5 + 1) at the front, which joins its
    # neighbour (a tail below a quarter of the bucket size would be a lone latency-bound collective after backward has ended)
    assert sizes == [33, 35, 46] and sum(sizes) == arena.numel, sizes
    assert sorted(ddp.buckets[-1][2]) == [0, 1, 2, 3] and ddp.describe()['n_buckets'] == 3
    names = {id(p): n for n, p in model.named_parameters()}
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(16, 3, generator=g), torch.randn(16, 1, generator=g)
    res = dict(sizes=sizes, grads=[], traces=[], learned=[], layout=[], values=[])
    for step, plan in enumerate(_SKIP_PLAN):
        arena.zero_grad()
        loss = (ddp(X[rank * 4:(rank + 1) * 4], plan[rank]) - Y[rank * 4:(rank + 1) * 4]).pow(2).mean()
        loss.backward()
        ddp.finish()
        res['grads'].append({names[id(p)]: v.clone() for p, v in zip(arena.params, arena.views)})
        res['traces'].append(ddp.bucket_trace())
        res['learned'].append(bool(ddp._order_learned))
        res['layout'].append(dict(order=[names[id(p)] for p in arena.params], sizes=[hi - lo for lo, hi, _ in ddp.buckets]))
        # parameter VALUES travel with their slices, and p.data / p.grad stay views of the (new) arena
        assert all(p.data_ptr() == arena.flat_param[o:o + 1].data_ptr() for p, o in zip(arena.params, arena.offsets))
        with torch.no_grad():
            arena.flat_param.add_(arena.flat_grad, alpha=-0.05)      # a plain SGD step through the arena
        res['values'].append({n: p.detach().clone() for n, p in model.named_parameters()})
    torch.save(res, os.path.join(tmp, f'w{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def _worker_silent(rank, world, port, tmp, edge_mb):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), GE_DDP_EDGE_MB=edge_mb)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(3)
    model = Branchy()
    arena = GradArena(model.parameters(), align=1, adopt=True)
    ddp = FlatDDP(model, arena, bucket_mb=30 * 4 / 2 ** 20)
    names = {id(p): n for n, p in model.named_parameters()}
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(8, 3, generator=g), torch.randn(8, 1, generator=g)
    learned, orders, grads = [], [], []
    for step in range(5):
        arena.zero_grad()
        use_skip = (edge_mb == '0')                          # re-layout off: every parameter reports; else `skip` NEVER gets a gradient
        if step == 1 and edge_mb != '0':                     # a step with two backward passes: hooks fire twice (repeats are dropped)
            (ddp(X[rank * 4:(rank + 1) * 4], False) - Y[rank * 4:(rank + 1) * 4]).pow(2).mean().backward()
        (ddp(X[rank * 4:(rank + 1) * 4], use_skip) - Y[rank * 4:(rank + 1) * 4]).pow(2).mean().backward()
        ddp.finish()
        learned.append(bool(ddp._order_learned))
        orders.append(list(ddp.order))
        grads.append({names[id(p)]: v.clone() for p, v in zip(arena.params, arena.views)})
    torch.save(dict(learned=learned, orders=orders, grads=grads, layout=[names[id(p)] for p in arena.params],
                    arrival=ddp.arrival_order, buckets=[sorted(m) for _, _, m in ddp.buckets]), os.path.join(tmp, f's{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('edge_mb', ['8', '0'])
def test_flat_ddp_arrival_order_terminates_and_edge0_keeps_launch_order(tmp_path, edge_mb):
    """(a) A parameter that NEVER receives a gradient (and a step whose hooks fire twice): the arrival order used to stay unlearned for ever —
    one blocking broadcast + host sync per step, and every hipGraph capture failing on it.  Now the third incomplete step adopts rank 0's
    sequence with the silent parameter last, on every rank in the same step; gradients stay the mean over ranks throughout.
    (b) ``GE_DDP_EDGE_MB=0`` (no arena re-layout): the registration-order buckets are still LAUNCHED in the order in which they complete."""
    port = free_port()
    mp.spawn(_worker_silent, args=(2, port, str(tmp_path), edge_mb), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f's{i}.pt', weights_only=False) for i in range(2))
    assert r0['learned'] == r1['learned'] and r0['orders'] == r1['orders'] and r0['layout'] == r1['layout']
    for a, b in zip(r0['grads'], r1['grads']):
        assert all(torch.equal(a[k], b[k]) for k in a)
    if edge_mb == '0':
        assert r0['learned'] == [True] * 5                   # complete at the first step
        assert r0['layout'] == [n for n, _ in Branchy().named_parameters()]          # the arena did not move
        when = {idx: t for t, idx in enumerate(r0['arrival'])}
        done = [max(when[m] for m in members) for members in r0['buckets']]
        assert r0['orders'][-1] == sorted(range(len(done)), key=lambda b: done[b]) and sorted(r0['orders'][-1]) == list(range(len(done)))
    else:
        assert r0['learned'] == [False, False, True, True, True]
        assert r0['layout'][-2:] == ['skip.bias', 'skip.weight'] or set(r0['layout'][-2:]) == {'skip.weight', 'skip.bias'}      # silent parameters last
        assert all(torch.count_nonzero(g['skip.weight']) == 0 for g in r0['grads'])


def test_flat_ddp_gloo_world4_ragged_buckets_and_late_gradients(tmp_path):
    """World size 4 (the N = 4 point of the driver's scaling run): buckets of unequal size with a partial tail bucket, gradients
    that become ready out of bucket order, a parameter without a gradient on SOME RANKS in some steps (its slice is reduced as zeros
    there), four iterations.  The arrival order is adopted by a collective decision: only when rank 0's sequence is complete, by every
    rank in the same step, whatever each rank saw locally; the arena is then permuted into arrival order and re-cut into buckets.
    Every rank must end each step with the mean of the per-rank gradients, and training through the permuted arena must equal
    training a plain replica."""
    port = free_port()
    mp.spawn(_worker4, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    rs = [torch.load(tmp_path / f'w{i}.pt', weights_only=False) for i in range(4)]
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(16, 3, generator=g), torch.randn(16, 1, generator=g)
    torch.manual_seed(3)                                    # rank 0's initial weights were broadcast to everyone
    ref = Branchy()
    for step, plan in enumerate(_SKIP_PLAN):
        for r in rs[1:]:
            assert r['grads'][step].keys() == rs[0]['grads'][step].keys()
            assert all(torch.equal(r['grads'][step][k], v) for k, v in rs[0]['grads'][step].items()), step
        want = {n: torch.zeros_like(p) for n, p in ref.named_parameters()}
        for rank in range(4):
            ref.zero_grad()
            (ref(X[rank * 4:(rank + 1) * 4], plan[rank]) - Y[rank * 4:(rank + 1) * 4]).pow(2).mean().backward()
            for n, p in ref.named_parameters():
                if p.grad is not None:
                    want[n] += p.grad / 4
        for n, v in want.items():
            assert torch.allclose(rs[0]['grads'][step][n], v, rtol=1e-5, atol=1e-7), (step, n)
        if not any(plan):                                    # `skip` had no gradient anywhere: its slice is exactly zero after the exchange
            assert torch.count_nonzero(rs[0]['grads'][step]['skip.weight']) == 0
        with torch.no_grad():
            for n, p in ref.named_parameters():
                p.add_(want[n], alpha=-0.05)
        for n, p in ref.named_parameters():
            assert torch.allclose(rs[2]['values'][step][n], p.detach(), rtol=1e-5, atol=1e-6), (step, n)
    # collective decision: step 0 (rank 0 incomplete) nobody learns; step 1 (rank 0 complete, rank 1 not) EVERYBODY does
    for r in rs:
        assert r['learned'] == [False, True, True, True]
        assert r['layout'] == rs[0]['layout']
    reg = ['late.weight', 'late.bias', 'body.weight', 'body.bias', 'skip.weight', 'skip.bias', 'early.weight', 'early.bias']
    assert rs[0]['layout'][0]['order'] == reg and rs[0]['layout'][0]['sizes'] == [33, 35, 46]
    arr = rs[0]['layout'][1]['order']
    # arrival order of rank 0's step 1: the `late` head first, `early` last (registration order has it the other way round)
    assert sorted(arr) == sorted(reg) and arr != reg
    assert {arr[0], arr[1]} == {'late.weight', 'late.bias'} and {arr[-1], arr[-2]} == {'early.weight', 'early.bias'}
    assert rs[0]['layout'][1] == rs[0]['layout'][2] == rs[0]['layout'][3]                 # learned once, then fixed
    new_sizes = rs[0]['layout'][1]['sizes']
    assert sum(new_sizes) == 114 and new_sizes[0] <= 30 and new_sizes[-1] <= 30, new_sizes
    # the trace (GE_DDP_TRACE=1): one record per bucket and step, completion after launch; the launch sequence is the SAME on every rank
    for step in range(4):
        seq0 = [t['bucket'] for t in rs[0]['traces'][step]]
        nb = len(rs[0]['layout'][step - 1]['sizes']) if step else 3       # a step runs on the layout the previous step left
        assert seq0 == list(range(nb)), (step, seq0)
        for r in rs:
            tr = r['traces'][step]
            assert [t['bucket'] for t in tr] == seq0, (step, seq0)
            assert all(t['done_ms'] >= t['launch_ms'] >= 0 for t in tr)


class BNNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(2, 4, 1)
        self.bn = nn.BatchNorm2d(4)

    def forward(self, x):
        return self.bn(self.conv(x))


def _bn_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from gedepth_amd.mmrt.runner import EvalHook
    torch.manual_seed(0)
    net = BNNet().train()
    net(torch.randn(8, 2, 5, 5, generator=torch.Generator().manual_seed(10 + rank)) * (1 + rank))     # per-rank statistics
    before = (net.bn.running_mean.clone(), net.bn.running_var.clone())

    import types
    R = types.SimpleNamespace(model=net, iter=0, rank=rank, work_dir=tmp, logger=lambda *a: None)
    seen = {}
    hook = EvalHook(lambda runner: seen.update(mean=runner.model.bn.running_mean.clone(), var=runner.model.bn.running_var.clone()) or {},
                    interval=1)
    hook.after_train_iter(R)
    torch.save(dict(before=before, seen=seen), os.path.join(tmp, f'bn{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_eval_hook_broadcasts_bn_buffers_before_distributed_eval(tmp_path):
    """depth/core/evaluation/eval_hooks.py:75-87: BatchNorm running statistics are per GPU during training; before a distributed
    evaluation rank 0's are broadcast, so that every rank scores its shard with the same model."""
    port = free_port()
    mp.spawn(_bn_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f'bn{i}.pt', weights_only=False) for i in range(2))
    assert not torch.equal(r0['before'][0], r1['before'][0]) and not torch.equal(r0['before'][1], r1['before'][1])
    for k, i in (('mean', 0), ('var', 1)):
        assert torch.equal(r0['seen'][k], r0['before'][i])           # rank 0 keeps its own statistics
        assert torch.equal(r1['seen'][k], r0['before'][i])           # rank 1 evaluates with rank 0's


def test_grad_arena_permute_moves_values_views_shadow_and_followers():
    """GradArena.permute (FlatDDP's arrival-order re-layout): parameter / gradient / bf16-shadow values travel with their slices,
    ``p.data`` / ``p.grad`` / ``p._ge_lp`` are views of the new buffers (channels-last convolution weights keep their NHWC slice order),
    registered followers (optimizer moments) are remapped with the same index, and a second permutation composes."""
    from gedepth_amd.mmrt.optim import lowp
    torch.manual_seed(0)
    net = nn.Sequential(nn.Conv2d(4, 6, 3), nn.Linear(5, 3), nn.LayerNorm(7))
    net[0].weight.data = net[0].weight.data.contiguous(memory_format=torch.channels_last)
    arena = GradArena(net.parameters(), align=8, adopt=False)
    arena.enable_shadow()
    arena.refresh_shadow(copy=True)
    follower = {'m': torch.arange(arena.numel, dtype=torch.float32)}
    arena.on_permute(lambda remap: follower.update(m=remap(follower['m'])))
    params = list(net.parameters())
    want_p = [p.detach().clone() for p in params]
    for i, p in enumerate(params):
        p.grad.fill_(float(i + 1))
    old = {id(p): (off, p.numel()) for p, off in zip(arena.params, arena.offsets)}
    order = [3, 0, 5, 1, 4, 2]
    arena.permute(order)
    assert [id(p) for p in arena.params] == [id(params[i]) for i in order]
    for i, p in enumerate(params):
        j = [id(q) for q in arena.params].index(id(p))
        off = arena.offsets[j]
        assert torch.equal(p.detach(), want_p[i]) and bool((p.grad == i + 1).all())
        assert p.data_ptr() == arena.flat_param[off:off + 1].data_ptr() and p.grad.data_ptr() == arena.flat_grad[off:off + 1].data_ptr()
        assert p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last)
        assert lowp(p, torch.bfloat16).data_ptr() == arena.flat_shadow[off:off + 1].data_ptr()        # shadow still current
        assert torch.equal(p._ge_lp, p.detach().to(torch.bfloat16))
        o0, n = old[id(p)]
        assert torch.equal(follower['m'][off:off + n], torch.arange(o0, o0 + n, dtype=torch.float32))   # follower moved identically
    # autograd still accumulates into the (new) arena
    before = arena.flat_grad.clone()
    net[1](torch.ones(2, 5)).sum().backward()
    assert not torch.equal(before, arena.flat_grad)
    arena.permute([5, 4, 3, 2, 1, 0])
    assert all(torch.equal(p.detach(), w) for p, w in zip(params, want_p))
    with pytest.raises(AssertionError):
        arena.permute([0, 0, 1, 2, 3, 4])
