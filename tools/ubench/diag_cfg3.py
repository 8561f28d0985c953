"""Config #3 (Swin-L-Adaptive, 2 x 352 x 1120): where does |g_bf16| / |g_fp32| come from?  Per-parameter norm ratio and cosine of the
bf16 step's gradient against the fp32 step's, sorted by the tensor's share of |g_bf16|^2 - |g_fp32|^2, plus the totals per top-level
module.  A/B with GE_DISABLE=msda_mm,gemm,... in the environment.  Usage: python tools/ubench/diag_cfg3.py [runs]"""
import collections
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gedepth_amd.depth.datasets.synthetic import synthetic_batch
from gedepth_amd.depth.models import build_depther
from gedepth_amd.depth.models.utils import to_channels_last
from gedepth_amd.mmrt.config import Config
from gedepth_amd.mmrt.optim import build_optimizer

dev = torch.device('cuda:0')
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.manual_seed(0)
cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', 'depthformer_a.py'))
cfg.model.pretrained = None
cfg.model.backbone.drop_path_rate = 0.0                     # deterministic comparison, as tests/test_model_gpu.py::build
model = build_depther(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))
model.init_weights()
model = model.to(dev).train()
to_channels_last(model)
model.neck.multi_att.dropout.p = 0.0
model.neck.self_attn.dropout.p = 0.0
batch = synthetic_batch(2, 352, 1120, seed=1234, device=dev)

# the regressor's pre-activation c (decode_head.depth_pred: pred = relu(c) (1 - y) + pe + min_depth) and its gradient, per pass
from gedepth_amd import kernels as K
CAP = {}
_c1 = K.conv3x3_c1


def _c1_cap(conv, feat, **kw):
    c = _c1(conv, feat, **kw)
    CAP['c'] = c.detach().float().clone()
    if c.requires_grad:
        c.register_hook(lambda g: CAP.__setitem__('dc', g.detach().float().clone()))
    return c


K.conv3x3_c1 = _c1_cap


def _module_cap(mod, inp, out):                             # the library path of the same layer (fp32 pass when the HIP kernel declines)
    CAP['c'] = out.detach().float().clone()
    if out.requires_grad:
        out.register_hook(lambda g: CAP.__setitem__('dc', g.detach().float().clone()))


model.decode_head.conv_depth.register_forward_hook(_module_cap)
ref = model.train_step(batch, None)
ref['loss'].backward()
g32 = {n: p.grad.detach().double().flatten().clone() for n, p in model.named_parameters()}
cap32 = dict(CAP)
for p in model.parameters():
    p.grad = None
optimizer = build_optimizer(model, cfg.optimizer, cfg.optimizer_config.get('grad_clip'))
for r in range(runs):
    optimizer.zero_grad()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out = model.train_step(batch, optimizer)
    out['loss'].backward()
    optimizer.arena.collect()
    g16 = {n: p.grad.detach().double().flatten().clone() for n, p in model.named_parameters()}
    n32 = sum(v.pow(2).sum().item() for v in g32.values())
    n16 = sum(v.pow(2).sum().item() for v in g16.values())
    dot = sum((g16[n] * g32[n]).sum().item() for n in g32)
    print(f'run {r}: GE_DISABLE={os.environ.get("GE_DISABLE", "")} losses fp32 {dict(ref["log_vars"])} bf16 {dict(out["log_vars"])}')
    print(f'  |g16|/|g32| {(n16 / n32) ** 0.5:.4f}  cosine {dot / (n16 * n32) ** 0.5:.5f}')
    c32, c16, d32, d16 = cap32['c'].flatten(), CAP['c'].flatten(), cap32['dc'].flatten().double(), CAP['dc'].flatten().double()
    on32, on16 = c32 > 0, c16 > 0
    e32, e16 = d32.pow(2).sum().item(), d16.pow(2).sum().item()
    print(f'  regressor pre-activation c: {c32.numel()} px, active fp32 {on32.sum().item()} bf16 {on16.sum().item()}, '
          f'off->on {(~on32 & on16).sum().item()} on->off {(on32 & ~on16).sum().item()}; |c| quantiles (fp32) '
          f'{[round(v, 5) for v in c32.abs().quantile(torch.tensor([0.1, 0.5, 0.9], device=dev)).tolist()]}')
    print(f'  |dc|^2: fp32 {e32:.4e} bf16 {e16:.4e} (ratio of norms {(e16 / e32) ** 0.5:.4f}); share of fp32 |dc|^2 in its top 100 px '
          f'{d32.pow(2).topk(100).values.sum().item() / e32:.3f}, top 1000 px {d32.pow(2).topk(1000).values.sum().item() / e32:.3f}; '
          f'bf16 |dc|^2 on px that are off in fp32 {d16[~on32].pow(2).sum().item() / e16:.4f}, fp32 |dc|^2 on px off in bf16 '
          f'{d32[~on16].pow(2).sum().item() / e32:.4f}')
    top = d32.abs().topk(8).indices
    print('  top 8 px by |dc| (fp32): c fp32 / c bf16 / dc fp32 / dc bf16: ' + '; '.join(
        f'{c32[i].item():.4f} / {c16[i].item():.4f} / {d32[i].item():+.3e} / {d16[i].item():+.3e}' for i in top.tolist()))
    both = on32 & on16
    print(f'  on px active in both: |dc16|/|dc32| {(d16[both].pow(2).sum() / d32[both].pow(2).sum()).sqrt().item():.4f}, cosine '
          f'{(d16[both] * d32[both]).sum().item() / (d16[both].norm() * d32[both].norm()).item():.5f}')
    rows = []
    mods = collections.defaultdict(lambda: [0.0, 0.0, 0.0])
    for n in g32:
        a, b = g16[n], g32[n]
        sa, sb, sab = a.pow(2).sum().item(), b.pow(2).sum().item(), (a * b).sum().item()
        rows.append((sa - sb, n, (sa / max(sb, 1e-300)) ** 0.5, sab / max((sa * sb) ** 0.5, 1e-300), sb ** 0.5))
        key = '.'.join(n.split('.')[:3 if n.startswith('backbone.stages') else 2])
        mods[key][0] += sa; mods[key][1] += sb; mods[key][2] += sab
    print('  per module: share of |g32|^2, norm ratio, cosine')
    for k, (sa, sb, sab) in sorted(mods.items(), key=lambda kv: -kv[1][1]):
        print(f'    {k:40s} {sb / n32:8.4f}  {(sa / max(sb, 1e-300)) ** 0.5:7.4f}  {sab / max((sa * sb) ** 0.5, 1e-300):8.5f}')
    if r == 0:
        print('  top tensors by |g16|^2 - |g32|^2 (share of |g32|^2), norm ratio, cosine, |g32|')
        for d, n, ratio, c, nb in sorted(rows, key=lambda t: -abs(t[0]))[:25]:
            print(f'    {n:64s} {d / n32:+8.4f}  {ratio:7.4f}  {c:8.5f}  {nb:.3e}')
