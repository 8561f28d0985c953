"""Localise the fp32 gradient gap of the Swin-L-A end-to-end step: activations and gradients at the module boundaries
(backbone outputs, neck outputs) of the product on the GPU vs the CPU oracle in float64 / float32."""
import json
import os
import sys
import numpy as np
import torch
ROOT = '/root/repo'
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT + '/tests')
from oracle import gedepth_oracle as O
from oracle.fill import fill_state_dict, load_filled
from gedepth_amd.depth.models import build_depther
from gedepth_amd.mmrt.config import Config

torch.set_num_threads(16)
tag = sys.argv[1] if len(sys.argv) > 1 else 'e2e_L_A'
g = np.load(f'{ROOT}/tests/golden/{tag}.npz')
arch = dict(O.SWIN_L if '_L_' in tag else O.SWIN_T, adaptive=tag.endswith('A'))
T = lambda a: torch.from_numpy(np.asarray(a))
img, gt, kgt = T(g['img']), T(g['depth_gt']), T(g['pe_k_gt'])


def l2rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return ((a - b).norm() / (b.norm() + 1e-300)).item()


def oracle(dtype):
    spec = json.loads(str(g['spec']))
    P = {}
    for k, v in fill_state_dict([(k, s) for k, s in spec], 'e2e').items():
        if v.is_floating_point():
            v = v.to(dtype)
            if not k.endswith(('running_mean', 'running_var')):
                v.requires_grad_(True)
        P[k] = v
    x = O.backbone(img.to(dtype), P, arch, True)
    for t in x:
        t.retain_grad()
    n = O.hahi_neck(x, P, True)
    for t in n:
        t.retain_grad()
    y = O.pe_mask_neck(n, P)
    y = torch.nn.functional.interpolate(y, size=img.shape[2:], mode='bilinear')
    if arch['adaptive']:
        pe_mask, logits, _ = O.dynamic_pe(O.dynamic_pe_neck(n, P), y, img[:, 4].to(dtype), 1.65, 200.0)
    else:
        pe_mask, logits = O.vanilla_pe(y, img[:, 3].to(dtype)), None
    pred = O.depth_pred(O.densedepth_head(n, P), pe_mask, y, P)
    pred_up = torch.nn.functional.interpolate(pred, size=gt.shape[2:], mode='bilinear', align_corners=True)
    loss = O.sigloss(pred_up, gt.to(dtype))
    if logits is not None:
        loss = loss + O.ce_loss(logits, kgt)
    loss.backward()
    return x, n, P


x64, n64, P64 = oracle(torch.float64)
x32, n32, P32 = oracle(torch.float32)

cfgname = {'e2e_L_A': 'depthformer_a.py', 'e2e_T_A': 'depthformer_swint_a.py', 'e2e_T_V': 'depthformer_swint_v.py'}[tag]
cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', cfgname))
cfg.model.pretrained = None
cfg.model.backbone.drop_path_rate = 0.0
m = build_depther(cfg.model)
m.neck.multi_att.dropout.p = 0.0
m.neck.self_attn.dropout.p = 0.0
load_filled(m, 'e2e')
dev = torch.device('cuda')
m = m.to(dev).train()
for mod in m.modules():
    if hasattr(mod, 'kernel_variant'):
        mod.kernel_variant = 1
cap = {}


def grab(name):
    def hook(module, inp, out):
        outs = list(out) if isinstance(out, (tuple, list)) else [out]
        for i, t in enumerate(outs):
            t.retain_grad()
            cap[f'{name}{i}'] = t
    return hook


m.backbone.register_forward_hook(grab('x'))
m.neck.register_forward_hook(grab('n'))
metas = [dict(flip=False, ori_shape=(64, 96, 3))] * 2
kw = dict(pe_k_gt=kgt.to(dev)) if arch['adaptive'] else {}
out = m.train_step(dict(img=img.to(dev), img_metas=metas, depth_gt=gt.to(dev), **kw), None)
out['loss'].backward()
for i in range(5):
    print(f'backbone out {i}: act HIP {l2rel(cap[f"x{i}"], x64[i]):.2e} cpu32 {l2rel(x32[i], x64[i]):.2e} | grad HIP '
          f'{l2rel(cap[f"x{i}"].grad, x64[i].grad):.2e} cpu32 {l2rel(x32[i].grad, x64[i].grad):.2e}')
for i in range(5):
    print(f'neck out {i}:     act HIP {l2rel(cap[f"n{i}"], n64[i]):.2e} cpu32 {l2rel(n32[i], n64[i]):.2e} | grad HIP '
          f'{l2rel(cap[f"n{i}"].grad, n64[i].grad):.2e} cpu32 {l2rel(n32[i].grad, n64[i].grad):.2e}')
rows = []
for k, p in m.named_parameters():
    r = P64[k].grad
    if r.norm() < 1e-12:
        continue
    rows.append((l2rel(p.grad, r) / max(l2rel(P32[k].grad, r), 3e-5), l2rel(p.grad, r), l2rel(P32[k].grad, r), k))
rows.sort(reverse=True)
for _, a, b, k in rows[:25]:
    print(f'{k:72s} HIP {a:.2e}  cpu32 {b:.2e}')
