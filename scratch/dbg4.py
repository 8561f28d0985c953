# Which conv layer's MIOpen backward is imprecise in fp32 for the Swin-L shapes?
import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch, torch.nn as nn, torch.nn.functional as F
from test_model_gpu import build, set_exact, T
from oracle.fill import load_filled
g = np.load('/root/repo/tests/golden/e2e_L_A.npz')
dev = torch.device('cuda')
model = build('depthformer_a.py'); load_filled(model, 'e2e'); model = model.to(dev); set_exact(model)
rec = {}
for m_ in model.modules():
    if hasattr(m_, 'inplace'): m_.inplace = False
def mk(name):
    def fwd(m, inp, out): rec.setdefault(name, {})['x'] = inp[0].detach()
    def bwd(m, gin, gout): rec[name]['go'] = gout[0].detach(); rec[name]['gi'] = None if gin[0] is None else gin[0].detach()
    return fwd, bwd
for n, m in model.named_modules():
    if isinstance(m, (nn.Conv2d, nn.Linear)):
        f, b = mk(n); m.register_forward_hook(f); m.register_full_backward_hook(b)
img, gt, kgt = T(g['img']).to(dev), T(g['depth_gt']).to(dev), T(g['pe_k_gt']).to(dev)
metas = [dict(flip=False, ori_shape=(64, 96, 3))] * 2
model.train()
out = model.train_step(dict(img=img, img_metas=metas, depth_gt=gt, pe_k_gt=kgt), None)
out['loss'].backward()
mods = dict(model.named_modules())
rows = []
for n, r in rec.items():
    m = mods[n]
    if 'go' not in r: continue
    x = r['x'].cpu().double().requires_grad_(True); go = r['go'].cpu().double()
    w = m.weight.detach().cpu().double().requires_grad_(True)
    if isinstance(m, nn.Conv2d):
        y = F.conv2d(x, w, None, m.stride, m.padding)
    else:
        y = F.linear(x, w)
    y.backward(go)
    ew = ((m.weight.grad.cpu().double() - w.grad).norm() / (w.grad.norm() + 1e-30)).item()
    ei = -1 if r['gi'] is None else ((r['gi'].cpu().double() - x.grad).norm() / (x.grad.norm() + 1e-30)).item()
    rows.append((max(ew, ei), n, tuple(m.weight.shape), tuple(r['x'].shape), ew, ei))
rows.sort(reverse=True)
for r in rows[:25]:
    print(f'{r[1]:55s} w{r[2]} x{r[3]} dW {r[4]:.2e} dX {r[5]:.2e}')
print('---- norm layers')
rows = []
for n, m in model.named_modules():
    if isinstance(m, (nn.BatchNorm2d, nn.LayerNorm, nn.GELU)):
        pass
