# per-module backward precision probe for norm / activation layers in the L model (fp32)
import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch, torch.nn as nn, torch.nn.functional as F, copy
from test_model_gpu import build, set_exact, T
from oracle.fill import load_filled
_relu = F.relu
F.relu = lambda x, inplace=False: _relu(x)
g = np.load('/root/repo/tests/golden/e2e_L_A.npz')
dev = torch.device('cuda')
model = build('depthformer_a.py'); load_filled(model, 'e2e'); model = model.to(dev); set_exact(model)
for m_ in model.modules():
    if hasattr(m_, 'inplace'): m_.inplace = False
rec = {}
def mk(name):
    def fwd(m, inp, out): rec.setdefault(name, {})['x'] = inp[0].detach()
    def bwd(m, gin, gout): rec[name]['go'] = gout[0].detach(); rec[name]['gi'] = None if gin[0] is None else gin[0].detach()
    return fwd, bwd
KINDS = (nn.BatchNorm2d, nn.LayerNorm, nn.GELU, nn.LeakyReLU, nn.ReLU, nn.Sigmoid)
for n, m in model.named_modules():
    if isinstance(m, KINDS):
        f, b = mk(n); m.register_forward_hook(f); m.register_full_backward_hook(b)
img, gt, kgt = T(g['img']).to(dev), T(g['depth_gt']).to(dev), T(g['pe_k_gt']).to(dev)
metas = [dict(flip=False, ori_shape=(64, 96, 3))] * 2
model.train()
# snapshot BN running stats before the step (they are updated in train mode)
out = model.train_step(dict(img=img, img_metas=metas, depth_gt=gt, pe_k_gt=kgt), None)
out['loss'].backward()
params = dict(model.named_parameters())
for k in g.files:
    if k.startswith('grad::') and ('patch_embed' in k or 'level_embed' in k):
        gr = params[k[6:]].grad.flatten(); gr = gr[::max(1, gr.numel() // 50000)].cpu().double(); ref = T(g[k]).double()
        print(f'  (inplace=False) {k[6:]:60s} l2rel {((gr-ref).norm()/ref.norm()).item():.2e}')
mods = dict(model.named_modules())
rows = []
for n, r in rec.items():
    if 'go' not in r or r['gi'] is None: continue
    m = copy.deepcopy(mods[n]).cpu().double()
    m.train()
    x = r['x'].cpu().double().requires_grad_(True)
    y = m(x); y.backward(r['go'].cpu().double())
    e = ((r['gi'].cpu().double() - x.grad).norm() / (x.grad.norm() + 1e-30)).item()
    rows.append((e, n, type(mods[n]).__name__, tuple(r['x'].shape)))
rows.sort(reverse=True)
for r in rows[:15]: print(f'{r[1]:55s} {r[2]:12s} x{r[3]} dX {r[0]:.2e}')

print('---- standalone BN probe')
for shape in [(2,1536,2,3),(2,768,4,6),(2,384,8,12),(2,192,16,24),(2,512,2,3),(2,512,16,24),(2,64,32,48),(2,768,2,3)]:
    torch.manual_seed(0)
    bn = nn.BatchNorm2d(shape[1]).to(dev).train()
    bn.weight.data.normal_(1, 0.1); bn.bias.data.normal_(0, 0.1)
    x = torch.randn(shape, device=dev, requires_grad=True); go = torch.randn(shape, device=dev)
    y = bn(x); y.backward(go)
    bc = copy.deepcopy(bn).cpu().double(); bc.running_mean.zero_(); bc.running_var.fill_(1)
    xc = x.detach().cpu().double().requires_grad_(True); yc = bc(xc); yc.backward(go.cpu().double())
    print(shape, 'y', ((y.detach().cpu().double()-yc).norm()/yc.norm()).item(), 'dx', ((x.grad.cpu().double()-xc.grad).norm()/xc.grad.norm()).item(),
          'dw', ((bn.weight.grad.cpu().double()-bc.weight.grad).norm()/bc.weight.grad.norm()).item())
