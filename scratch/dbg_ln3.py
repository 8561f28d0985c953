import os, sys, json, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_model_gpu as M
from gedepth_amd.mmrt import bricks
from gedepth_amd import kernels
torch.backends.cudnn.allow_tf32 = False; torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device('cuda:0')
g = np.load('/root/repo/tests/golden/e2e_T_V.npz', allow_pickle=True)
log = []
class Probe(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps, name):
        ctx.save_for_backward(x, w, b); ctx.eps = eps; ctx.name = name
        y = kernels.layer_norm(x.detach(), w.detach(), b.detach(), eps)
        with torch.no_grad():
            yr = F.layer_norm(x.double(), (x.shape[-1],), w.double(), b.double(), eps)
            ya = F.layer_norm(x, (x.shape[-1],), w, b, eps)
        log.append((name, 'fwd', tuple(x.shape), x.is_contiguous(), ((y.double()-yr).norm()/yr.norm()).item(), ((ya.double()-yr).norm()/yr.norm()).item()))
        return y
    @staticmethod
    def backward(ctx, dy):
        x, w, b = ctx.saved_tensors
        res = []
        for kind in ('f64', 'aten', 'hip'):
            xx = (x.double() if kind == 'f64' else x).detach().requires_grad_(True)
            ww = (w.double() if kind == 'f64' else w).detach().requires_grad_(True)
            bb = (b.double() if kind == 'f64' else b).detach().requires_grad_(True)
            dd = dy.double() if kind == 'f64' else dy
            with torch.enable_grad():
                y = kernels.layer_norm(xx, ww, bb, ctx.eps) if kind == 'hip' else F.layer_norm(xx, (x.shape[-1],), ww, bb, ctx.eps)
                gx, gw, gb = torch.autograd.grad(y, (xx, ww, bb), dd)
            res.append((gx, gw, gb))
        rel = lambda a, r: ((a.double()-r).norm()/(r.norm()+1e-300)).item()
        log.append((ctx.name, 'bwd', tuple(dy.shape), dy.is_contiguous(), [rel(res[2][i], res[0][i]) for i in range(3)], [rel(res[1][i], res[0][i]) for i in range(3)]))
        return res[2][0], res[2][1], res[2][2], None, None
model = M.build('depthformer_swint_v.py'); M.load_filled(model, 'e2e'); model = model.to(dev); M.set_exact(model); model.train()
names = {id(m): n for n, m in model.named_modules()}
bricks.LayerNorm.forward = lambda self, x: Probe.apply(x, self.weight, self.bias, self.eps, names[id(self)])
T = M.T
img, gt = T(g['img']).to(dev), T(g['depth_gt']).to(dev)
out = model.train_step(dict(img=img, img_metas=[dict(flip=False, ori_shape=(64, 96, 3))] * 2, depth_gt=gt), None)
out['loss'].backward()
for r in log:
    if 'stages.0' in r[0] or 'patch_embed' in r[0] or 'norm0' in r[0] or 'stages.1.blocks.0' in r[0]: print(r)
