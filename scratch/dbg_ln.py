import os, sys, json, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_model_gpu as M
from gedepth_amd.mmrt import bricks
torch.backends.cudnn.allow_tf32 = False; torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device('cuda:0')
g = np.load('/root/repo/tests/golden/e2e_T_V.npz', allow_pickle=True)
def run(aten_ln):
    orig = bricks.LayerNorm.forward
    if aten_ln:
        bricks.LayerNorm.forward = lambda self, x: torch.nn.LayerNorm.forward(self, x)
    model = M.build('depthformer_swint_v.py'); M.load_filled(model, 'e2e'); model = model.to(dev); M.set_exact(model); model.train()
    T = M.T
    img, gt = T(g['img']).to(dev), T(g['depth_gt']).to(dev)
    metas = [dict(flip=False, ori_shape=(64, 96, 3))] * 2
    out = model.train_step(dict(img=img, img_metas=metas, depth_gt=gt), None)
    out['loss'].backward()
    params = dict(model.named_parameters())
    errs = {}
    for k in g.files:
        if k.startswith('grad::'):
            gr = params[k[6:]].grad.flatten(); gr = gr[::max(1, gr.numel() // 50000)].cpu()
            refg = T(g[k]); errs[k[6:]] = ((gr - refg).norm() / (refg.norm() + 1e-30)).item()
    bricks.LayerNorm.forward = orig
    return errs
a, b = run(True), run(False)
for k in a:
    if b[k] > 5e-5 or a[k] > 5e-5: print(f'{k:70s} aten {a[k]:.2e}  hip {b[k]:.2e}')
print('max aten', max(a.values()), 'max hip', max(b.values()))
