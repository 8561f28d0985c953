import sys, os, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from test_model_gpu import build, set_exact, T
from oracle.fill import load_filled
g = np.load('/root/repo/tests/golden/e2e_L_A.npz')
dev = torch.device('cuda')
model = build('depthformer_a.py'); load_filled(model, 'e2e'); model = model.to(dev); set_exact(model)
img, gt, kgt = T(g['img']).to(dev), T(g['depth_gt']).to(dev), T(g['pe_k_gt']).to(dev)
metas = [dict(flip=False, ori_shape=(64, 96, 3))] * 2
model.train()
out = model.train_step(dict(img=img, img_metas=metas, depth_gt=gt, pe_k_gt=kgt), None)
print(dict(out['log_vars']), g['loss_values'])
out['loss'].backward()
params = dict(model.named_parameters())
for k in g.files:
    if k.startswith('grad::'):
        gr = params[k[6:]].grad.flatten(); gr = gr[::max(1, gr.numel() // 50000)].cpu().double()
        ref = T(g[k]).double()
        print(f'{k[6:]:70s} max {((gr-ref).abs().max()/ref.abs().max()).item():.2e} l2rel {((gr-ref).norm()/ref.norm()).item():.2e}')
