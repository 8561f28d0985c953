import sys, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from gedepth_amd.kernels import layer_norm
dev = torch.device('cuda')
torch.manual_seed(0)
for C in (96, 192, 384, 768):
    for scale, shift in ((1.0, 0.0), (2.0, 5.0), (0.1, 30.0)):
        x = (torch.randn(768, C, dtype=torch.float64) * scale + shift)
        w, b = torch.randn(C, dtype=torch.float64), torch.randn(C, dtype=torch.float64)
        go = torch.randn(768, C, dtype=torch.float64)
        xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
        yr = F.layer_norm(xr, (C,), wr, b, 1e-5); yr.backward(go)
        res = {}
        for name in ('aten', 'hip'):
            xg = x.float().to(dev).requires_grad_(True); wg = w.float().to(dev).requires_grad_(True); bg = b.float().to(dev).requires_grad_(True)
            y = F.layer_norm(xg, (C,), wg, bg, 1e-5) if name == 'aten' else layer_norm(xg, wg, bg, 1e-5)
            y.backward(go.float().to(dev))
            rel = lambda a, r: ((a.double().cpu() - r).norm() / r.norm()).item()
            res[name] = (rel(y, yr.detach()), rel(xg.grad, xr.grad), rel(wg.grad, wr.grad))
        print(C, scale, shift, 'aten y/dx/dw %.1e %.1e %.1e' % res['aten'], ' hip %.1e %.1e %.1e' % res['hip'])
