"""Per-level accuracy of the deformable-attention backward (fp32) at the e2e fixture's geometry against float64."""
import sys
import torch
sys.path.insert(0, '/root/repo')
from oracle import gedepth_oracle as O
from gedepth_amd.kernels import ms_deform_attn, msda_mode

dev = torch.device('cuda')
shapes = [(16, 24), (8, 12), (4, 6), (2, 3)]
nv = sum(h * w for h, w in shapes)
g = torch.Generator().manual_seed(3)
for name, nq in (('self', nv), ('cross', 32 * 48)):
    value = torch.randn(2, nv, 8, 64, generator=g)
    loc = torch.rand(2, nq, 8, 4, 8, 2, generator=g) * 1.2 - 0.1
    aw = torch.rand(2, nq, 8, 32, generator=g).softmax(-1).view(2, nq, 8, 4, 8)
    go = torch.randn(2, nq, 512, generator=g)
    v64, l64, a64 = (t.double().requires_grad_(True) for t in (value, loc, aw))
    O.msda_core(v64, shapes, l64, a64).backward(go.double())
    v32, l32, a32 = (t.clone().requires_grad_(True) for t in (value, loc, aw))
    O.msda_core(v32, shapes, l32, a32).backward(go)
    for mode in (0, 7):
        msda_mode(mode)
        vg, lg, ag = (t.to(dev).requires_grad_(True) for t in (value, loc, aw))
        out = ms_deform_attn(vg, shapes, lg, ag, query_shapes=shapes if name == 'self' else [(32, 48)])
        out.backward(go.to(dev))
        rel = lambda a, b: ((a.double().cpu() - b).norm() / b.norm()).item()
        s = 0
        for l, (h, w) in enumerate(shapes):
            sl = slice(s, s + h * w)
            print(f'{name} mode {mode} d_value level {l}: HIP {rel(vg.grad[:, sl], v64.grad[:, sl]):.2e}  cpu32 {rel(v32.grad[:, sl], v64.grad[:, sl]):.2e}   '
                  f'd_loc: HIP {rel(lg.grad[:, :, :, l], l64.grad[:, :, :, l]):.2e} cpu32 {rel(l32.grad[:, :, :, l], l64.grad[:, :, :, l]):.2e}   '
                  f'd_attw: HIP {rel(ag.grad[:, :, :, l], a64.grad[:, :, :, l]):.2e} cpu32 {rel(a32.grad[:, :, :, l], a64.grad[:, :, :, l]):.2e}')
            s += h * w
