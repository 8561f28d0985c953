"""Time the deformable-attention kernels (window vs streaming) at the bench shapes with the sampling geometry the HAHI neck
has at initialisation: offsets = the module's bias pattern (+ noise), self-attention reference = pixel centres,
cross-attention reference = sigmoid(Linear(pos_embed)).    MODE: 3 window, 0 streaming."""
import os
import sys
import torch
sys.path.insert(0, '/root/repo')
from gedepth_amd import hip
if os.environ.get('GE_LIB'):                                  # A/B a differently built library
    hip.LIB_PATH = os.path.abspath(os.environ['GE_LIB'])
from gedepth_amd import kernels
from gedepth_amd.kernels import ms_deform_attn, msda_mode
from gedepth_amd.mmrt import bricks
from gedepth_amd.depth.utils.position_encoding import SinePositionalEncoding

dev = torch.device('cuda')
H, W = int(os.environ.get('H', 352)), int(os.environ.get('W', 1120))
B = int(os.environ.get('B', 8))
NOISE = float(os.environ.get('NOISE', 1.0))
shapes = [(H // 4 // 2 ** i, W // 4 // 2 ** i) for i in range(4)]
Nv = sum(h * w for h, w in shapes)
torch.manual_seed(0)
bias = bricks.msda_offset_bias(8, 4, 8).view(1, 1, 8, 4, 8, 2).to(dev)
norm = torch.tensor([[w, h] for h, w in shapes], device=dev, dtype=torch.float32).view(1, 1, 1, 4, 1, 2)
pe = SinePositionalEncoding(num_feats=256)


def centres(shp):
    pts = []
    for h, w in shp:
        gy, gx = torch.meshgrid((torch.arange(h, device=dev) + 0.5) / h, (torch.arange(w, device=dev) + 0.5) / w, indexing='ij')
        pts.append(torch.stack((gx.reshape(-1), gy.reshape(-1)), -1))
    return torch.cat(pts, 0)


for name, qshapes in (('cross', [(H // 2, W // 2)]), ('self', shapes)):
    Nq = sum(h * w for h, w in qshapes)
    if name == 'self':
        ref = centres(qshapes)
    else:
        lin = torch.nn.Linear(512, 2).to(dev)
        torch.nn.init.xavier_uniform_(lin.weight)
        torch.nn.init.zeros_(lin.bias)
        pm = pe.grid(qshapes[0][0], qshapes[0][1], dev).flatten(2)[0]
        ref = torch.sigmoid(pm.t() @ lin.weight.t().detach())
    value = torch.randn(B, Nv, 8, 64, device=dev).bfloat16().requires_grad_(True)
    off = bias + NOISE * torch.randn(B, Nq, 8, 4, 8, 2, device=dev)
    loc = (ref.view(1, Nq, 1, 1, 1, 2) + off / norm).detach().requires_grad_(True)
    aw = torch.rand(B, Nq, 8, 32, device=dev).softmax(-1).view(B, Nq, 8, 4, 8).requires_grad_(True)
    go = torch.randn(B, Nq, 512, device=dev).bfloat16()
    res = {}
    for mode in ([int(m) for m in os.environ['MODES'].split(',')] if os.environ.get('MODES') else (13, 61, 29)):
        msda_mode(mode)
        for it in range(4):
            if it == 1:
                kernels.PROFILER.enable()
            out = ms_deform_attn(value, shapes, loc, aw, query_shapes=qshapes)
            out.backward(go)
            if it == 3:
                res[mode] = (out.detach().float(), loc.grad.clone(), aw.grad.clone(), value.grad.float().clone())
            value.grad = loc.grad = aw.grad = None
        kernels.PROFILER.disable()
        for r in kernels.PROFILER.summary() + kernels.PROFILER.msda_bwd_stages():
            print(f'{name:5s} mode {mode} {r["name"]:48s} {r["avg_us"] / 1e3:8.3f} ms')
    modes = list(res)
    for m in modes[1:]:
        for i, n in enumerate(('out', 'd_loc', 'd_attw', 'd_value')):
            a, b = res[m][i], res[modes[0]][i]
            print(f'{name} {n}: max |mode {m} - mode {modes[0]}| = {(a - b).abs().max().item():.3e} (scale {b.abs().max().item():.3e})')
