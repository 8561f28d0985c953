"""Time the MSDA fwd/bwd kernels at the bench shapes (B=8, cross- and self-attention of HAHI at 352x1120)."""
import sys, os, torch
sys.path.insert(0, '/root/repo')
from gedepth_amd import kernels
from gedepth_amd.kernels import ms_deform_attn
dev = torch.device('cuda')
shapes = [(88, 280), (44, 140), (22, 70), (11, 35)]
Nv = sum(h * w for h, w in shapes)
B = int(os.environ.get('B', 8))
torch.manual_seed(0)
for name, Nq in (('cross', 176 * 560), ('self', Nv)):
    value = torch.randn(B, Nv, 8, 64, device=dev).bfloat16().requires_grad_(True)
    ref = torch.sigmoid(torch.randn(B, Nq, 1, 1, 1, 2, device=dev))
    off = torch.randn(B, Nq, 8, 4, 8, 2, device=dev) * 4
    norm = torch.tensor([[w, h] for h, w in shapes], device=dev, dtype=torch.float32).view(1, 1, 1, 4, 1, 2)
    loc = (ref + off / norm).requires_grad_(True)
    aw = torch.rand(B, Nq, 8, 32, device=dev).softmax(-1).view(B, Nq, 8, 4, 8).requires_grad_(True)
    go = torch.randn(B, Nq, 512, device=dev).bfloat16()
    for binned in (True, False):
        kernels.MSDA_BINNED_BACKWARD = binned
        for it in range(3):
            if it == 1:
                kernels.PROFILER.enable()
            out = ms_deform_attn(value, shapes, loc, aw)
            out.backward(go)
            value.grad = loc.grad = aw.grad = None
        kernels.PROFILER.disable()
        for r in kernels.PROFILER.summary():
            print(name, 'binned' if binned else 'atomic', r['name'], f"{r['avg_us'] / 1e3:.2f} ms")
