"""LayerNorm backward at the Swin shapes (optionally with another library: GE_LIB=...): python tools/ubench/ln_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gedepth_amd import hip
if os.environ.get('GE_LIB'):
    hip.LIB_PATH = os.path.abspath(os.environ['GE_LIB'])
from gedepth_amd import kernels
dev = torch.device('cuda:0')
for rows, C, dt in [(197120, 96, torch.float32), (197120, 96, torch.bfloat16), (49280, 192, torch.bfloat16), (12320, 384, torch.bfloat16), (3080, 768, torch.bfloat16), (9196, 768, torch.bfloat16),
                    (147136, 192, torch.float32)]:
    x = torch.randn(rows, C, device=dev).to(dt).requires_grad_(True)
    w, b = torch.ones(C, device=dev, requires_grad=True), torch.zeros(C, device=dev, requires_grad=True)
    go = torch.randn(rows, C, device=dev).bfloat16()
    def run():
        y = kernels.layer_norm(x, w, b, 1e-5, torch.bfloat16)
        y.backward(go)
        x.grad = None
    for _ in range(3): run()
    kernels.PROFILER.enable()
    for _ in range(20): run()
    torch.cuda.synchronize()
    kernels.PROFILER.disable()
    for r in kernels.PROFILER.summary():
        if 'layernorm_bwd' in r['name']:
            print(f"{r['name']:46s} {r['avg_us']:7.1f} us {r['bytes_per_launch'] / r['avg_us'] / 1e3:7.1f} GB/s")
    kernels.PROFILER.summary().clear() if hasattr(kernels.PROFILER.summary(), 'clear') else None
