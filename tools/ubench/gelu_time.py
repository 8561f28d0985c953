"""bias + GELU epilogue kernels vs ATen's gelu / gelu_backward + column sum on the FFN shapes of the bench."""
import sys, torch
sys.path.insert(0, '/root/repo')
import torch.nn.functional as F
from gedepth_amd import kernels
dev = torch.device('cuda')
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for R, C in ((197120, 384), (49280, 768), (12320, 1536), (3080, 3072)):
    x = torch.randn(R, C, device=dev).bfloat16()
    b = torch.randn(C, device=dev)
    dg = torch.randn(R, C, device=dev).bfloat16()
    xb = x.float().add_(b).bfloat16()
    print(f'{R}x{C}: aten gelu {timeit(lambda: F.gelu(xb)):.0f} us | bias_gelu_fwd {timeit(lambda: kernels.bias_gelu_fwd(x, b)):.0f} us | '
          f'aten gelu_backward {timeit(lambda: torch.ops.aten.gelu_backward(dg, xb)):.0f} + colsum {timeit(lambda: kernels.colsum(dg)):.0f} us | '
          f'bias_gelu_bwd {timeit(lambda: kernels.bias_gelu_bwd(dg, x, b)):.0f} us | copy {timeit(lambda: x.clone()):.0f} us')
