"""Split-K sweep for the weight-gradient GEMMs of the Swin stages with FEW tokens (stage 2: 12320 x 384, stage 3: 3080 x 768 at
8 x 352 x 1120): dW = dY^T X has 9 - 36 output tiles of 256 x 256 on a 256-CU chip (tools/library_roofline.py: 66 - 215 TFLOP/s)."""
import sys, torch
sys.path.insert(0, '/root/repo')
from gedepth_amd.mmrt.tuning import use_tuned_gemms
use_tuned_gemms('load')
dev = torch.device('cuda')
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for K, M, N in ((12320, 1536, 384), (12320, 384, 1536), (12320, 1152, 384), (12320, 384, 384), (3080, 3072, 768), (3080, 768, 3072),
                (3080, 2304, 768), (3080, 768, 768), (49280, 768, 192), (49280, 192, 768), (49280, 576, 192), (49280, 192, 192),
                (24640, 1536, 384), (6160, 3072, 768)):
    dy = torch.randn(K, M, device=dev, dtype=torch.bfloat16)
    x = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    t0 = timeit(lambda: dy.t() @ x)
    line = f'K={K} M={M} N={N}: plain {t0*1e3:.0f} us ({2*K*M*N/t0/1e9:.0f} TF/s)'
    for S in (2, 4, 5, 7, 8, 10, 11, 14, 16, 20, 22, 28, 32, 40, 44, 56, 64):
        if K % S or K // S < 128: continue
        def f():
            p = torch.bmm(dy.view(S, K // S, M).transpose(1, 2), x.view(S, K // S, N))
            return p.sum(0, dtype=torch.float32)
        t = timeit(f)
        line += f' | S{S} {t*1e3:.0f}'
    print(line, flush=True)
