"""Is split-K via batched GEMM faster than the library's single wgrad GEMM for huge-K, small-MxN shapes?"""
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
for K, M, N in ((788480, 768, 512), (788480, 512, 512), (261800, 768, 512), (261800, 512, 512), (197120, 384, 96), (197120, 96, 384),
                (197120, 288, 96), (197120, 96, 96), (49280, 768, 192), (49280, 192, 768), (49280, 576, 192), (49280, 192, 192)):
    dy = torch.randn(K, M, device=dev, dtype=torch.bfloat16)
    x = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    ref = None
    t0 = timeit(lambda: dy.t() @ x)
    ref = (dy.t().float() @ x.float())
    line = f'K={K} M={M} N={N}: plain {t0:.3f} ms ({2*K*M*N/t0/1e9:.0f} TF/s)'
    for S in (4, 8, 16, 32, 64):
        if K % S: continue
        def f():
            p = torch.bmm(dy.view(S, K // S, M).transpose(1, 2), x.view(S, K // S, N))
            return p.float().sum(0)
        t = timeit(f)
        err = ((f() - ref).norm() / ref.norm()).item()
        line += f' | S{S} {t:.3f}'
    err0 = (((dy.t() @ x).float() - ref).norm() / ref.norm()).item()
    line += f' | relerr plain {err0:.1e} splitK(last) {err:.1e}'
    print(line, flush=True)
