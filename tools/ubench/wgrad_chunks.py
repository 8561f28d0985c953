"""Round 6: the split-K weight gradient of the 261 800-token Linears (value_proj / output_proj of the HAHI neck): bmm over S token chunks + fp32 sum.
bricks._split_k picks S = 56 -> chunks of 4 675 tokens (odd); other divisors give chunks that are multiples of 8.   python tools/ubench/wgrad_chunks.py"""
import sys, torch
sys.path.insert(0, '.')
dev = 'cuda'
def t_of(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for tokens, M, N in ((261800, 512, 512), (261800, 768, 512), (788480, 512, 512), (788480, 768, 512), (197120, 96, 96), (197120, 288, 96), (49280, 192, 192)):
    dy = torch.randn(tokens, M, device=dev).bfloat16(); x = torch.randn(tokens, N, device=dev).bfloat16()
    res = []
    for s in range(64, 3, -1):
        if tokens % s or tokens // s < 1024: continue
        kc = tokens // s
        def f():
            part = torch.bmm(dy.view(s, kc, M).transpose(1, 2), x.view(s, kc, N))
            return part.sum(0, dtype=torch.float32)
        res.append((t_of(f), s, kc))
    def plain(): return torch.mm(dy.t(), x, out_dtype=torch.float32)
    tp = t_of(plain)
    res.sort()
    print(f'{tokens} x ({M} x {N}): plain mm {tp:.0f} us | best splits: ' + ', '.join(f'S={s} (chunk {kc}{"" if kc % 8 else " /8"}{"" if kc % 64 else " /64"}) {t:.0f} us' for t, s, kc in res[:5]) + f' | current rule: ' + ', '.join(f'S={s} {t:.0f} us' for t, s, kc in res if s == max(r[1] for r in res)))
