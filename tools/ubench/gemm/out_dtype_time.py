"""Weight-gradient GEMMs dW = dY^T X of the Swin-L stage-3 Linears at 2 images per GPU (3 080 tokens): bf16 result + the widening copy into
the fp32 gradient arena, against torch.mm(..., out_dtype=torch.float32, out=arena slice) — one kernel, no rounding of the result."""
import sys, torch
sys.path.insert(0, '.')
dev = 'cuda'
torch.manual_seed(0)


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, K, N) in [(3072, 3080, 768), (768, 3080, 3072), (2304, 3080, 768), (768, 3080, 768), (1536, 12320, 384), (6144, 770, 1536), (1536, 770, 6144)]:
    dy = torch.randn(K, M, device=dev).bfloat16()
    x = torch.randn(K, N, device=dev).bfloat16()
    arena = torch.zeros(M, N, device=dev)
    a = t(lambda: arena.copy_(torch.mm(dy.t(), x)))
    try:
        b = t(lambda: torch.mm(dy.t(), x, out_dtype=torch.float32, out=arena))
        ref = torch.mm(dy.t().float(), x.float())
        err = ((arena - ref).norm() / ref.norm()).item()
    except Exception as ex:
        b, err = float('nan'), repr(ex)[:120]
    print(f'dW {M}x{N} over {K} tokens: bf16 + copy {a:7.1f} us   fp32 out {b:7.1f} us   rel err {err}')
