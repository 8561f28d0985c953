"""Probe: the 1x1 convolutions of the HAHI neck (channels-last bf16) through MIOpen vs as token Linears (library GEMM + split-K weight
gradient of mmrt.bricks.linear_tokens).  Prints fwd+bwd time per layer for both."""
import sys, torch
sys.path.insert(0, '.')
from gedepth_amd.mmrt import bricks
from gedepth_amd.mmrt.tuning import use_miopen_find_db, use_tuned_gemms
use_miopen_find_db(); use_tuned_gemms('load')
torch.backends.cudnn.benchmark = True
dev = 'cuda'
shapes = [(64, 64, 176, 560), (96, 96, 88, 280), (192, 192, 44, 140), (384, 384, 22, 70), (768, 768, 11, 35),
          (64, 512, 176, 560), (96, 512, 88, 280), (192, 512, 44, 140), (384, 512, 22, 70), (768, 512, 11, 35)]
tot = [0.0, 0.0]
for ci, co, h, w in shapes:
    x = torch.randn(8, ci, h, w, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wt = (torch.randn(co, ci, 1, 1, device=dev) * 0.05).requires_grad_(True)
    go = torch.randn(8, co, h, w, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    def conv():
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = torch.nn.functional.conv2d(x, wt)
        y.backward(go)
    def lin():
        with torch.autocast('cuda', dtype=torch.bfloat16):
            t = x.permute(0, 2, 3, 1).reshape(8, h * w, ci)
            y = bricks.linear_tokens(t, wt.view(co, ci), None)
        y.backward(go.permute(0, 2, 3, 1).reshape(8, h * w, co))
    res = []
    for fn in (conv, lin):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / 10)
    tot[0] += res[0]; tot[1] += res[1]
    print(f'{ci:4d}->{co:4d} @{h}x{w}: conv2d {res[0]*1e3:8.1f} us   linear_tokens {res[1]*1e3:8.1f} us')
print(f'total: conv2d {tot[0]:.3f} ms, linear {tot[1]:.3f} ms')
