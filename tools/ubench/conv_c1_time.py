"""ge_conv3x3_c1_fwd / _bwd (64 -> 1, 3x3) against MIOpen at the bench shape: python tools/ubench/conv_c1_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from gedepth_amd import hip
dev = torch.device('cuda:0')
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for N, H, W, Ci in [(8, 176, 560, 64), (2, 176, 560, 64), (1, 608, 968, 64)]:
    x = torch.randn(N, Ci, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(1, Ci, 3, 3, device=dev) / 24).contiguous(memory_format=torch.channels_last)
    wb = w.bfloat16()
    b = torch.randn(1, device=dev)
    w_ohwi = w.permute(0, 2, 3, 1).contiguous()
    for fp32 in (False, True):
        y = torch.empty(N, 1, H, W, device=dev, dtype=torch.float32 if fp32 else torch.bfloat16)
        go = torch.randn_like(y)
        dx = torch.empty_like(x); dw = torch.empty(1, 3, 3, Ci, device=dev); db = torch.empty(1, device=dev)
        f = lambda: hip.check(hip.lib().ge_conv3x3_c1_fwd(x.data_ptr(), w_ohwi.data_ptr(), b.data_ptr(), y.data_ptr(), N, H, W, Ci, 0 if fp32 else 1, hip.stream()), 'fwd')
        g = lambda: hip.check(hip.lib().ge_conv3x3_c1_bwd(x.data_ptr(), go.data_ptr(), w_ohwi.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), N, H, W, Ci, 0 if fp32 else 1, hip.stream()), 'bwd')
        f(); g()
        ref = F.conv2d(x.float(), wb.float(), b, padding=1)
        err = (y.float() - ref).abs().max().item() / ref.abs().max().item()
        tf, tb = timeit(f), timeit(g)
        mb = x.numel() * 2 / 1e6
        print(f'{N}x{Ci}->1 @{H}x{W} out {"f32" if fp32 else "bf16"}: rel err {err:.1e} | fwd {tf:6.1f} us ({mb / tf * 1e3:6.0f} GB/s) | bwd {tb:6.1f} us ({2 * mb / tb * 1e3:6.0f} GB/s)', flush=True)
    gl = torch.randn(N, 1, H, W, device=dev).bfloat16()
    tl = timeit(lambda: F.conv2d(x, wb, None, padding=1))
    tlb = timeit(lambda: torch.ops.aten.convolution_backward(gl, x, wb, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (True, True, False)))
    print(f'   MIOpen fwd {tl:6.1f} us | bwd (d + w) {tlb:6.1f} us', flush=True)
