"""Does MIOpen run channels_last convolutions without batched_transpose kernels?  Times conv fwd+bwd for the neck / head
shapes in NCHW and channels_last and lists the kernels the profiler sees."""
import os, sys, time
import torch
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda')
torch.backends.cudnn.benchmark = os.environ.get('BENCH', '1') == '1'
shapes = [(8, 576, 64, 176, 560, 3), (8, 64, 512, 176, 560, 1), (8, 704, 96, 88, 280, 3), (8, 160, 64, 176, 560, 3), (8, 96, 64, 88, 280, 3)]
for fmt in ('nchw', 'nhwc'):
    for (B, Ci, Co, H, W, k) in shapes:
        conv = torch.nn.Conv2d(Ci, Co, k, padding=k // 2, bias=False).to(dev).bfloat16()
        x = torch.randn(B, Ci, H, W, device=dev, dtype=torch.bfloat16, requires_grad=True)
        if fmt == 'nhwc':
            conv = conv.to(memory_format=torch.channels_last)
            x = x.detach().to(memory_format=torch.channels_last).requires_grad_(True)
        for _ in range(3):
            y = conv(x); y.sum().backward()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            y = conv(x); y.backward(torch.ones_like(y))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            y = conv(x); y.backward(torch.ones_like(y)); torch.cuda.synchronize()
        names = sorted({e.key[:60] for e in prof.key_averages() if e.device_time_total > 0})
        tr = sum(e.device_time_total for e in prof.key_averages() if 'transpose' in e.key.lower()) / 1e3
        print(f'{fmt} {Ci}->{Co} k{k} @{H}x{W}: {dt * 1e3:.2f} ms fwd+bwd; out is channels_last: {y.is_contiguous(memory_format=torch.channels_last)}; transpose kernels {tr:.2f} ms')
        print('     ', [n for n in names if 'transpose' in n.lower() or 'igemm' in n.lower() or 'conv' in n.lower() or 'Cijk' in n][:8])
