"""ge_conv1x1_nhwc_wgrad (csrc/conv1x1_wgrad.hip) against the library's weight gradient (aten.convolution_backward -> MIOpen / CK) on the 1x1
convolutions of the HAHI neck at the bench shape (8 images): time, GB/s of algorithmic bytes, error vs float64."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gedepth_amd import kernels
from gedepth_amd.mmrt.tuning import use_miopen_find_db
torch.backends.cudnn.benchmark = bool(use_miopen_find_db())
dev = torch.device('cuda')


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


shapes = [(64, 64, 176, 560), (96, 96, 88, 280), (192, 192, 44, 140), (384, 384, 22, 70), (768, 768, 11, 35),
          (64, 512, 176, 560), (96, 512, 88, 280), (192, 512, 44, 140), (384, 512, 22, 70), (768, 512, 11, 35)]
N = int(os.environ.get('N', 8))
tot = [0.0, 0.0]
for ci, co, h, w in shapes:
    torch.manual_seed(0)
    x = torch.randn(N, ci, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, co, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    wt = torch.randn(co, ci, 1, 1, device=dev).bfloat16()
    lib = lambda: torch.ops.aten.convolution_backward(dy, x, wt, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, (False, True, False))[1]
    own = lambda: kernels.conv1x1_wgrad(x, dy)
    ref = torch.einsum('nohw,nihw->oi', dy.double(), x.double())
    err = ((own().double() - ref).abs().max() / ref.abs().max()).item()
    t_own, t_lib = timeit(own), timeit(lib)
    gb = (x.numel() + dy.numel()) * 2 / 1e9
    tot[0] += t_own
    tot[1] += t_lib
    print(f'1x1 {ci:4d}->{co:4d} @{h}x{w} N{N}: ours {t_own:7.1f} us ({gb / t_own * 1e6:6.0f} GB/s, incl. the zero fill)  library {t_lib:7.1f} us  rel err {err:.1e}')
print(f'total: ours {tot[0]:.0f} us, library {tot[1]:.0f} us')
