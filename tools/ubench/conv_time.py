"""ge_conv3x3_nhwc_fwd (hand-written implicit-GEMM MFMA convolution) against MIOpen through F.conv2d on the 3x3 shapes of the bench step:
correctness (vs fp32 conv of the bf16-rounded operands) and time, forward and as data gradient."""
import ctypes, sys, torch
sys.path.insert(0, '/root/repo')
import torch.nn.functional as F
from gedepth_amd import hip
from gedepth_amd.mmrt.tuning import use_miopen_find_db
torch.backends.cudnn.benchmark = bool(use_miopen_find_db())
dev = torch.device('cuda')
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
def ours(x_cl, w_ohwi, bias, act, slope):
    N, C, H, W = x_cl.shape
    Co = w_ohwi.shape[0]
    y = torch.empty((N, Co, H, W), device=dev, dtype=torch.bfloat16, memory_format=torch.channels_last)
    hip.check(hip.lib().ge_conv3x3_nhwc_fwd(x_cl.data_ptr(), w_ohwi.data_ptr(), None if bias is None else bias.data_ptr(), y.data_ptr(), N, H, W, C, Co,
                                             act, slope, 1, hip.stream()), 'ge_conv3x3_nhwc_fwd')
    return y
shapes = [(8, 176, 560, 576, 64), (8, 176, 560, 160, 64), (8, 176, 560, 64, 64), (8, 88, 280, 288, 96), (8, 88, 280, 608, 96), (8, 88, 280, 96, 96),
          (8, 44, 140, 576, 192), (8, 44, 140, 704, 192), (8, 22, 70, 1152, 384), (8, 22, 70, 896, 384), (8, 11, 35, 1280, 768), (2, 13, 37, 64, 96)]
import os
if os.environ.get('CONV_ONLY'):
    shapes = [(8, 176, 560, 576, 64), (8, 88, 280, 288, 96)]
if os.environ.get('CONV_SWINL'):                                   # config #3: Swin-L + GEDepth-Adaptive, 2 images per GPU
    shapes = [(2, 176, 560, 576, 64), (2, 88, 280, 704, 192), (2, 88, 280, 576, 192), (2, 44, 140, 1152, 384), (2, 22, 70, 2304, 768), (2, 176, 560, 64, 64),
              (2, 44, 140, 896, 384), (2, 176, 560, 256, 64), (2, 22, 70, 1280, 768), (2, 22, 70, 768, 768), (2, 11, 35, 2048, 1536), (2, 88, 280, 192, 192),
              (2, 44, 140, 384, 384), (2, 88, 280, 192, 64), (1, 608, 968, 576, 64), (1, 304, 484, 704, 192), (1, 152, 242, 1152, 384)]
for N, H, W, Ci, Co in shapes:
    torch.manual_seed(0)
    x = torch.randn(N, Ci, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Co, Ci, 3, 3, device=dev) / (3 * Ci ** 0.5)).bfloat16().contiguous(memory_format=torch.channels_last)
    b = torch.randn(Co, device=dev)
    w_ohwi = w.permute(0, 2, 3, 1).contiguous()                  # = the channels-last storage, as an explicit (O, H, W, I) tensor
    y = ours(x, w_ohwi, b, 1, 0.01)
    ref = F.leaky_relu(F.conv2d(x.float(), w.float(), b, padding=1), 0.01)
    err = (y.float() - ref).abs().max().item() / ref.abs().max().item()
    t_ours = timeit(lambda: ours(x, w_ohwi, b, 1, 0.01))
    t_lib = timeit(lambda: F.conv2d(x, w, None, padding=1))
    fl = 2 * N * H * W * Ci * Co * 9
    go = torch.randn(N, Co, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    def wg_ours():
        dw = torch.zeros(Co, 3, 3, Ci, device=dev, dtype=torch.float32)
        hip.check(hip.lib().ge_conv3x3_nhwc_wgrad(x.data_ptr(), go.data_ptr(), dw.data_ptr(), N, H, W, Ci, Co, 1, hip.stream()), 'wgrad')
        return dw
    def wg_lib():
        return torch.ops.aten.convolution_backward(go, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False))[1]
    def dg_lib():
        return torch.ops.aten.convolution_backward(go, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (True, False, False))[0]
    dw_ref = torch.ops.aten.convolution_backward(go.float(), x.float(), w.float(), None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False))[1]
    werr = (wg_ours().permute(0, 3, 1, 2) - dw_ref).abs().max().item() / dw_ref.abs().max().item()
    print(f'     wgrad rel err {werr:.1e} | ours {timeit(wg_ours):7.1f} us | MIOpen wgrad {timeit(wg_lib):7.1f} us | MIOpen dgrad {timeit(dg_lib):7.1f} us', flush=True)
    print(f'3x3 {Ci:4d}->{Co:3d} @{H}x{W} N{N}: rel err {err:.1e} | ours {t_ours:7.1f} us ({fl / t_ours / 1e6:6.0f} TF/s) | MIOpen {t_lib:7.1f} us ({fl / t_lib / 1e6:6.0f} TF/s)', flush=True)
