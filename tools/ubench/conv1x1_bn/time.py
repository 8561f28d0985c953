"""conv_proj at the bench shape (8 x 64 -> 512 x 176 x 560, bf16): the fused 1x1 conv + BatchNorm + ReLU + position add (csrc/conv1x1_bn.hip)
against the two-pass composition (library conv, ge_bn_act_nhwc_*, ge_add_rows, autograd's gradient add), forward + backward, HIP-event times
per kernel (kernels.PROFILER) and wall time per iteration.  GE_LIB=<path> times a differently built library (variants.sh)."""
import os, sys, time, torch
sys.path.insert(0, '.')
from gedepth_amd import hip
if os.environ.get('GE_LIB'):
    hip.LIB_PATH = os.path.abspath(os.environ['GE_LIB'])
from gedepth_amd import kernels as K
from gedepth_amd.mmrt.bricks import ConvModule
from gedepth_amd.depth.utils.position_encoding import SinePositionalEncoding
dev = 'cuda'
torch.manual_seed(0)
B, H, W, E = 8, 176, 560, 512
block = ConvModule(64, E, 1, norm_cfg=dict(type='BN', requires_grad=True), act_cfg=dict(type='ReLU')).to(dev).train()
pos = SinePositionalEncoding(num_feats=E // 2, normalize=False).grid(H, W, dev)
x = torch.relu(torch.randn(B, 64, H, W, device=dev)).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
gq = torch.randn(B, H * W, E, device=dev).bfloat16()
gw = torch.randn(B, E + 64, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)     # gradient of the concat the map feeds


def step(fused):
    x.grad = None
    with torch.autocast('cuda', dtype=torch.bfloat16):
        if fused:
            y, q = K.conv1x1_bn_act_pos(block, x, pos)
        else:
            y = block(x)
            q = K.tokens_from_map(y, pos)
        wide = torch.cat([y, x], 1)
    torch.autograd.backward([wide, q], [gw, gq])


for tag, fused in (('fused', True), ('two-pass', False))[:1 if os.environ.get('GE_LIB') else 2]:
    for _ in range(3):
        step(fused)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step(fused)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 10 * 1e3
    K.PROFILER.enable()
    for _ in range(5):
        step(fused)
    K.PROFILER.disable()
    print(f'{tag}: {wall:.3f} ms per forward + backward (incl. the concat and its backward)')
    for r in K.PROFILER.summary():
        print(f'   {r["name"]:52s} {r["avg_us"]:9.1f} us  {r.get("GBps", 0):7.0f} GB/s')
