"""Probe: does the d_value drain speed up when the cross-attention's token tensors (raw, reference points, d_out) are STORED in the
sorted query order (so that the gradient rows a value tile's records gather are contiguous), instead of being reached through `order`?"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from gedepth_amd import kernels as K
from gedepth_amd.mmrt.bricks import msda_offset_bias
dev = 'cuda'
KS = ((88, 280), (44, 140), (22, 70), (11, 35))
B, nq = 8, 176 * 560
nv = sum(h * w for h, w in KS)
g = torch.Generator().manual_seed(1)
value = torch.randn(B, nv, 8, 64, generator=g).bfloat16().to(dev).requires_grad_(True)
raw0 = torch.cat((msda_offset_bias(8, 4, 8)[None, None].expand(B, nq, 512) + 0.05 * torch.randn(B, nq, 512, generator=g), 0.1 * torch.randn(B, nq, 256, generator=g)), -1).bfloat16().to(dev)
ref0 = torch.from_numpy(np.fromfile('tools/ubench/msda_mm/data/ref_cross.bin', dtype=np.float32).reshape(-1, 2)).to(dev)
go0 = torch.randn(B, nq, 512, generator=g).bfloat16().to(dev)
order = K.msda_ref_order(ref0, KS[0])
idx = order.long()
for tag, raw, ref, go, od in (('through order[]', raw0, ref0, go0, order), ('stored sorted', raw0[:, idx].contiguous(), ref0[idx].contiguous(), go0[:, idx].contiguous(), None),
                              ('raster, no order', raw0, ref0, go0, None)):
    raw = raw.clone().requires_grad_(True)
    refd = ref[None, :, None, :].expand(B, nq, 4, 2)
    def run():
        o = K.ms_deform_attn_mm(value, raw, refd, KS, od); o.backward(go)
    for _ in range(3): run()
    torch.cuda.synchronize()
    K.PROFILER.enable()
    for _ in range(5): run()
    K.PROFILER.disable()
    print(tag)
    for r in K.PROFILER.summary() + K.PROFILER.msda_bwd_stages():
        print(f'   {r["name"]:44s} {r["avg_us"]:9.1f} us')
