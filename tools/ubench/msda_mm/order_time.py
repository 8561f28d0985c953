"""Round 6: the cross-attention's query ORDER (kernels.msda_ref_order: Morton code of the level-0 cell of the reference point at quarter-cell
resolution) against coarser / finer sort keys, model geometry, forward + d_raw of the MFMA kernels.   python tools/ubench/msda_mm/order_time.py"""
import sys, torch
sys.path.insert(0, '.')
import os as _os
from gedepth_amd import hip as _hip
if _os.environ.get('GE_LIB'):                                  # A/B a differently built library
    _hip.LIB_PATH = _os.path.abspath(_os.environ['GE_LIB'])
from gedepth_amd import kernels as K
from gedepth_amd.depth.utils.position_encoding import SinePositionalEncoding
from gedepth_amd.mmrt.bricks import msda_offset_bias
dev = 'cuda'
KS = ((88, 280), (44, 140), (22, 70), (11, 35))
B, nq = 8, 176 * 560
nv = sum(h * w for h, w in KS)
g = torch.Generator().manual_seed(1)
torch.manual_seed(1234)
pe = SinePositionalEncoding(num_feats=256, normalize=False)
pos = pe.grid(176, 560, 'cpu')
lin = torch.nn.Linear(512, 2); torch.nn.init.xavier_uniform_(lin.weight); torch.nn.init.constant_(lin.bias, 0.)
ref0 = torch.sigmoid(lin(pos.flatten(2)[0].t())).detach().to(dev)
value = torch.randn(B, nv, 8, 64, generator=g).bfloat16().to(dev).requires_grad_(True)
raw0 = torch.cat((msda_offset_bias(8, 4, 8)[None, None].expand(B, nq, 512) + 0.05 * torch.randn(B, nq, 512, generator=g), 0.1 * torch.randn(B, nq, 256, generator=g)), -1).bfloat16().to(dev)
go = torch.randn(B, nq, 512, generator=g).bfloat16().to(dev)
refd = ref0[None, :, None, :].expand(B, nq, 4, 2)


def spread(v):
    v = (v | (v << 8)) & 0x00FF00FF
    v = (v | (v << 4)) & 0x0F0F0F0F
    v = (v | (v << 2)) & 0x33333333
    return (v | (v << 1)) & 0x55555555


def morton(res):
    h, w = KS[0]
    x = (ref0[:, 0].float() * (res * w)).clamp_(0, res * w - 1).to(torch.int64)
    y = (ref0[:, 1].float() * (res * h)).clamp_(0, res * h - 1).to(torch.int64)
    return torch.argsort((spread(y) << 1) | spread(x), stable=True).to(torch.int32)


def raster_blocks(bh, bw, res=4):
    """cells grouped into bh x bw blocks (raster over blocks), Morton at `res` inside a block"""
    h, w = KS[0]
    x = (ref0[:, 0].float() * (res * w)).clamp_(0, res * w - 1).to(torch.int64)
    y = (ref0[:, 1].float() * (res * h)).clamp_(0, res * h - 1).to(torch.int64)
    by, bx = y // (res * bh), x // (res * bw)
    key = ((by * 4096 + bx) << 24) | (spread(y % (res * bh)) << 1) | spread(x % (res * bw))
    return torch.argsort(key, stable=True).to(torch.int32)


import os
os.environ['GE_MSDA_VALUE'] = 'records'
cands = {'morton 1/4 cell (current)': morton(4), 'morton 1/2 cell': morton(2), 'morton cell': morton(1), 'morton 1/8 cell': morton(8),
         'blocks 2x2 cells': raster_blocks(2, 2), 'blocks 1x4 cells': raster_blocks(1, 4), 'blocks 4x1 cells': raster_blocks(4, 1), 'natural': None}
for tag, order in cands.items():
    raw = raw0.clone().requires_grad_(True)

    def run():
        value.grad = None
        K.ms_deform_attn_mm(value, raw, refd, KS, order).backward(go)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    K.PROFILER.enable()
    for _ in range(4):
        run()
    K.PROFILER.disable()
    rows = {r['name'].split('[')[0]: r['avg_us'] for r in K.PROFILER.summary()}
    print(f'{tag:28s} fwd {rows.get("msda_mm_fwd_k", 0):8.1f} us   d_raw {rows.get("msda_mm_bwd_lw_k", 0):8.1f} us')
    if order is None:
        break
