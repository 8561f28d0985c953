"""Round 6: the SELF-attention's sampling kernels (8 x 32 725 grid queries) — the LDS-window gather kernels (current) against the MFMA decomposition on
all queries / on the level-0 queries only (75 % of them; the coarse-level queries' 4 x 8 patches span 16 - 64 level-0 cells: windows of many chunks)
with several query-tile shapes.  Prints HIP-event times per kernel.   python tools/ubench/msda_mm/self_split_time.py"""
import os, sys, torch
sys.path.insert(0, '.')
import os as _os
from gedepth_amd import hip as _hip
if _os.environ.get('GE_LIB'):                                  # A/B a differently built library
    _hip.LIB_PATH = _os.path.abspath(_os.environ['GE_LIB'])
from gedepth_amd import kernels as K
from gedepth_amd.mmrt.bricks import msda_offset_bias
dev = 'cuda'
KS = ((88, 280), (44, 140), (22, 70), (11, 35))
B = 8
nv = sum(h * w for h, w in KS)
g = torch.Generator().manual_seed(1)
noise = float(os.environ.get('VS_NOISE', 0.05))


def refs(shapes):
    return torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing='ij')[::-1], -1).reshape(-1, 2) for h, w in shapes])


value = torch.randn(B, nv, 8, 64, generator=g).bfloat16().to(dev)
raw_all = torch.cat((msda_offset_bias(8, 4, 8)[None, None].expand(B, nv, 512) + noise * torch.randn(B, nv, 512, generator=g), 0.1 * torch.randn(B, nv, 256, generator=g)), -1).bfloat16().to(dev)
go_all = torch.randn(B, nv, 512, generator=g).bfloat16().to(dev)
ref_all = refs(KS).to(dev)


def bench(tag, fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    K.PROFILER.enable()
    for _ in range(n):
        fn()
    K.PROFILER.disable()
    rows = K.PROFILER.summary()
    print(tag)
    for r in rows:
        print(f'   {r["name"]:52s} {r["avg_us"]:9.1f} us')
    return sum(r['avg_us'] for r in rows if 'value' not in r['name'])


def run(kind, q0, q1, qshapes, order=None):
    v = value.clone().requires_grad_(True)
    raw = raw_all[:, q0:q1].contiguous().requires_grad_(True)
    ref = ref_all[q0:q1][None, :, None, :].expand(B, q1 - q0, 4, 2)
    go = go_all[:, q0:q1].contiguous()

    def f():
        v.grad = None
        raw.grad = None
        if kind == 'win':
            o = K.ms_deform_attn_raw(v, raw, ref, KS, qshapes, 8, 4, 8)
        else:
            o = K.ms_deform_attn_mm(v, raw, ref, KS, order, 8, 4, 8)
        o.backward(go)
    return f


os.environ['GE_MSDA_VALUE'] = 'records'
n0 = KS[0][0] * KS[0][1]
t = {}
t['win all'] = bench('window kernels, all 32 725 queries', run('win', 0, nv, KS))
t['mm all'] = bench('MFMA kernels, all queries, 4 x 8 tiles', run('mm', 0, nv, KS, K.msda_tile_order(KS, dev)))
t['win coarse'] = bench('window kernels, the 8 085 coarse-level queries', run('win', n0, nv, KS[1:]))
t['win fine'] = bench('window kernels, the 24 640 level-0 queries', run('win', 0, n0, KS[:1]))
for th, tw in ((4, 8), (2, 16), (8, 4), (1, 32), (16, 2)):
    t[f'mm fine {th}x{tw}'] = bench(f'MFMA kernels, level-0 queries, {th} x {tw} tiles', run('mm', 0, n0, KS[:1], K.msda_tile_order(KS[:1], dev, th, tw)))
for th, tw in ((2, 16), (1, 32)):
    t[f'mm coarse {th}x{tw}'] = bench(f'MFMA kernels, coarse queries, {th} x {tw} tiles', run('mm', n0, nv, KS[1:], K.msda_tile_order(KS[1:], dev, th, tw)))
n1 = n0 + KS[1][0] * KS[1][1]
for th, tw in ((4, 8), (2, 16), (2, 8)):
    if th * tw == 32:
        o = torch.cat((K.msda_tile_order(KS[:1], dev), K.msda_tile_order(KS[1:2], dev, th, tw) + n0))
    else:      # 16-query patches of level 1 paired into 32-query tiles
        o = torch.cat((K.msda_tile_order(KS[:1], dev), K.msda_tile_order(KS[1:2], dev, th, tw) + n0))
    t[f'mm L0+L1 {th}x{tw}'] = bench(f'MFMA kernels, level-0 + level-1 queries (L1 patches {th} x {tw})', run('mm', 0, n1, KS[:2], o))
    t[f'mm L1 only {th}x{tw}'] = bench(f'MFMA kernels, level-1 queries only ({th} x {tw})', run('mm', n0, n1, KS[1:2], K.msda_tile_order(KS[1:2], dev, th, tw)))
t['win L2+L3'] = bench('window kernels, level-2 + level-3 queries', run('win', n1, nv, KS[2:]))
t['win L1'] = bench('window kernels, level-1 queries', run('win', n0, n1, KS[1:2]))
print('forward + d_raw totals (us, d_value kernels excluded):', {k: round(v) for k, v in t.items()})
