"""Round 5: the cross-attention's backward at the bench shape (8 x 98 560 queries, init-like geometry), d_value on the MFMA kernel
(ge_msda_bwd_value_mm) vs the record pipeline (GE_DISABLE=msda_value_mm), same session; prints per-kernel HIP-event times and the
difference of the two d_value results."""
import sys, torch
sys.path.insert(0, '.')
import os
from gedepth_amd import hip
if os.environ.get('GE_LIB'):                                  # A/B a differently built library
    hip.LIB_PATH = os.path.abspath(os.environ['GE_LIB'])
from gedepth_amd import kernels as K
from gedepth_amd.depth.utils.position_encoding import SinePositionalEncoding
from gedepth_amd.mmrt.bricks import msda_offset_bias
dev = 'cuda'
KS = ((88, 280), (44, 140), (22, 70), (11, 35))
B, nq = 8, 176 * 560
nv = sum(h * w for h, w in KS)
g = torch.Generator().manual_seed(1)
torch.manual_seed(1234)
# geometry: 'model' (default) = the bench model at initialisation: sigmoid(Linear(sine embedding)) with the configs' un-normalised sine
# embedding (hahi.py:294-302): reference points over half of the level-0 cells, ~8 queries per cell, a 32-query tile spans ~4 x 3 cells;
# 'concentrated' = the same with normalize=True (what tools/ubench/msda_mm/dump_inputs.py fed the round-4 harness: all 98 560 reference
# points inside 22 % x 27 % of the map, ~110 queries per cell — NOT the model's geometry); 'spread' = reference point = own position
geom = sys.argv[1] if len(sys.argv) > 1 else 'model'
pe = SinePositionalEncoding(num_feats=256, normalize=(geom == 'concentrated'))
pos = pe.grid(176, 560, 'cpu')
lin = torch.nn.Linear(512, 2); torch.nn.init.xavier_uniform_(lin.weight); torch.nn.init.constant_(lin.bias, 0.)
spread = geom == 'spread'
if spread:
    gy, gx = torch.meshgrid((torch.arange(176) + 0.5) / 176, (torch.arange(560) + 0.5) / 560, indexing='ij')
    ref0 = torch.stack((gx.reshape(-1), gy.reshape(-1)), -1).to(dev)
else:
    ref0 = torch.sigmoid(lin(pos.flatten(2)[0].t())).detach().to(dev)
value = torch.randn(B, nv, 8, 64, generator=g).bfloat16().to(dev).requires_grad_(True)
raw0 = torch.cat((msda_offset_bias(8, 4, 8)[None, None].expand(B, nq, 512) + 0.05 * torch.randn(B, nq, 512, generator=g), 0.1 * torch.randn(B, nq, 256, generator=g)), -1).bfloat16().to(dev)
go = torch.randn(B, nq, 512, generator=g).bfloat16().to(dev)
order = K.msda_ref_order(ref0, KS[0])
refd = ref0[None, :, None, :].expand(B, nq, 4, 2)
res = {}
for tag, dis in (('mfma d_value', set()), ('record pipeline', {'msda_value_mm'}), ('per level', {'auto'}))[:1 if os.environ.get('GE_LIB') else 3]:
    K.DISABLED.clear()
    K._MM_VALUE_CHOICE.clear()
    os.environ.pop('GE_MSDA_VALUE', None)
    if 'auto' not in dis:
        os.environ['GE_MSDA_VALUE'] = 'records' if dis else 'mm'
    raw = raw0.clone().requires_grad_(True)
    def run():
        value.grad = None
        o = K.ms_deform_attn_mm(value, raw, refd, KS, order); o.backward(go)
    for _ in range(3): run()
    torch.cuda.synchronize()
    K.PROFILER.enable()
    for _ in range(5): run()
    K.PROFILER.disable()
    print(tag, f'({geom} reference points)')
    for r in K.PROFILER.summary() + K.PROFILER.msda_bwd_stages():
        print(f'   {r["name"]:44s} {r["avg_us"]:9.1f} us')
    res[tag] = value.grad.float().clone()
if os.environ.get('GE_LIB'):
    sys.exit(0)
a, b = res['mfma d_value'], res['record pipeline']
c = res['per level']
print('per-level choice vs records: max abs diff / scale', ((c - b).abs().max() / b.abs().max()).item(), ' l2 rel', ((c - b).norm() / b.norm()).item())
print('d_value: max abs diff / scale', ((a - b).abs().max() / b.abs().max()).item(), ' l2 rel', ((a - b).norm() / b.norm()).item())
# run statistics of the MFMA d_value path: read the workspace back
import ctypes
lib = hip.lib()
ld, n_off = 768, 512
ws_bytes = int(lib.ge_msda_bwd_mm_workspace(B, nq, 8, 4))
ws = torch.zeros(ws_bytes, device=dev, dtype=torch.uint8)
arr = (ctypes.c_int * 8)(*[v for hw in KS for v in hw])
shapes_p = ctypes.cast(arr, ctypes.c_void_p)
raw = raw0.contiguous(); d_raw = torch.empty_like(raw); ref = refd.contiguous(); dv = torch.zeros(B, nv, 8, 64, device=dev)
base, dbase = raw.data_ptr(), d_raw.data_ptr()
hip.check(lib.ge_msda_bwd_lw_mm(value.data_ptr(), shapes_p, base, ld, base + n_off * 2, ld, ref.data_ptr(), ref.stride(0), ref.stride(1), ref.stride(2),
                                order.data_ptr(), go.data_ptr(), dbase, ld, dbase + n_off * 2, ld, ws.data_ptr(), B, nv, nq, 8, 4, 8, 1, None), 'lw')
hip.check(lib.ge_msda_bwd_value_mm(shapes_p, base, ld, base + n_off * 2, ld, ref.data_ptr(), ref.stride(0), ref.stride(1), ref.stride(2),
                                   order.data_ptr(), go.data_ptr(), dv.data_ptr(), ws.data_ptr(), ws_bytes, 15, B, nv, nq, 8, 4, 8, 1, None), 'dv')
torch.cuda.synchronize()
ntiles, segs = (nq + 31) // 32, B * 8 * 4
r256 = lambda x: (x + 255) // 256 * 256
cap = (segs + 7) // 8 * ntiles
o_runs = r256(segs * ntiles * 16); o_ctrl = o_runs + r256(8 * cap * 32)
ctrl = ws[o_ctrl:o_ctrl + 128].view(torch.int32).cpu()
runs = ws[o_runs:o_runs + 8 * cap * 32].view(torch.int32).view(8, cap, 8).cpu()
tot = dict(runs=0, rows=0, tiles=0)
per_level = [[0, 0, 0] for _ in range(4)]
for x in range(8):
    n = int(ctrl[x]); r = runs[x, :n]
    rows = (r[:, 3] & 0xffff) * (r[:, 3] >> 16) + (r[:, 5] & 0xffff) * (r[:, 5] >> 16); tiles = (r[:, 1] >> 24) & 0xff
    tot['runs'] += n; tot['rows'] += int(rows.sum()); tot['tiles'] += int(tiles.sum())
    for l in range(4):
        m = (r[:, 0] & 3) == l
        per_level[l][0] += int(m.sum()); per_level[l][1] += int(rows[m].sum()); per_level[l][2] += int(tiles[m].sum())
print('runs', tot, 'per level (runs, rows, tiles):', per_level, ' kernel statistics: rows', int(ctrl[16]), 'tile passes', int(ctrl[17]))
# the adaptive choice (kernels._MMValueChoice): 40 backward calls, what it settles on
K.DISABLED.clear()
K._MM_VALUE_CHOICE.clear()
os.environ.pop('GE_MSDA_VALUE', None)
raw = raw0.clone().requires_grad_(True)
for i in range(40):
    value.grad = None
    K.ms_deform_attn_mm(value, raw, refd, KS, order).backward(go)
    if i % 8 == 0: torch.cuda.synchronize()
torch.cuda.synchronize()
for k, c in K._MM_VALUE_CHOICE.items():
    print('choice for', k[:4], ': MFMA kernel on levels', [l for l in range(4) if (c.mm_mask >> l) & 1], 'statistics', c.last)
