"""Round 6: d_value on the value-stationary kernel (ge_msda_bwd_value_vs) vs the record pipeline (ge_msda_bwd_value_raw_levels) at the bench
shapes, straight through the C ABI: correctness (difference of the two results) and HIP-event times.
    python tools/ubench/msda_mm/vs_time.py [model|concentrated|spread|self|random]"""
import ctypes, os, sys, torch
sys.path.insert(0, '.')
from gedepth_amd import hip
if os.environ.get('GE_LIB'):
    hip.LIB_PATH = os.path.abspath(os.environ['GE_LIB'])
from gedepth_amd import kernels as K
from gedepth_amd.depth.utils.position_encoding import SinePositionalEncoding
from gedepth_amd.mmrt.bricks import msda_offset_bias
dev = 'cuda'
KS = ((88, 280), (44, 140), (22, 70), (11, 35))
geom = sys.argv[1] if len(sys.argv) > 1 else 'model'
B = int(os.environ.get('VS_B', 8))
nv = sum(h * w for h, w in KS)
g = torch.Generator().manual_seed(1)
torch.manual_seed(1234)
if geom == 'self':
    nq = nv
    ref0 = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing='ij')[::-1], -1).reshape(-1, 2) for h, w in KS]).to(dev)
    order = K.msda_tile_order(KS, dev)
else:
    nq = 176 * 560
    if geom == 'spread':
        gy, gx = torch.meshgrid((torch.arange(176) + 0.5) / 176, (torch.arange(560) + 0.5) / 560, indexing='ij')
        ref0 = torch.stack((gx.reshape(-1), gy.reshape(-1)), -1).to(dev)
    elif geom == 'random':
        ref0 = torch.rand(nq, 2, generator=g).to(dev)
    else:
        pe = SinePositionalEncoding(num_feats=256, normalize=(geom == 'concentrated'))
        pos = pe.grid(176, 560, 'cpu')
        lin = torch.nn.Linear(512, 2); torch.nn.init.xavier_uniform_(lin.weight); torch.nn.init.constant_(lin.bias, 0.)
        ref0 = torch.sigmoid(lin(pos.flatten(2)[0].t())).detach().to(dev)
    order = K.msda_ref_order(ref0, KS[0])
    if geom == 'random':
        order = torch.randperm(nq, generator=g).to(torch.int32).to(dev)          # no locality at all: every tile is a stray
noise = float(os.environ.get('VS_NOISE', 0.05))
value = torch.randn(B, nv, 8, 64, generator=g).bfloat16().to(dev)
raw = torch.cat((msda_offset_bias(8, 4, 8)[None, None].expand(B, nq, 512) + noise * torch.randn(B, nq, 512, generator=g), 0.1 * torch.randn(B, nq, 256, generator=g)), -1).bfloat16().to(dev).contiguous()
go = torch.randn(B, nq, 512, generator=g).bfloat16().to(dev)
ref = ref0[None, :, None, :].expand(B, nq, 4, 2).contiguous()
lib = hip.lib()
arr = (ctypes.c_int * 8)(*[v for hw in KS for v in hw])
sp = ctypes.cast(arr, ctypes.c_void_p)
ld, n_off = 768, 512
base = raw.data_ptr()
d_raw = torch.empty_like(raw)
ws_bytes = int(lib.ge_msda_bwd_vs_workspace(sp, B, nv, nq, 8, 4, 8))
assert ws_bytes > 0
ws = torch.zeros(ws_bytes, device=dev, dtype=torch.uint8)
print(f'{geom}: B {B} Nq {nq} Nv {nv}; vs workspace {ws_bytes / 2**20:.1f} MiB')
hip.check(lib.ge_msda_bwd_lw_mm(value.data_ptr(), sp, base, ld, base + n_off * 2, ld, ref.data_ptr(), ref.stride(0), ref.stride(1), ref.stride(2),
                                order.data_ptr(), go.data_ptr(), d_raw.data_ptr(), ld, d_raw.data_ptr() + n_off * 2, ld, ws.data_ptr(), B, nv, nq, 8, 4, 8, 1, None), 'lw')
rec_bytes = int(lib.ge_msda_bwd_workspace(sp, B, nv, nq, 8, 4, 8))
rws = torch.empty(rec_bytes, device=dev, dtype=torch.uint8)


def records(dv):
    hip.check(lib.ge_msda_bwd_value_raw_levels(sp, base, ld, base + n_off * 2, ld, ref.data_ptr(), ref.stride(0), ref.stride(1), ref.stride(2),
                                               go.data_ptr(), dv.data_ptr(), rws.data_ptr(), rec_bytes, 15, B, nv, nq, 8, 4, 8, 1, None), 'records')


def vs(dv):
    hip.check(lib.ge_msda_bwd_value_vs(sp, base, ld, base + n_off * 2, ld, ref.data_ptr(), ref.stride(0), ref.stride(1), ref.stride(2),
                                       order.data_ptr(), go.data_ptr(), dv.data_ptr(), ws.data_ptr(), ws_bytes, B, nv, nq, 8, 4, 8, 1, None), 'vs')


def mm(dv):
    hip.check(lib.ge_msda_bwd_value_mm(sp, base, ld, base + n_off * 2, ld, ref.data_ptr(), ref.stride(0), ref.stride(1), ref.stride(2),
                                       order.data_ptr(), go.data_ptr(), dv.data_ptr(), ws.data_ptr(), ws_bytes, 15, B, nv, nq, 8, 4, 8, 1, None), 'mm')


def timed(fn, n=10):
    dv = torch.zeros(B, nv, 8, 64, device=dev)
    fn(dv)
    torch.cuda.synchronize()
    out = dv.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        dv.zero_()
        e0.record(); fn(dv); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return out, ts[len(ts) // 2]


a, ta = timed(records)
b, tb = timed(vs)
so = int(lib.ge_msda_bwd_vs_stats_offset(sp, B, nv, nq, 8, 4, 8))
st = ws[so:so + 16].view(torch.int32).cpu().tolist()
print(f'records {ta:.3f} ms | value-stationary {tb:.3f} ms | statistics: visits {st[0]} stray tiles {st[1]} items {st[2]} multi-chunk items {st[3]}')
sc = a.abs().max()
print('vs - records: max abs / scale', ((b - a).abs().max() / sc).item(), ' l2 rel', ((b - a).norm() / a.norm()).item())
if os.environ.get('VS_MM'):
    c, tc = timed(mm)
    print(f'query-stationary mm {tc:.3f} ms; mm - records l2 rel', ((c - a).norm() / a.norm()).item())
for l, (h, w) in enumerate(KS):
    s0 = sum(hh * ww for hh, ww in KS[:l])
    print(f'   level {l}: l2 rel', ((b[:, s0:s0 + h * w] - a[:, s0:s0 + h * w]).norm() / a[:, s0:s0 + h * w].norm()).item())
