"""Round-4 bring-up: ms_deform_attn_mm forward + backward against the CPU oracle (autograd) on bf16-rounded inputs, and timing."""
import sys, torch
sys.path.insert(0, '.')
from gedepth_amd import kernels as K
from oracle import gedepth_oracle as O
dev = 'cuda'

def refs_grid(qshapes):
    r = []
    for h, w in qshapes:
        gy, gx = torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing='ij')
        # off the k/64 grid of bf16 offsets: an exact-pixel sample sits ON the kink of the bilinear gradient, where the last bit decides
        r.append(torch.stack((gx.reshape(-1) + 0.013 / w, gy.reshape(-1) + 0.017 / h), -1))
    return torch.cat(r, 0)

def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()

def case(name, shapes, qshapes, B=2, jitter=2.5, ref_mode='grid', order_mode='tile', seed=0):
    g = torch.Generator().manual_seed(seed)
    nH, L, P = 8, 4, 8
    nv = sum(h * w for h, w in shapes); nq = sum(h * w for h, w in qshapes)
    value = torch.randn(B, nv, 8, 64, generator=g).bfloat16()
    raw = torch.cat((torch.randn(B, nq, 512, generator=g) * jitter, torch.randn(B, nq, 256, generator=g)), -1).bfloat16()
    ref = refs_grid(qshapes) if ref_mode == 'grid' else torch.rand(nq, 2, generator=g) * 1.2 - 0.1
    ref = ref[None, :, None, :].expand(B, nq, 4, 2).contiguous()
    go = torch.randn(B, nq, 512, generator=g).bfloat16()
    vc, rc, fc = value.float().requires_grad_(True), raw.float().requires_grad_(True), ref.clone().requires_grad_(True)
    n_off = 512
    off = rc[..., :n_off].view(B, nq, nH, L, P, 2)
    norm = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32).view(1, 1, 1, L, 1, 2)
    loc = fc[:, :, None, :, None, :] + off / norm
    aw = rc[..., n_off:].view(B, nq, nH, L * P).softmax(-1).view(B, nq, nH, L, P)
    o = O.msda_core(vc, shapes, loc, aw)
    o.backward(go.float())
    order = None
    if order_mode == 'tile':
        order = K.msda_tile_order(qshapes, dev)
    elif order_mode == 'ref':
        order = K.msda_ref_order(ref[0, :, 0].to(dev), shapes[0])
    elif order_mode == 'randperm':
        order = torch.randperm(nq, generator=g).to(torch.int32).to(dev)
    v, r, f = value.to(dev).requires_grad_(True), raw.to(dev).requires_grad_(True), ref.to(dev).requires_grad_(True)
    out = K.ms_deform_attn_mm(v, r, f, shapes, order)
    out.backward(go.to(dev))
    torch.cuda.synchronize()
    e = dict(out=rel(out.float().cpu(), o), dv=rel(v.grad.float().cpu(), vc.grad), doff=rel(r.grad[..., :n_off].float().cpu(), rc.grad[..., :n_off]),
             dlog=rel(r.grad[..., n_off:].float().cpu(), rc.grad[..., n_off:]), dref=rel(f.grad.cpu(), fc.grad))
    print(f'{name:28s} nq {nq:6d} ' + ' '.join(f'{k} {x:.2e}' for k, x in e.items()), flush=True)
    return e

S = ((44, 70), (22, 35), (11, 18), (6, 9))
res = []
res.append(case('self tile-order', S, S))
res.append(case('self identity-order', S, S, order_mode='none'))
res.append(case('cross grid tile-order', ((22, 35), (11, 18), (6, 9), (3, 5)), ((44, 70),)))
res.append(case('cross random ref sorted', S, ((37, 41),), ref_mode='rand', order_mode='ref'))
res.append(case('random ref, random order', S, ((37, 41),), ref_mode='rand', order_mode='randperm', jitter=6.0))
res.append(case('ragged', ((37, 53), (19, 27), (10, 14), (5, 7)), ((21, 45), (3, 5)), jitter=4.0))
res.append(case('tiny', ((3, 5), (2, 3), (1, 2), (1, 1)), ((2, 3),), jitter=1.0))
for k, tol in (('out', 1.5e-2), ('dv', 1.5e-2), ('doff', 1.5e-2), ('dlog', 1.5e-2), ('dref', 1.5e-2)):
    w = max(r[k] for r in res)
    print('worst', k, w)
    assert w < tol, (k, w)
print('PARITY OK')

# timing at the bench shapes: mm path vs the round-3 raw path (forward + backward)
import time
def bench(name, shapes, qshapes, B=8):
    nv = sum(h * w for h, w in shapes); nq = sum(h * w for h, w in qshapes)
    g = torch.Generator().manual_seed(1)
    value = torch.randn(B, nv, 8, 64, generator=g).bfloat16().to(dev).requires_grad_(True)
    from gedepth_amd.mmrt.bricks import msda_offset_bias
    bias = msda_offset_bias(8, 4, 8)
    raw = torch.cat((bias[None, None].expand(B, nq, 512) + 0.05 * torch.randn(B, nq, 512, generator=g), 0.1 * torch.randn(B, nq, 256, generator=g)), -1).bfloat16().to(dev).requires_grad_(True)
    if len(qshapes) == 1:
        import numpy as np
        ref = torch.from_numpy(np.fromfile('tools/ubench/msda_mm/data/ref_cross.bin', dtype=np.float32).reshape(-1, 2))
    else:
        ref = refs_grid(qshapes)
    refd = ref.to(dev)[None, :, None, :].expand(B, nq, 4, 2)
    order = K.msda_ref_order(refd[0, :, 0], shapes[0]) if len(qshapes) == 1 else K.msda_tile_order(qshapes, dev)
    go = torch.randn(B, nq, 512, generator=g).bfloat16().to(dev)
    def run_mm():
        o = K.ms_deform_attn_mm(value, raw, refd, shapes, order); o.backward(go)
    def run_old():
        o = K.ms_deform_attn_raw(value, raw, refd, shapes, list(qshapes), 8, 4, 8); o.backward(go)
    for fn, tag in ((run_mm, 'mm fwd+bwd'), (run_old, 'r3 fwd+bwd')):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        K.PROFILER.enable()
        for _ in range(5): fn()
        K.PROFILER.disable()
        for r in K.PROFILER.summary() + K.PROFILER.msda_bwd_stages():
            print(f'   {name} {tag}: {r["name"]:44s} {r["avg_us"]:9.1f} us')
KS = ((88, 280), (44, 140), (22, 70), (11, 35))
bench('cross', KS, ((176, 560),))
bench('self', KS, KS)
