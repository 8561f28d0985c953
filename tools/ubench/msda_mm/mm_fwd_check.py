"""Round-4 bring-up: ge_msda_fwd_mm against the CPU oracle on bf16-rounded inputs + timing at the bench shapes."""
import sys, time, torch
sys.path.insert(0, '.')
from gedepth_amd import kernels as K
from oracle import gedepth_oracle as O
dev = 'cuda'

def refs_grid(qshapes):
    r = []
    for h, w in qshapes:
        gy, gx = torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing='ij')
        r.append(torch.stack((gx.reshape(-1), gy.reshape(-1)), -1))
    return torch.cat(r, 0)

def oracle(value, raw, ref, shapes, nH=8, L=4, P=8):
    B, nq = raw.shape[:2]
    n_off = nH * L * P * 2
    off = raw[..., :n_off].float().view(B, nq, nH, L, P, 2)
    norm = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32).view(1, 1, 1, L, 1, 2)
    loc = ref[:, :, None, :, None, :] + off / norm
    aw = raw[..., n_off:].float().view(B, nq, nH, L * P).softmax(-1).view(B, nq, nH, L, P)
    return O.msda_core(value.float(), shapes, loc, aw), loc, aw

def case(name, shapes, qshapes, B=2, jitter=2.5, ref_mode='grid', order_mode='tile', seed=0):
    g = torch.Generator().manual_seed(seed)
    nv = sum(h * w for h, w in shapes); nq = sum(h * w for h, w in qshapes)
    value = torch.randn(B, nv, 8, 64, generator=g).bfloat16()
    raw = torch.cat((torch.randn(B, nq, 512, generator=g) * jitter, torch.randn(B, nq, 256, generator=g)), -1).bfloat16()
    if ref_mode == 'grid':
        ref = refs_grid(qshapes)
    else:
        ref = torch.rand(nq, 2, generator=g) * 1.2 - 0.1
    ref = ref[None, :, None, :].expand(B, nq, 4, 2).contiguous()
    want, loc_w, aw_w = oracle(value, raw, ref, shapes)
    order = None
    if order_mode == 'tile':
        order = K.msda_tile_order(qshapes, dev)
    elif order_mode == 'ref':
        order = K.msda_ref_order(ref[0, :, 0].to(dev), shapes[0])
    elif order_mode == 'randperm':
        order = torch.randperm(nq, generator=g).to(torch.int32).to(dev)
    out, loc, aw = K.msda_fwd_mm(value.to(dev), raw.to(dev), ref.to(dev), shapes, order, want_loc=True)
    torch.cuda.synchronize()
    err = (out.float().cpu() - want).abs().max().item(); sc = want.abs().max().item()
    el = (loc.cpu() - loc_w).abs().max().item(); ea = (aw.cpu() - aw_w).abs().max().item()
    print(f'{name:28s} nq {nq:6d} out err {err:.3e} (scale {sc:.2f}, rel {err/sc:.2e})  loc {el:.2e} attw {ea:.2e}', flush=True)
    return err / sc

S = ((44, 70), (22, 35), (11, 18), (6, 9))
worst = 0
worst = max(worst, case('self tile-order', S, S))
worst = max(worst, case('self identity-order', S, S, order_mode='none'))
worst = max(worst, case('cross grid tile-order', ((22, 35), (11, 18), (6, 9), (3, 5)), ((44, 70),)))
worst = max(worst, case('cross random ref sorted', S, ((37, 41),), ref_mode='rand', order_mode='ref'))
worst = max(worst, case('random ref, random order', S, ((37, 41),), ref_mode='rand', order_mode='randperm', jitter=6.0))
worst = max(worst, case('ragged', ((37, 53), (19, 27), (10, 14), (5, 7)), ((21, 45), (3, 5)), jitter=4.0))
worst = max(worst, case('tiny', ((3, 5), (2, 3), (1, 2), (1, 1)), ((2, 3),), jitter=1.0))
print('worst rel', worst)
assert worst < 1.5e-2

# ---- timing at the bench shapes
def bench(name, shapes, qshapes, B=8, mode='init', sort=True):
    nv = sum(h * w for h, w in shapes); nq = sum(h * w for h, w in qshapes)
    g = torch.Generator().manual_seed(1)
    value = torch.randn(B, nv, 8, 64, generator=g).bfloat16().to(dev)
    from gedepth_amd.mmrt.bricks import msda_offset_bias
    bias = msda_offset_bias(8, 4, 8)
    raw = torch.cat((bias[None, None].expand(B, nq, 512) + 0.05 * torch.randn(B, nq, 512, generator=g), 0.1 * torch.randn(B, nq, 256, generator=g)), -1).bfloat16().to(dev)
    if len(qshapes) == 1:   # cross: sigmoid(Linear(sine)) reference points of the bench model
        from gedepth_amd.depth.utils.position_encoding import SinePositionalEncoding
        torch.manual_seed(1234)
        pe = SinePositionalEncoding(num_feats=256, normalize=True)
        pos = pe.grid(qshapes[0][0], qshapes[0][1], 'cpu')
        lin = torch.nn.Linear(512, 2); torch.nn.init.xavier_uniform_(lin.weight); torch.nn.init.constant_(lin.bias, 0.)
        ref = torch.sigmoid(lin(pos.flatten(2)[0].t())).detach()
    else:
        ref = refs_grid(qshapes)
    refd = ref.to(dev)[None, :, None, :].expand(B, nq, 4, 2)
    t0 = time.time()
    order = K.msda_ref_order(refd[0, :, 0], shapes[0]) if len(qshapes) == 1 else K.msda_tile_order(qshapes, dev)
    torch.cuda.synchronize()
    if not sort:
        order = None
    def run_mm(want):
        return K.msda_fwd_mm(value, raw, refd, shapes, order, want_loc=want)
    def run_old():
        return K.ms_deform_attn_raw(value, raw, refd, shapes, list(qshapes), 8, 4, 8)
    for fn, tag in ((lambda: run_mm(False), 'mm'), (lambda: run_mm(True), 'mm+loc'), (run_old, 'win raw (r3)')):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        print(f'{name:8s} {tag:14s} {s.elapsed_time(e) / 10:.3f} ms', flush=True)
    a = run_mm(False).float(); b = run_old().float()
    print('   mm vs r3 kernel: max abs', (a - b).abs().max().item(), 'scale', b.abs().max().item())

KS = ((88, 280), (44, 140), (22, 70), (11, 35))
bench('cross', KS, ((176, 560),))
bench('cross-unsorted', KS, ((176, 560),), sort=False)
bench('self', KS, KS)
bench('self-raster', KS, KS, sort=False)
