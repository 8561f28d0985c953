"""End-to-end parity of the product model (HIP kernels + library GEMM/conv) on MI355X against
(a) fixtures produced by the reference itself and (b) the CPU oracle at BASELINE.json's full 352x1120 size."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import gedepth_oracle as O
from oracle.fill import fill_state_dict, load_filled

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(cfg_name, **backbone_over):
    from gedepth_amd.depth.models import build_depther
    from gedepth_amd.mmrt.config import Config
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', cfg_name))
    cfg.model.pretrained = None
    cfg.model.backbone.drop_path_rate = 0.0
    for k, v in backbone_over.items():
        cfg.model.backbone[k] = v
    m = build_depther(cfg.model)
    m.neck.multi_att.dropout.p = 0.0
    m.neck.self_attn.dropout.p = 0.0
    return m


def T(a):
    return torch.from_numpy(np.asarray(a))


def set_exact(model):
    for mod in model.modules():
        if hasattr(mod, 'kernel_variant'):
            mod.kernel_variant = 1          # exact-fp32 window attention


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    return torch.device('cuda:0')


PARITY_LOG = os.path.join(ROOT, 'gpurun_out', 'parity_e2e.json')


def _log_parity(tag, record):
    try:
        os.makedirs(os.path.dirname(PARITY_LOG), exist_ok=True)
        data = json.load(open(PARITY_LOG)) if os.path.isfile(PARITY_LOG) else {}
        data[tag] = record
        json.dump(data, open(PARITY_LOG, 'w'), indent=1)
    except OSError:
        pass


import contextlib


@contextlib.contextmanager
def _capture_kink_decisions(model):
    """Record which side the HIP forward takes at every gradient kink, in the layout oracle.KINK_FORCE expects: the pass
    mask of every ConvModule activation / the stem / conv_depth's ReLU (``output > 0``; ReLU and LeakyReLU keep the sign),
    and the integer sampling cell of every deformable-attention tap (from the fp32 locations the kernels consume)."""
    from gedepth_amd.depth.models.necks import hahi
    from gedepth_amd.mmrt.bricks import ConvModule
    dec = dict(act={}, floors=[])
    handles = []
    for name, mod in model.named_modules():
        if isinstance(mod, ConvModule) or name == 'decode_head.conv_depth':
            handles.append(mod.register_forward_hook(lambda m, i, o, name=name: dec['act'].__setitem__(name, (o.detach() > 0).cpu())))
    bb = model.backbone
    stem = bb.conv_stem
    bb.conv_stem = lambda x: (lambda y: (dec['act'].__setitem__('backbone.bn1', (y.detach() > 0).cpu()), y)[1])(stem(x))
    orig = hahi.ms_deform_attn_raw

    def spy(value, raw, ref, shapes, query_shapes, nH, L, P):
        # the locations the fused kernel samples at: the prepare kernel evaluates the same fp32 expression (ref + off / (W, H))
        from gedepth_amd.kernels import msda_prepare
        with torch.no_grad():
            loc, _ = msda_prepare(raw.detach(), ref.detach(), shapes, nH, L, P)
        dec['floors'].append(O.sampling_cells(loc.float().cpu(), shapes))
        return orig(value, raw, ref, shapes, query_shapes, nH, L, P)
    hahi.ms_deform_attn_raw = spy
    try:
        yield dec
    finally:
        hahi.ms_deform_attn_raw = orig
        bb.conv_stem = stem
        for h in handles:
            h.remove()


@pytest.mark.parametrize('cfg_name,tag,adaptive', [('depthformer_swint_v.py', 'T_V', False), ('depthformer_swint_a.py', 'T_A', True)])
def test_test_path_vs_reference_fixture(dev, golden, cfg_name, tag, adaptive):
    """The product's test path (``forward(return_loss=False)`` -> ``simple_test`` / ``aug_test`` -> ``inference`` with the flip-back,
    depther/encoder_decoder.py) on the HIP kernels against arrays the reference's OWN ``simple_test`` / ``aug_test`` wrote
    (tests/golden/make_golden.py -> test_path_*.npz): fp32, exact-fp32 window attention, 1e-4 relative like the eval depth."""
    g = golden(f'test_path_{tag}')
    model = build(cfg_name)
    load_filled(model, 'e2e')
    model = model.to(dev).eval()
    set_exact(model)
    img = T(g['img']).to(dev)
    meta = dict(ori_shape=(64, 96, 3), flip=False, flip_direction='horizontal')
    zero = [torch.zeros(2, 2, device=dev)] * 2
    rel = lambda a, b: (np.abs(a - b) / np.maximum(np.abs(b), 1e-3)).max()
    with torch.no_grad():
        simple = np.stack(model([img], [[meta] * 2], return_loss=False, pe_ori_point=zero[:1]))
        assert rel(simple, g['simple']) <= 1e-4, rel(simple, g['simple'])
        for direction, dim, key in (('horizontal', 3, 'aug_h'), ('vertical', 2, 'aug_v')):
            flipped = dict(meta, flip=True, flip_direction=direction)
            aug = np.stack(model([img, img.flip(dim)], [[meta] * 2, [flipped] * 2], return_loss=False, pe_ori_point=zero))
            assert aug.shape == g[key].shape
            assert rel(aug, g[key]) <= 1e-4, (direction, rel(aug, g[key]))


@pytest.mark.parametrize('cfg_name,tag,adaptive', [('depthformer_swint_v.py', 'e2e_T_V', False),
                                                   ('depthformer_swint_a.py', 'e2e_T_A', True),
                                                   ('depthformer_a.py', 'e2e_L_A', True)])
def test_e2e_vs_reference_fixture(dev, golden, cfg_name, tag, adaptive):
    """Whole model, fp32, exact-fp32 window attention, against (1) the fixture the reference itself produced on CPU and
    (2) the same algorithm evaluated in FLOAT64 (tests/f64ref.py), which arbitrates where two correct fp32
    implementations disagree.  north_star: predicted depth within 1e-4 rel."""
    import f64ref
    g = golden(tag)
    f64 = f64ref.run(tag)
    model = build(cfg_name)
    load_filled(model, 'e2e')
    model = model.to(dev)
    set_exact(model)
    img, gt, kgt = T(g['img']).to(dev), T(g['depth_gt']).to(dev), T(g['pe_k_gt']).to(dev)
    metas = [dict(flip=False, ori_shape=(64, 96, 3))] * 2
    model.eval()
    with torch.no_grad():
        depth = model.encode_decode(img, metas).cpu().double()
    ref, ref64 = T(g['depth_eval']).double(), f64['depth_eval']
    rel = lambda a, b: ((a - b).abs() / b.abs().clamp_min(1e-3)).max().item()
    rec = dict(depth_hip_vs_fixture=rel(depth, ref), depth_hip_vs_f64=rel(depth, ref64), depth_fixture_vs_f64=rel(ref, ref64))
    print(f'\n[{tag}] eval depth max rel err: HIP-fixture {rec["depth_hip_vs_fixture"]:.2e}  HIP-f64 {rec["depth_hip_vs_f64"]:.2e}  '
          f'fixture-f64 {rec["depth_fixture_vs_f64"]:.2e}')
    # north star: predicted depth within 1e-4 rel of the reference.  Against float64 the reference itself can be further
    # than that where its own 0/1 ground mask flips a pixel (T-A: 2.1e-4), so that comparison is relative to the reference's
    assert rec['depth_hip_vs_fixture'] <= 1e-4, rec
    assert rec['depth_hip_vs_f64'] <= max(1e-4, 1.5 * rec['depth_fixture_vs_f64']), rec
    model.train()
    kw = dict(pe_k_gt=kgt) if adaptive else {}
    with _capture_kink_decisions(model) as decisions:
        out = model.train_step(dict(img=img, img_metas=metas, depth_gt=gt, **kw), None)
    names = json.loads(str(g['loss_names']))
    for n, v in zip(names, g['loss_values']):
        assert abs(out['log_vars'][n] - v) <= 1e-5 * abs(v) + 1e-6, (n, out['log_vars'][n], v)
        assert abs(out['log_vars'][n] - f64['log_vars'][n]) <= 1e-5 * abs(v) + 1e-6, (n, out['log_vars'][n], f64['log_vars'][n])
    out['loss'].backward()
    params = dict(model.named_parameters())
    assert all(p.grad is not None for p in params.values())
    # Gradients of EVERY parameter.  ReLU / LeakyReLU and bilinear sampling have kinks: an element whose pre-activation (or
    # sampling coordinate) lies within fp32 rounding of the kink takes one slope or the other depending on the last bits, in
    # ANY fp32 implementation — the reference on CPU flips 2-3 of its 4.1 M activations against float64 at this fixture
    # (depending on its thread count), and ONE flipped element of the 2 x 1536 x 2 x 3 top-level map moves every stage-3
    # gradient by 4e-3 (tests/test_oracle_golden.py::test_fp32_gradient_gap_is_kink_decisions).  So the float64 oracle is
    # evaluated with the HIP run's own decisions at every kink (oracle.KINK_FORCE; only rounding-level moves are accepted:
    # flipped pre-activations < 1e-4, sampling shifts < 1e-3 px) and the HIP gradients must then agree with it to 1e-4.
    f64k = f64ref.run(tag, force=decisions)
    st = f64k['force_stats']
    print(f'[{tag}] kink decisions that differ from plain float64: {st["act_flipped"]} of {st["act_seen"]} activations '
          f'(largest flipped |pre-activation| {st["max_flipped_preact"]:.1e}), {st["floor_flipped"]} of {st["floor_seen"]} sampling '
          f'cells (largest shift {st["max_shift_px"]:.1e} px)')
    assert st['max_flipped_preact'] <= 1e-4 and st['act_flipped'] <= 1e-5 * st['act_seen'] + 2, st
    assert st['floor_flipped'] <= 1e-5 * st['floor_seen'] + 2, st
    total64 = torch.sqrt(sum((v.double() ** 2).sum() for v in f64k['grads'].values())).item()
    table = []
    for k, p in params.items():
        gk = f64ref.sample(f64k['grads'][k])
        if gk.norm().item() < 1e-10 * total64:                 # true gradient zero (stage-norm biases in front of a BatchNorm)
            assert f64ref.sample(p.grad).double().norm().item() <= 1e-6 * total64, (k, 'zero-gradient tensor above the noise floor')
            continue
        table.append((k, f64ref.l2rel(f64ref.sample(p.grad), gk), f64ref.l2rel(f64ref.sample(p.grad), f64ref.sample(f64['grads'][k]))))
    table.sort(key=lambda r: -r[1])
    print(f'[{tag}] gradient l2rel, {len(table)} tensors: worst vs float64 with the same kink decisions (| vs plain float64):')
    for k, e, e_plain in table[:8]:
        print(f'   {k:72s} {e:.2e} | {e_plain:.2e}')
    rec['grad_worst_vs_f64_same_kinks'] = table[0][1]
    rec['grad_worst_vs_f64_plain'] = max(r[2] for r in table)
    rec['kink_stats'] = st
    rec['grad_table'] = table[:40]
    _log_parity(tag, rec)
    bad = [(k, e) for k, e, _ in table if e > 1e-4]
    assert not bad, bad[:8]
    f64 = f64k
    total = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params.values())).item()
    assert abs(total - total64) <= 1e-4 * total64, (total, total64)
    # ... and DIRECTLY against the gradients the reference itself produced (`grad::*` of the fixture: the reference's fp32 CPU
    # backward, strided samples), at the bound the oracle is held to in tests/test_oracle_golden.py::test_e2e (2e-3 of the
    # tensor's max magnitude; the gap is the kink decisions discussed above, on both sides), so that no link of
    # reference -> oracle -> HIP is only transitive.
    worst_fix = ('', 0.0)
    n_fix = 0
    for k in g.files:
        if not k.startswith('grad::'):
            continue
        gr = params[k[6:]].grad.detach().float().cpu().flatten()
        gr = gr[::max(1, gr.numel() // 50000)]
        ref_g = T(g[k])
        scale = ref_g.abs().max().item() + 1e-12
        e = (gr - ref_g).abs().max().item() / scale
        n_fix += 1
        if e > worst_fix[1]:
            worst_fix = (k[6:], e)
    print(f'[{tag}] HIP gradients vs the reference fixture directly: {n_fix} tensors, worst max-abs / scale = {worst_fix[1]:.2e} ({worst_fix[0]})')
    rec['grad_worst_vs_fixture'] = worst_fix[1]
    _log_parity(tag, rec)
    # Swin-T: the oracle's own bound (2e-3) holds directly (measured 5.0e-4 / 5.7e-4).  Swin-L-A: the fixture has activations within
    # 4e-7 of a ReLU kink; the reference's CPU run and the HIP run land on different sides of different ones (each flips 1-3 of
    # 4.15 M decisions against float64, see above), and ONE flipped element moves the cancelling sums of the early backbone (q/k/v
    # biases, patch embed) by several 1e-3 of their scale: measured 7.8e-3 on stages.0.blocks.1.attn.w_msa.qkv.bias.  The bound for
    # that fixture is therefore 1.5e-2 here, and the 1e-4 agreement is asserted above against float64 under equal decisions.
    bound = 1.5e-2 if tag == 'e2e_L_A' else 2e-3
    assert n_fix > 0 and worst_fix[1] <= bound, worst_fix
    assert abs(total - float(g['grad_norm_total'])) <= 1e-3 * float(g['grad_norm_total']), (total, float(g['grad_norm_total']))


def test_full_size_forward_vs_oracle(dev):
    """BASELINE.json configs[0]: DepthFormer-SwinT + GEDepth-Vanilla, 1x352x1120 forward — the oracle runs it on
    the host CPU, the product on the MI355X, same seeded input and weights."""
    from gedepth_amd.depth.datasets.synthetic import synthetic_batch
    model = build('depthformer_swint_v.py')
    sd = load_filled(model, 'full')
    P = {k: v.clone() for k, v in sd.items()}
    model = model.to(dev).eval()
    set_exact(model)
    batch = synthetic_batch(1, 352, 1120, seed=1234)
    with torch.no_grad():
        ref = O.encode_decode(batch['img'], P, dict(O.SWIN_T, adaptive=False))
        out = model.encode_decode(batch['img'].to(dev), batch['img_metas'])
    assert out.shape == (1, 1, 352, 1120)
    rel = ((out.cpu() - ref).abs() / ref.abs().clamp_min(1e-3))
    print(f'\n[full size 1x352x1120 Swin-T-V] eval depth rel err vs fp32 oracle: max {rel.max().item():.2e} mean {rel.mean().item():.2e}')
    _log_parity('full_size_T_V', dict(depth_max_rel=rel.max().item(), depth_mean_rel=rel.mean().item()))
    # north_star: predicted depth within 1e-4 relative in fp32 (measured 2.5e-5 max, 1.0e-6 mean at this size)
    assert rel.max().item() <= 1e-4, rel.max().item()
    assert rel.mean().item() <= 5e-6, rel.mean().item()


def test_bf16_autocast_train_step(dev):
    """Perf-mode numerics: bf16 autocast (fp32 master weights, fp32 losses / ground embedding) stays close to the
    fp32 loss on the same input."""
    from gedepth_amd.depth.datasets.synthetic import synthetic_batch
    torch.manual_seed(0)
    model = build('depthformer_swint_a.py')
    load_filled(model, 'bf16')
    model = model.to(dev).train()
    batch = synthetic_batch(2, 128, 160, seed=7, device=dev, valid_fraction=0.3)
    set_exact(model)
    ref = model.train_step(batch, None)
    for mod in model.modules():
        if hasattr(mod, 'kernel_variant'):
            mod.kernel_variant = 0
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out = model.train_step(batch, None)
    out['loss'].backward()
    for k, v in ref['log_vars'].items():
        assert abs(out['log_vars'][k] - v) <= 5e-2 * abs(v) + 1e-3, (k, out['log_vars'][k], v)
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())


# ------------------------------------------------------------------ BASELINE.json configs on the HIP path
def _grad_vector(model):
    return torch.cat([p.grad.detach().flatten().float() for p in model.parameters()])


def test_config2_bf16_full_shape_step_vs_fp32(dev):
    """configs[1] at its real shape — DepthFormer-SwinT + GEDepth-Vanilla, 8x352x1120, bf16 autocast, full step — against
    the fp32 step of the same model on the same batch: loss within the bf16 bound, gradient direction preserved."""
    from gedepth_amd.depth.datasets.synthetic import synthetic_batch
    from gedepth_amd.mmrt.config import Config
    from gedepth_amd.mmrt.optim import build_optimizer
    from gedepth_amd import kernels as _k
    _k.FALLBACKS.clear()
    torch.manual_seed(0)
    model = build('depthformer_swint_v.py')
    model.init_weights()
    model = model.to(dev).train()
    batch = synthetic_batch(8, 352, 1120, seed=1234, device=dev)
    ref = model.train_step(batch, None)                     # fp32 storage + arithmetic (exact-fp32 window attention)
    ref['loss'].backward()
    g32 = _grad_vector(model)
    msda_names = [n for n, _ in model.named_parameters() if 'sampling_offsets' in n or 'attention_weights' in n]
    assert len(msda_names) == 8
    named32 = {n: p.grad.detach().double().flatten().clone() for n, p in model.named_parameters() if n in msda_names}
    for p in model.parameters():
        p.grad = None
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', 'depthformer_swint_v.py'))
    optimizer = build_optimizer(model, cfg.optimizer, cfg.optimizer_config.get('grad_clip'))
    optimizer.zero_grad()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out = model.train_step(batch, optimizer)
    out['loss'].backward()
    g16 = _grad_vector(model)
    l32, l16 = ref['log_vars']['loss'], out['log_vars']['loss']
    cos = torch.nn.functional.cosine_similarity(g16.double(), g32.double(), dim=0).item()
    nrm = (g16.norm() / g32.norm()).item()
    print(f'\n[config #2 8x352x1120] loss fp32 {l32:.6f} bf16 {l16:.6f} rel {abs(l16 - l32) / abs(l32):.2e}; '
          f'grad cosine {cos:.5f}, |g_bf16|/|g_fp32| {nrm:.4f}')
    _log_parity('config2_bf16_vs_fp32', dict(loss_fp32=l32, loss_bf16=l16, grad_cosine=cos, grad_norm_ratio=nrm))
    from gedepth_amd import kernels
    assert not kernels.FALLBACKS, f'modules fell back to ATen on the hot path: {kernels.FALLBACKS}'
    assert abs(l16 - l32) <= 1e-2 * abs(l32), (l16, l32)        # bf16 has 8 mantissa bits: 4e-3 per rounding
    assert torch.isfinite(g16).all() and cos >= 0.98 and 0.9 <= nrm <= 1.1, (cos, nrm)
    # the gradients that exist only through d_loc / d_attw of the deformable attention are a small part of the global vector:
    # check them one by one (a wrong bf16 dot product in those kernels left the global cosine at 0.99)
    worst = 1.0
    for n, p in model.named_parameters():
        if n in msda_names:
            c = torch.nn.functional.cosine_similarity(p.grad.detach().double().flatten(), named32[n], dim=0).item()
            worst = min(worst, c)
            assert c >= 0.97, (n, c)
    print(f'[config #2] worst gradient cosine over the sampling-offset / attention-weight projections: {worst:.5f}')
    optimizer.step()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p).all() for p in model.parameters())


def _learnable_batch(B, H, W, seed, dev, valid_fraction=0.3):
    """synthetic_batch with a target the network can actually learn from its input: the flat-ground depth of channel 4, modulated by
    the red channel (the stock synthetic target is noise — fine for throughput, useless for a convergence comparison)."""
    from gedepth_amd.depth.datasets.synthetic import synthetic_batch
    b = synthetic_batch(B, H, W, seed=seed, device=dev, valid_fraction=valid_fraction)
    ground = b['img'][:, 4:5].abs().clamp(1.0, 70.0)
    target = (ground * (1.0 + 0.25 * torch.tanh(b['img'][:, 0:1]))).clamp(1.0, 80.0)
    b['depth_gt'] = torch.where(b['depth_gt'] > 0, target, torch.zeros_like(target))
    return b


def test_bf16_eval_error_budget_per_module(dev):
    """Round-3 review, weak #2: bf16 eval depth sits ~2 % (mean relative) from fp32 at the bench shape on random-init weights, and
    nothing said where that comes from.  The model is evaluated at 1 x 352 x 1120 with ONE top-level module at a time under bf16
    autocast (its inputs and outputs cast at the boundary, everything else fp32): the per-module contributions are recorded in
    gpurun_out/parity_e2e.json and bounded, and the all-bf16 error is checked to be explained by them (no super-additive blow-up)."""
    from gedepth_amd.depth.datasets.synthetic import synthetic_batch
    torch.manual_seed(0)
    model = build('depthformer_swint_v.py')
    model.init_weights()
    model = model.to(dev).eval()
    batch = synthetic_batch(1, 352, 1120, seed=7, device=dev)
    parts = dict(backbone=model.backbone, neck=model.neck, pe_mask_neck=model.pe_mask_neck, decode_head=model.decode_head)

    def cast(o, dt):
        if torch.is_tensor(o):
            return o.to(dt) if o.is_floating_point() else o
        if isinstance(o, (list, tuple)):
            return type(o)(cast(v, dt) for v in o)
        return o

    def run(low):
        saved = {}
        for name, mod in parts.items():
            saved[name] = mod.forward

            def fwd(*a, _f=mod.forward, _lp=name in low, **k):
                if not _lp:
                    with torch.autocast('cuda', enabled=False):
                        return _f(*cast(a, torch.float32), **{kk: cast(v, torch.float32) for kk, v in k.items()})
                with torch.autocast('cuda', dtype=torch.bfloat16):
                    return cast(_f(*a, **k), torch.float32)
            mod.forward = fwd
        try:
            with torch.no_grad():
                return model.encode_decode(batch['img'], batch['img_metas']).float()
        finally:
            for name, mod in parts.items():
                mod.forward = saved[name]
    ref = run(())
    rel = lambda d: ((d - ref).abs() / ref.abs().clamp_min(1e-3))
    rec = {}
    for name in parts:
        r = rel(run((name,)))
        rec[name] = dict(mean_rel=r.mean().item(), max_rel=r.max().item())
    r = rel(run(tuple(parts)))
    rec['all'] = dict(mean_rel=r.mean().item(), max_rel=r.max().item())
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        r = rel(model.encode_decode(batch['img'], batch['img_metas']).float())
    rec['autocast_whole_model'] = dict(mean_rel=r.mean().item(), max_rel=r.max().item())
    print('\n[bf16 eval error budget, 1x352x1120, random init] ' + '  '.join(f'{k}: mean {v["mean_rel"]:.2e} max {v["max_rel"]:.2e}' for k, v in rec.items()))
    _log_parity('bf16_eval_error_budget', rec)
    # bounds = measured x 2 (round 4: backbone 2.7e-3, neck 1.2e-2, PE neck 2.3e-3, head 1.3e-2, all 2.4e-2): the two deformable-attention
    # blocks of the neck and the eight convolutions of the head carry the error, not the 24 backbone layers
    budget = dict(backbone=6e-3, neck=2.5e-2, pe_mask_neck=5e-3, decode_head=2.7e-2)
    for name, b in budget.items():
        assert rec[name]['mean_rel'] <= b, (name, rec[name])
    parts_sum = sum(rec[n]['mean_rel'] for n in parts)
    assert rec['all']['mean_rel'] <= 1.5 * parts_sum + 1e-3, (rec['all'], parts_sum)
    assert rec['autocast_whole_model']['mean_rel'] <= 1.5 * parts_sum + 1e-3, rec


def test_bf16_convergence_parity_with_fp32(dev):
    """Round-3 review, missing #6: the bf16 performance mode (autocast + fused AdamW with bf16 shadow weights, MFMA attention /
    convolutions / deformable attention) must TRAIN like the fp32 path.  150 seeded steps from the same initialisation on a learnable
    synthetic task (2 x 176 x 560, four alternating batches), once in fp32 and once in bf16: the loss curves are compared at the end
    (mean of the last 20 steps) and both models are scored on a held-out batch with the reference's Abs Rel (fp32 eval of either)."""
    import copy
    from gedepth_amd.mmrt.config import Config
    from gedepth_amd.mmrt.optim import build_optimizer
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', 'depthformer_swint_v.py'))
    torch.manual_seed(0)
    base = build('depthformer_swint_v.py')
    base.init_weights()
    init = copy.deepcopy(base.state_dict())
    batches = [_learnable_batch(2, 176, 560, 100 + i, dev) for i in range(4)]
    held = _learnable_batch(2, 176, 560, 999, dev)
    steps = 150

    def train(bf16):
        torch.manual_seed(1)
        model = build('depthformer_swint_v.py')
        model.load_state_dict(init)
        model = model.to(dev).train()
        opt_cfg = dict(cfg.optimizer)
        opt_cfg['lr'] = 1e-4
        opt = build_optimizer(model, opt_cfg, cfg.optimizer_config.get('grad_clip'))
        curve = []
        for it in range(steps):
            opt.zero_grad()
            with torch.autocast('cuda', dtype=torch.bfloat16, enabled=bf16):
                out = model.train_step(batches[it % 4], opt)
            out['loss'].backward()
            opt.step()
            curve.append(float(out['log_vars']['loss']))
        model.eval()
        with torch.no_grad():
            pred = model.encode_decode(held['img'], held['img_metas']).float()
        gt = held['depth_gt']
        m = gt > 0
        absrel = ((pred[m] - gt[m]).abs() / gt[m]).mean().item()
        return curve, absrel
    c32, a32 = train(False)
    c16, a16 = train(True)
    end32, end16 = float(np.mean(c32[-20:])), float(np.mean(c16[-20:]))
    print(f'\n[convergence, {steps} steps] loss start {c32[0]:.4f}; end fp32 {end32:.4f} bf16 {end16:.4f}; held-out Abs Rel fp32 {a32:.4f} bf16 {a16:.4f}')
    _log_parity('bf16_convergence', dict(steps=steps, loss_start=c32[0], loss_end_fp32=end32, loss_end_bf16=end16, absrel_fp32=a32, absrel_bf16=a16,
                                         curve_fp32=c32[::10], curve_bf16=c16[::10]))
    assert end32 < 0.7 * c32[0], 'the task must be learnable: the fp32 run did not converge'      # otherwise the comparison says nothing
    assert abs(end16 - end32) <= 0.1 * end32, (end16, end32)
    assert abs(a16 - a32) <= 0.03, (a16, a32)


def test_config4_ddad_native_resolution_train_step(dev):
    """configs[3]: DepthFormer-SwinL + GEDepth-Adaptive at the native DDAD resolution 1x5x1216x1936 (per-camera height
    kwarg, loading.py:923-932): one full bf16 training step on the HIP path — finite losses, every parameter gets a finite
    gradient, and the deformable-attention backward runs the binned path (6e3 value tiles per head: 24 KB LDS histograms)."""
    import ctypes
    from gedepth_amd import hip
    from gedepth_amd.depth.datasets.synthetic import synthetic_batch
    from gedepth_amd.mmrt.config import Config
    from gedepth_amd.mmrt.optim import build_optimizer
    H, W = 1216, 1936
    shapes = [(H // 4 // 2 ** i, W // 4 // 2 ** i) for i in range(4)]
    arr = (ctypes.c_int * 8)(*[v for hw in shapes for v in hw])
    out4 = (ctypes.c_int * 4)()
    nv = sum(h * w for h, w in shapes)
    hip.check(hip.lib().ge_msda_bwd_plan(ctypes.cast(arr, ctypes.c_void_p), 1, nv, (H // 2) * (W // 2), 8, 4, 8,
                                         ctypes.cast(out4, ctypes.c_void_p)), 'ge_msda_bwd_plan')
    assert out4[0] == 1 and out4[1] == 64, list(out4)           # binned; 8 heads x 64 query ranges = 512 work units of the counting sort
    kitti4 = (ctypes.c_int * 4)()
    ks = [(88, 280), (44, 140), (22, 70), (11, 35)]
    hip.check(hip.lib().ge_msda_bwd_plan(ctypes.cast((ctypes.c_int * 8)(*[v for hw in ks for v in hw]), ctypes.c_void_p), 8, 32725,
                                         98560, 8, 4, 8, ctypes.cast(kitti4, ctypes.c_void_p)), 'ge_msda_bwd_plan')
    assert kitti4[0] == 1 and kitti4[1] == 8, list(kitti4)      # KITTI shape, 8 images: 8 x 8 x 8 units
    torch.manual_seed(0)
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', 'depthformer_a_ddad.py'))
    cfg.model.pretrained = None
    from gedepth_amd.depth.models import build_depther
    model = build_depther(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))
    model.init_weights()
    model = model.to(dev).train()
    optimizer = build_optimizer(model, cfg.optimizer, cfg.optimizer_config.get('grad_clip'))
    batch = synthetic_batch(1, H, W, seed=4, device=dev)
    batch['height'] = torch.tensor([1.56], device=dev)
    optimizer.zero_grad()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out = model.train_step(batch, optimizer)
    out['loss'].backward()
    lv = out['log_vars']
    assert set(lv) >= {'decode.loss_depth', 'decode.loss_dynamic_pe', 'loss'} and all(np.isfinite(v) for v in lv.values()), dict(lv)
    missing = [k for k, p in model.named_parameters() if p.grad is None]
    assert not missing, missing
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())
    assert model.last_valid_mask.shape == (1, H, W) and model.last_valid_mask.dtype == torch.uint8
    optimizer.step()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p).all() for p in model.parameters())
    print(f'\n[config #4 1x1216x1936 Swin-L-A bf16] losses {dict((k, round(v, 5)) for k, v in lv.items())}')


def test_config3_swinl_adaptive_full_shape_bf16_step_vs_fp32(dev):
    """configs[2] at its real per-GPU shape — DepthFormer-SwinL + GEDepth-Adaptive, 2 x 5 x 352 x 1120 (samples_per_gpu = 2 of
    depthformer_a.py), bf16 autocast, channels-last, full step incl. the fused optimizer — against the fp32 step of the same model
    on the same batch: both losses within the bf16 bound, gradient direction preserved globally and on each of the eight
    sampling-offset / attention-weight projections (the tensors that only see d_loc / d_attw of the deformable attention)."""
    from gedepth_amd import kernels
    from gedepth_amd.depth.datasets.synthetic import synthetic_batch
    from gedepth_amd.depth.models.utils import to_channels_last
    from gedepth_amd.mmrt.config import Config
    from gedepth_amd.mmrt.optim import build_optimizer
    kernels.FALLBACKS.clear()
    torch.manual_seed(0)
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', 'depthformer_a.py'))
    assert cfg.data.samples_per_gpu == 2 and cfg.model.backbone.embed_dims == 192 and 'dynamic_pe_neck' in cfg.model
    model = build('depthformer_a.py')
    model.init_weights()
    model = model.to(dev).train()
    to_channels_last(model)
    batch = synthetic_batch(2, 352, 1120, seed=1234, device=dev)
    ref = model.train_step(batch, None)                     # fp32 storage + arithmetic (exact-fp32 window attention)
    ref['loss'].backward()
    g32 = _grad_vector(model)
    msda_names = [n for n, _ in model.named_parameters() if 'sampling_offsets' in n or 'attention_weights' in n]
    named32 = {n: p.grad.detach().double().flatten().clone() for n, p in model.named_parameters() if n in msda_names}
    all32 = {n: p.grad.detach().double().flatten().clone() for n, p in model.named_parameters()
             if n.startswith(('pe_mask_neck.', 'dynamic_pe_neck.'))}
    for p in model.parameters():
        p.grad = None
    optimizer = build_optimizer(model, cfg.optimizer, cfg.optimizer_config.get('grad_clip'))
    optimizer.zero_grad()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out = model.train_step(batch, optimizer)
    out['loss'].backward()
    optimizer.arena.collect()
    g16 = _grad_vector(model)
    assert not kernels.FALLBACKS, f'modules fell back to ATen on the hot path: {kernels.FALLBACKS}'
    cos = torch.nn.functional.cosine_similarity(g16.double(), g32.double(), dim=0).item()
    nrm = (g16.norm() / g32.norm()).item()
    rels = {k: abs(out['log_vars'][k] - v) / abs(v) for k, v in ref['log_vars'].items()}
    per = {n: torch.nn.functional.cosine_similarity(p.grad.detach().double().flatten(), named32[n], dim=0).item()
           for n, p in model.named_parameters() if n in msda_names}
    for n, c in per.items():
        print(f'   {n:48s} cosine {c:.5f}  |g32| {named32[n].norm().item():.3e}')
    worst = min(per.values())
    print(f'\n[config #3 2x352x1120 Swin-L-A] losses fp32 {dict(ref["log_vars"])} bf16 {dict(out["log_vars"])}; grad cosine {cos:.5f}, '
          f'|g_bf16|/|g_fp32| {nrm:.4f}, worst sampling-projection cosine {worst:.5f}')
    _log_parity('config3_bf16_vs_fp32', dict(loss_rel=rels, grad_cosine=cos, grad_norm_ratio=nrm, worst_msda_projection_cosine=worst))
    assert set(out['log_vars']) >= {'decode.loss_depth', 'decode.loss_dynamic_pe', 'loss'}
    assert all(r <= 2e-2 for r in rels.values()), rels
    # Per-projection bound: at initialisation the gradient of the CROSS-attention's sampling offsets is what is left after the
    # per-query contributions (zero-mean: the value maps carry no structure yet) cancel over 2 x 98 560 queries (|g| = 2.8e-2 against
    # 1.8e-1 for the self-attention), while the bf16 rounding of gradient / value ROWS (2^-9 per element) does not cancel: measured
    # cosine 0.77 / 0.79 (weight / bias) at 2 images, 0.987 at 8 images (config #2), >= 0.94 for the other six tensors.  The kernels
    # themselves are exact on bf16-rounded inputs (tests/test_kernels_gpu.py::test_msda_bf16_gradients_vs_oracle: 2e-4 / 2e-5).
    # Seen over eight runs of this test on the same build: 0.42, 0.78, 0.78, 0.79, 0.86, 0.87, 0.87, 0.89 for that tensor (fp32 atomics in
    # the weight-gradient / d_value reductions make the run-to-run rounding differ) while the global cosine stayed in 0.995 - 0.999: the
    # cross-attention offset tensors are REPORTED (parity_e2e.json), the assertions are on what carries signal.
    # Global norm ratio: bound 0.9 - 1.1 as for config #2.  Round 4 localised why this quantity moves by several per cent between
    # identical bf16 runs (1.007 ... 1.059 over six runs of one build, tools/ubench/diag_cfg3.py, profiles/r4_config3_gradient_diag.txt):
    # the synthetic batch has valid ground truth above the horizon, where the ground embedding is zero and pred = relu(c) (1 - y) + 1e-3;
    # SigLoss' d/dpred ~ 1 / pred makes the ~100 pixels whose regressor pre-activation c is barely positive carry 84 % of |dL/dc|^2,
    # and at those pixels 1 / (c (1 - y) + 2e-3) changes by a factor under a 1e-2 perturbation of c (bf16 activations; the order of fp32
    # atomics in the forward: loss differs by 4e-5 between identical runs).  The weight gradients still agree in direction (bulk pixels
    # add coherently: decode head cosine 0.997 - 0.9998) but their norm carries that heavy-tailed term.  What pins the SCALE of the
    # loss gradient instead is the ground-attention branch: it receives the same dL/dpred through d pe / d y, which has no such
    # singularity — asserted below to 1 % in norm and 1e-4 in direction (measured 0.997 - 1.0004, cosine >= 0.99996).
    assert torch.isfinite(g16).all() and cos >= 0.99 and 0.9 <= nrm <= 1.1, (cos, nrm, worst)
    pe_names = [n for n, _ in model.named_parameters() if n.startswith(('pe_mask_neck.', 'dynamic_pe_neck.'))]
    g16_pe = torch.cat([dict(model.named_parameters())[n].grad.detach().double().flatten() for n in pe_names])
    g32_pe = torch.cat([all32[n] for n in pe_names])
    cos_pe = torch.nn.functional.cosine_similarity(g16_pe, g32_pe, dim=0).item()
    nrm_pe = (g16_pe.norm() / g32_pe.norm()).item()
    print(f'   ground-attention branch (pe_mask_neck + dynamic_pe_neck): cosine {cos_pe:.6f}, |g_bf16|/|g_fp32| {nrm_pe:.5f}')
    _log_parity('config3_bf16_vs_fp32_pe_branch', dict(grad_cosine=cos_pe, grad_norm_ratio=nrm_pe))
    assert cos_pe >= 0.9999 and 0.99 <= nrm_pe <= 1.01, (cos_pe, nrm_pe)
    cross_offsets = [n for n in per if 'multi_att.sampling_offsets' in n]
    assert all(per[n] > 0.2 for n in cross_offsets), per
    assert all(c >= 0.8 for n, c in per.items() if n not in cross_offsets), per
    optimizer.step()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p).all() for p in model.parameters())
    # Second phase (round-4 review): the bound above on the cross-attention's offset projection asserts little because AT INITIALISATION
    # that gradient is the residue of cancelling zero-mean terms.  After 20 optimizer steps the value maps carry structure and the
    # same comparison is well-conditioned: bf16 step vs fp32 step at the trained state, the two tensors held to cosine >= 0.95.
    for _ in range(19):
        optimizer.zero_grad()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            o = model.train_step(batch, optimizer)
        o['loss'].backward()
        optimizer.step()
    optimizer.zero_grad()
    optimizer.arena.refresh_shadow(copy=True)
    r2 = model.train_step(batch, None)                      # fp32 at the trained state
    r2['loss'].backward()
    optimizer.arena.collect()
    t32 = {n: p.grad.detach().double().flatten().clone() for n, p in model.named_parameters() if n in msda_names}
    optimizer.zero_grad()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        o2 = model.train_step(batch, optimizer)
    o2['loss'].backward()
    optimizer.arena.collect()
    per2 = {n: torch.nn.functional.cosine_similarity(p.grad.detach().double().flatten(), t32[n], dim=0).item()
            for n, p in model.named_parameters() if n in msda_names}
    for n, c in per2.items():
        print(f'   after 20 steps {n:48s} cosine {c:.5f}  |g32| {t32[n].norm().item():.3e}')
    _log_parity('config3_bf16_vs_fp32_after_20_steps', dict(per_projection_cosine=per2, loss_fp32=dict(r2['log_vars']), loss_bf16=dict(o2['log_vars'])))
    assert all(per2[n] >= 0.95 for n in cross_offsets), per2


def test_config5_fp8_window_attention_workload(dev):
    """configs[4] at the WORKLOAD level (the kernel-level bound is tests/test_kernels_gpu.py::test_window_attention_fp8_forward):
    the whole model with every window attention on the fp8 (OCP e4m3) MFMA forward (`kernel_variant = 3`, bench.py --attn fp8),
    (1) eval depth at 1 x 5 x 352 x 1120 against the exact-fp32 HIP path, with the bf16-MFMA attention model on the same weights as
    the yardstick — the restated tolerance: fp8 attention may cost at most 3x the bf16 path's own error, and stays inside 5 % mean /
    relative depth error; (2) one full 8 x 352 x 1120 bf16 training step (fp8 forward, bf16 backward kernels) against the bf16 step:
    loss within 2 %, gradient cosine >= 0.9, every gradient finite, optimizer step finite."""
    from gedepth_amd.depth.datasets.synthetic import synthetic_batch
    from gedepth_amd.mmrt.config import Config
    from gedepth_amd.mmrt.optim import build_optimizer
    torch.manual_seed(0)
    model = build('depthformer_swint_v.py')
    model.init_weights()
    model = model.to(dev)

    def variant(v):
        n = 0
        for mod in model.modules():
            if hasattr(mod, 'kernel_variant'):
                mod.kernel_variant = v
                n += 1
        assert n == 12                                       # every Swin-T block

    one = synthetic_batch(1, 352, 1120, seed=1234, device=dev)
    model.eval()
    with torch.no_grad():
        variant(1)
        d32 = model.encode_decode(one['img'], one['img_metas']).float()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            variant(0)
            d16 = model.encode_decode(one['img'], one['img_metas']).float()
            variant(3)
            d8 = model.encode_decode(one['img'], one['img_metas']).float()
    rel = lambda a: ((a - d32).abs() / d32.abs().clamp_min(1e-3))
    e16, e8 = rel(d16), rel(d8)
    print(f'\n[config #5 1x352x1120 eval] depth rel err vs fp32: bf16 attention mean {e16.mean().item():.2e} max {e16.max().item():.2e}; '
          f'fp8 attention mean {e8.mean().item():.2e} max {e8.max().item():.2e}')
    assert torch.isfinite(d8).all() and not torch.equal(d8, d16)                 # the fp8 kernels really ran
    assert e8.mean().item() <= max(3.0 * e16.mean().item(), 1e-3) and e8.mean().item() <= 5e-2, (e8.mean().item(), e16.mean().item())
    # ---- one full training step at the bench shape
    model.train()
    batch = synthetic_batch(8, 352, 1120, seed=1234, device=dev)
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', 'depthformer_swint_v.py'))
    optimizer = build_optimizer(model, cfg.optimizer, cfg.optimizer_config.get('grad_clip'))
    res = {}
    for tag, v in (('bf16', 0), ('fp8', 3)):
        variant(v)
        optimizer.zero_grad()
        torch.manual_seed(5)                                  # same DropPath / dropout draws in both runs
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out = model.train_step(batch, optimizer)
        out['loss'].backward()
        optimizer.arena.collect()
        res[tag] = (out['log_vars']['loss'], _grad_vector(model).clone())
    (l16, g16), (l8, g8) = res['bf16'], res['fp8']
    cos = torch.nn.functional.cosine_similarity(g8.double(), g16.double(), dim=0).item()
    print(f'[config #5 8x352x1120 train step] loss bf16 {l16:.5f} fp8 {l8:.5f}; gradient cosine fp8 vs bf16 {cos:.4f}')
    _log_parity('config5_fp8', dict(eval_mean_rel_bf16=e16.mean().item(), eval_mean_rel_fp8=e8.mean().item(), eval_max_rel_fp8=e8.max().item(),
                                    loss_bf16=l16, loss_fp8=l8, grad_cosine=cos))
    assert abs(l8 - l16) <= 2e-2 * abs(l16) and torch.isfinite(g8).all() and cos >= 0.9, (l16, l8, cos)
    optimizer.step()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p).all() for p in model.parameters())


def test_ddad_per_camera_height_vs_oracle(dev):
    """The DDAD branch of dynamic_pe (encoder_decoder.py:88-94: per-sample camera height) through the whole fp32 model
    against the CPU oracle on the same seeded input/weights (2x5x64x96, Swin-L-A)."""
    from gedepth_amd.depth.datasets.synthetic import synthetic_batch
    model = build('depthformer_a_ddad.py')
    sd = load_filled(model, 'ddad')
    P = {k: v.clone() for k, v in sd.items()}
    model = model.to(dev).train()
    set_exact(model)
    batch = synthetic_batch(2, 64, 96, seed=11, valid_fraction=0.3)
    heights = torch.tensor([1.56, 1.53])
    cfg_ddad = dict(O.SWIN_L, adaptive=True, depth_scale=250.0)     # configs/depthformer/depthformer_v_ddad.py: model.depth_scale
    assert model.depth_scale == 250
    losses, _ = O.forward_train(batch['img'], batch['depth_gt'], batch['pe_k_gt'], P, cfg_ddad, train_bn=True, height=heights)
    _, ref = O.parse_losses(losses)
    gb = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    out = model.train_step(dict(gb, height=heights.to(dev)), None)
    for k, v in ref.items():
        assert abs(out['log_vars'][k] - v) <= 1e-5 * abs(v) + 1e-6, (k, out['log_vars'][k], v)
    out0 = model.train_step(gb, None)                       # default 1.65 m must give a different ground embedding
    assert abs(out0['log_vars']['loss'] - out['log_vars']['loss']) > 1e-6


def test_rccl_gradient_exchange_single_rank(dev):
    """The N-GPU path on one GPU: FlatDDP with the nccl (= RCCL) backend forced on a single rank (GE_DDP_FORCE=1) — bucketed
    async all-reduce from the post-accumulate hooks, fused scalar reduce — must reproduce the plain step (to fp32 atomics' run-to-run order)."""
    import subprocess
    import sys
    env = dict(os.environ, GE_DDP_FORCE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29631', RANK='0', LOCAL_RANK='0', WORLD_SIZE='1',
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'ddp_rccl_worker.py')], env=env, capture_output=True, text=True,
                       timeout=600)
    if r.returncode != 0 or 'RCCL_DDP_OK' not in r.stdout:
        print('---- worker stdout ----\n' + r.stdout[-6000:] + '\n---- worker stderr ----\n' + r.stderr[-6000:])
    assert r.returncode == 0 and 'RCCL_DDP_OK' in r.stdout


def _run_rccl_worker(world, port, arg):
    import subprocess
    import sys
    env = dict(os.environ, GE_DDP_FORCE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    worker = os.path.join(ROOT, 'tests', 'ddp_rccl_worker.py')
    if world == 1:
        cmd = [sys.executable, worker, arg]
        env.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1')
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), worker, arg]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0 or 'RCCL_GRAPH_OK' not in r.stdout:
        print('---- worker stdout ----\n' + r.stdout[-6000:] + '\n---- worker stderr ----\n' + r.stderr[-6000:])
    assert r.returncode == 0 and 'RCCL_GRAPH_OK' in r.stdout


def test_rccl_exchange_inside_hip_graph_single_rank(dev):
    """GraphedTrainStep with FlatDDP active (GE_DDP_FORCE=1, nccl backend, one rank): the bucket all-reduces are captured in the hipGraph; the
    capture waits until the process-group watchdog has retired the eager steps' collectives (``quiesce_collectives``: observed through the flight
    recorder, no timed pause); 5 graphed steps train like 5 eager steps."""
    _run_rccl_worker(1, 29633, 'graphed')


def test_rccl_world2_eager_vs_graphed(dev):
    """The same on TWO ranks over RCCL / xGMI (different batches per rank, gradients averaged): eager vs graphed for 5 steps, identical logged
    losses on both ranks.  Needs a node with >= 2 GPUs: skipped on the 1-GPU box."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs >= 2 GPUs on the node')
    _run_rccl_worker(2, 29635, 'graphed')


@pytest.mark.parametrize('amp', [False, True])
def test_channels_last_model_matches_nchw(dev, amp):
    """depth.models.utils.to_channels_last: the same parameters, the same input -> the same losses and gradients as the NCHW
    execution (fp32: to accumulation order; bf16 autocast: to bf16 rounding), with the maps really flowing channels-last."""
    from gedepth_amd import kernels
    from gedepth_amd.depth.datasets.synthetic import synthetic_batch
    from gedepth_amd.depth.models.utils import to_channels_last
    batch = synthetic_batch(2, 128, 160, seed=9, device=dev, valid_fraction=0.3)
    res = {}
    for layout in ('nchw', 'nhwc'):
        torch.manual_seed(0)
        model = build('depthformer_swint_a.py')
        load_filled(model, 'cl')
        model = model.to(dev).train()
        if not amp:
            set_exact(model)
        if layout == 'nhwc':
            to_channels_last(model)
            assert model.backbone.conv1.weight.is_contiguous(memory_format=torch.channels_last)
        seen = []
        h = model.neck.register_forward_hook(lambda m, i, o: seen.extend(kernels._is_cl(t) for t in o))
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
            out = model.train_step(batch, None)
        out['loss'].backward()
        h.remove()
        assert all(seen) == (layout == 'nhwc') and (layout == 'nchw' or any(seen)), (layout, seen)
        res[layout] = (dict(out['log_vars']), torch.cat([p.grad.flatten().float() for p in model.parameters()]))
    (la, ga), (lb, gb) = res['nhwc'], res['nchw']
    for k in lb:
        assert abs(la[k] - lb[k]) <= (2e-2 if amp else 2e-5) * abs(lb[k]) + 1e-6, (k, la[k], lb[k])
    cos = torch.nn.functional.cosine_similarity(ga.double(), gb.double(), dim=0).item()
    rel = ((ga - gb).double().norm() / gb.double().norm()).item()
    print(f'\n[channels-last vs NCHW, amp={amp}] losses {la} | gradient l2rel {rel:.2e}, cosine {cos:.6f}')
    assert rel <= (0.15 if amp else 2e-3) and cos >= (0.99 if amp else 0.999999), (rel, cos)


def test_graphed_train_step_matches_eager_steps(dev):
    """gedepth_amd/mmrt/graph.py: forward + losses + backward + clip + AdamW captured in ONE hipGraph and replayed must train like the
    same steps launched from Python.  Two copies of the Adaptive model (stochastic depth and attention dropout off, so that both runs
    are deterministic up to the order of fp32 atomics), the same four batches: one eager run, one run through GraphedTrainStep (1 eager
    warm-up step, capture, 3 replays on batches COPIED into the static input buffers).  Losses of every step and the parameters after
    the last one must agree; the optimizer's step counter (bias corrections) advances per replay; the dropout counter advances INSIDE
    the graph (one per replay)."""
    from gedepth_amd.depth.datasets.synthetic import synthetic_batch
    from gedepth_amd.mmrt.config import Config
    from gedepth_amd.mmrt.graph import GraphedTrainStep
    from gedepth_amd.mmrt.optim import build_optimizer
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', 'depthformer_swint_a.py'))
    batches = [synthetic_batch(2, 128, 160, seed=40 + i, device=dev, valid_fraction=0.3) for i in range(4)]

    def make():
        torch.manual_seed(0)
        m = build('depthformer_swint_a.py', drop_path_rate=0.0)
        m.neck.multi_att.dropout.p = 0.0
        m.neck.self_attn.dropout.p = 0.0
        load_filled(m, 'graph')
        m = m.to(dev).train()
        return m, build_optimizer(m, cfg.optimizer, cfg.optimizer_config.get('grad_clip'))
    losses = {}
    finals = {}
    for mode in ('eager', 'graph'):
        model, opt = make()
        log = []
        if mode == 'eager':
            for b in batches:
                opt.zero_grad()
                with torch.autocast('cuda', dtype=torch.bfloat16):
                    out = model.train_step(b, opt)
                out['loss'].backward()
                opt.step()
                log.append(dict(out['log_vars']))
        else:
            static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batches[0].items()}
            gs = GraphedTrainStep(model, opt, static, amp_dtype=torch.bfloat16, warmup=1)
            for b in batches:
                out = gs(b)
                log.append(dict(out['log_vars']))
            assert gs.graph is not None and gs.replays == 3 and int(gs.salt.item()) == 3
            assert opt.step_count == 4
            gs.release()
        torch.cuda.synchronize()
        losses[mode] = log
        finals[mode] = {n: p.detach().float().clone() for n, p in model.named_parameters()}
    for i, (a, b) in enumerate(zip(losses['eager'], losses['graph'])):
        for k in a:
            # bf16 steps whose fp32-atomic order differs drift apart slowly (two eager runs do too): 2.1e-3 seen at the fourth step
            assert abs(a[k] - b[k]) <= (2e-3 if i < 2 else 6e-3) * abs(a[k]) + 1e-6, (i, k, a[k], b[k])
    # four AdamW steps at lr 1e-4: every element moves by about lr per step whatever the size of its gradient, so elements whose true
    # gradient is zero (stage-norm biases in front of a training-mode BatchNorm) take a random walk that differs between ANY two runs (fp32
    # atomics); what must agree is the update as a whole: direction of the total parameter change, and its size
    start = {n: p.detach().float() for n, p in make()[0].named_parameters()}
    de = torch.cat([(finals['eager'][n] - start[n]).flatten() for n in start]).double()
    dg = torch.cat([(finals['graph'][n] - start[n]).flatten() for n in start]).double()
    cos = torch.nn.functional.cosine_similarity(de, dg, dim=0).item()
    ratio = (dg.norm() / de.norm()).item()
    moved = de.abs().max().item()
    print(f'\n[hipGraph step vs eager] losses {losses["eager"][-1]} / {losses["graph"][-1]}; update cosine {cos:.5f}, norm ratio {ratio:.4f}, largest move {moved:.2e}')
    _log_parity('graphed_step_vs_eager', dict(update_cosine=cos, update_norm_ratio=ratio, last_loss_eager=losses['eager'][-1]['loss'], last_loss_graph=losses['graph'][-1]['loss']))
    assert moved > 1e-4 and cos >= 0.97 and abs(ratio - 1) <= 0.02, (cos, ratio, moved)


def test_dropout_counter_changes_the_mask_per_replay(dev):
    """The dropout masks of the neck's glue kernels are a hash of (seed ARGUMENT, element): captured in a graph the argument is frozen.
    With ge_rng_salt registered the device counter enters the hash at execution time: same seed, different counter -> different mask;
    forward and backward of one step (same counter) agree; no counter -> the round-4 behaviour."""
    from gedepth_amd import hip, kernels as K
    tok = torch.randn(2, 64, 128, device=dev).bfloat16()
    ident = torch.zeros_like(tok)
    salt = torch.zeros(1, device=dev, dtype=torch.int64)
    base = K.residual_dropout(ident, tok, 0.5, seed=77).float()
    hip.check(hip.lib().ge_rng_salt(salt.data_ptr()), 'ge_rng_salt')
    try:
        outs = []
        for v in (0, 1, 2, 1):
            salt.fill_(v)
            t = tok.clone().requires_grad_(True)
            o = K.residual_dropout(ident, t, 0.5, seed=77)
            o.float().sum().backward()
            kept_f, kept_b = (o != 0), (t.grad != 0)
            assert torch.equal(kept_f, kept_b), 'backward must regenerate the forward mask'
            outs.append(o.float())
    finally:
        hip.lib().ge_rng_salt(None)
    assert torch.equal(outs[0], base)                       # counter 0 = no counter
    assert torch.equal(outs[1], outs[3]) and not torch.equal(outs[1], outs[2]) and not torch.equal(outs[0], outs[1])
    keep = [(o != 0).float().mean().item() for o in outs]
    assert all(abs(k - 0.5) < 0.03 for k in keep), keep
    assert torch.equal(K.residual_dropout(ident, tok, 0.5, seed=77).float(), base)


def test_linear_weight_gradients_are_written_into_the_arena(dev):
    """optim.grad_target: the split-K weight gradient of a token Linear is summed straight into the parameter's slice of the gradient arena
    (autograd adopts the alias, GradArena.collect has nothing to copy).  Checked on the Swin-T model: after backward, before collect, the
    large Linear weights' ``.grad`` IS arena memory; the gradients equal those of the copying path (slices hidden from the producers)."""
    from gedepth_amd.depth.datasets.synthetic import synthetic_batch
    from gedepth_amd.mmrt.config import Config
    from gedepth_amd.mmrt.optim import build_optimizer
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', 'depthformer_swint_v.py'))
    torch.manual_seed(0)
    model = build('depthformer_swint_v.py')
    load_filled(model, 'arena_grads')
    model = model.to(dev).train()
    opt = build_optimizer(model, cfg.optimizer, cfg.optimizer_config.get('grad_clip'))
    batch = synthetic_batch(2, 352, 704, seed=3, device=dev, valid_fraction=0.3)
    names = {id(p): n for n, p in model.named_parameters()}

    def run():
        opt.zero_grad()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out = model.train_step(batch, opt)
        out['loss'].backward()
        direct = [names[id(p)] for p, v in zip(opt.arena.params, opt.arena.views) if p.grad is not None and p.grad.data_ptr() == v.data_ptr()]
        opt.arena.collect()
        return direct, {names[id(p)]: v.detach().clone() for p, v in zip(opt.arena.params, opt.arena.views)}
    direct, g_new = run()
    saved = [p._ge_grad_view for p in opt.arena.params]
    for p in opt.arena.params:
        p._ge_grad_view = None
    direct_off, g_old = run()
    for p, v in zip(opt.arena.params, saved):
        p._ge_grad_view = v
    assert not direct_off
    big = [n for n in direct if n.endswith('.weight')]
    elems = sum(g_new[n].numel() for n in direct)
    print(f'\n[arena-direct gradients] {len(direct)} tensors, {elems / 1e6:.1f} M of {opt.arena.numel / 1e6:.1f} M elements: e.g. {big[:3]}')
    assert any('attn.w_msa.qkv.weight' in n for n in direct) and any('ffn.layers.1.weight' in n for n in direct) and any('decode_head.conv_list' in n for n in direct), direct[:10]
    assert elems >= 0.9 * opt.arena.numel, (elems, opt.arena.numel)        # Linear (split-K and plain) and convolution weights; what is left are biases / norms
    # VALUES: two runs of a whole step differ by the order of fp32 atomics upstream (and this batch has the heavy-tailed loss gradient of
    # single tensors moved by 2 - 20 % between identical runs), so the two paths are compared where nothing upstream is noisy — one
    # token Linear (split-K and plain), one library convolution and one MFMA 3x3 convolution, each alone: bit-identical
    from gedepth_amd import kernels as K
    from gedepth_amd.mmrt import bricks
    from gedepth_amd.mmrt.optim import GradArena
    torch.manual_seed(1)
    lin_big, lin_small = bricks.Linear(96, 192).to(dev), bricks.Linear(96, 192).to(dev)
    conv1, conv3 = bricks.ConvModule(64, 96, 1, bias=False, act_cfg=None).to(dev), bricks.ConvModule(64, 64, 3, padding=1, act_cfg=dict(type='LeakyReLU')).to(dev)
    conv1_lib = bricks.ConvModule(64, 96, 1, act_cfg=None).to(dev)
    for m in (conv1, conv3, conv1_lib):
        m.to(memory_format=torch.channels_last)
    params = [lin_big.weight, lin_small.weight, conv1.conv.weight, conv3.conv.weight, conv1_lib.conv.weight]
    arena = GradArena([p for m in (lin_big, lin_small, conv1, conv3, conv1_lib) for p in m.parameters()])
    xb, xs = torch.randn(4, 4096, 96, device=dev), torch.randn(2, 300, 96, device=dev)
    xc = torch.randn(2, 64, 120, 160, device=dev).contiguous(memory_format=torch.channels_last)
    # what the model feeds its 1x1 layers under autocast: a bf16 channels-last map of <= 65 536 rows -> the token-GEMM path (kernels._Conv1x1Gemm,
    # fp32 accumulators written into the arena); the fp32 map sends conv1_lib through kernels._ConvLib -> MIOpen, whose dw is a bf16 tensor
    xc16 = xc.to(torch.bfloat16)
    assert K.conv1x1_gemm_ok(conv1.conv, xc16) and not K.conv1x1_gemm_ok(conv1_lib.conv, xc)

    def small():
        arena.zero_grad()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            loss = (lin_big(xb).float().square().mean() + lin_small(xs).float().square().mean() + conv1(xc16).float().square().mean()
                    + conv3(xc).float().square().mean() + conv1_lib(xc).float().square().mean())
        loss.backward()
        hit = [p.grad is not None and p.grad.data_ptr() == p._ge_grad_view.data_ptr() if p._ge_grad_view is not None else False for p in params]
        arena.collect()
        return hit, [p.grad.detach().clone() for p in params]
    hit, g1 = small()
    assert all(hit), hit
    keep = [p._ge_grad_view for p in arena.params]
    for p in arena.params:
        p._ge_grad_view = None
    hit0, g0 = small()
    for p, v in zip(arena.params, keep):
        p._ge_grad_view = v
    assert not any(hit0)
    # tolerance per path, never tighter than what the path's storage type and reduction order can hold between two identical runs:
    #   0, 1  token Linear (split-K bmm + ordered sum / plain mm): deterministic -> bit-identical
    #   2     1x1 as token GEMM: the same two reductions with fp32 accumulators (a library GEMM: allowed its last fp32 bits)
    #   3     MFMA 3x3 weight gradient: K-split partials flushed with fp32 atomics -> order-dependent last fp32 bits
    #   4     MIOpen 1x1 weight gradient: a bf16 tensor from a solver that may split K with atomics -> one bf16 ulp (2^-8) apart is legitimate
    bounds = [(0.0, 0.0), (0.0, 0.0), (1e-5, 1e-5), (1e-4, 1e-4), (1.6e-2, 8e-3)]
    for (rtol, atol_rel), a_, b_, p in zip(bounds, g1, g0, params):
        assert torch.isfinite(a_).all() and a_.abs().max().item() > 0
        if rtol == 0.0:
            assert torch.equal(a_, b_), (tuple(p.shape), (a_ - b_).abs().max().item())
        else:
            assert torch.allclose(a_, b_, rtol=rtol, atol=atol_rel * b_.abs().max().item()), (tuple(p.shape), (a_ - b_).abs().max().item(), b_.abs().max().item())


def test_runner_with_hip_graph_trains_like_the_eager_runner(dev, tmp_path):
    """``IterBasedRunner(hip_graph=True)`` (what ``tools/train.py --hip-graph`` builds): the hot loop with LR hook, logger hook and the captured
    step — 3 eager iterations, capture, replays on loader batches copied into the static buffers; the optimizer hook steps aside.  Against
    the eager runner on the same loader and seeds: the learning rate schedule reaches the device (first and last logged lr equal), the
    logged losses agree, the step counter advanced once per iteration."""
    from gedepth_amd.depth.datasets.loader import SyntheticKITTI
    from gedepth_amd.mmrt.config import Config
    from gedepth_amd.mmrt.optim import build_optimizer
    from gedepth_amd.mmrt.runner import IterBasedRunner
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', 'depthformer_swint_a.py'))
    data = SyntheticKITTI(8, 128, 160, adaptive=True, seed=5)
    from gedepth_amd.depth.datasets.loader import build_dataloader
    results = {}
    for mode in (False, True):
        torch.manual_seed(0)
        model = build('depthformer_swint_a.py')
        load_filled(model, 'runner_graph')
        model = model.to(dev).train()
        opt = build_optimizer(model, cfg.optimizer, cfg.optimizer_config.get('grad_clip'))
        logs = []
        runner = IterBasedRunner(model, opt, work_dir=str(tmp_path / f'g{int(mode)}'), logger=logs.append, max_iters=8, amp_dtype=torch.bfloat16,
                                 hip_graph=mode)
        runner.register_training_hooks(dict(policy='CosineAnnealing', min_lr_ratio=1e-2, warmup='linear', warmup_iters=4, warmup_ratio=0.1, by_epoch=False),
                                       dict(grad_clip=cfg.optimizer_config.get('grad_clip')), None, dict(interval=2, hooks=[dict(type='TextLoggerHook')]))
        loader = build_dataloader(data, 2, 0, dist=False, seed=3, shuffle=False, drop_last=True)
        runner.run([loader])
        torch.cuda.synchronize()
        assert runner.iter == 8 and opt.step_count == 8
        if mode:
            assert runner.graphed is None and runner.graph_stats == dict(captured=True, replays=5, disabled=False)      # released at the end of run()
        results[mode] = dict(lr=runner.current_lr, loss=float(runner.outputs['log_vars']['loss']),
                             params=torch.cat([p.detach().float().flatten() for p in model.parameters()]))
    assert results[False]['lr'] == results[True]['lr']
    a, b = results[False], results[True]
    assert abs(a['loss'] - b['loss']) <= 1e-2 * abs(a['loss']), (a['loss'], b['loss'])
    print(f'\n[runner hip_graph] last loss eager {a["loss"]:.5f} graph {b["loss"]:.5f}')
