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


@pytest.mark.parametrize('cfg_name,tag,adaptive', [('depthformer_swint_v.py', 'e2e_T_V', False),
                                                   ('depthformer_swint_a.py', 'e2e_T_A', True),
                                                   ('depthformer_a.py', 'e2e_L_A', True)])
def test_e2e_vs_reference_fixture(dev, golden, cfg_name, tag, adaptive):
    g = golden(tag)
    model = build(cfg_name)
    load_filled(model, 'e2e')
    model = model.to(dev)
    set_exact(model)
    img, gt, kgt = T(g['img']).to(dev), T(g['depth_gt']).to(dev), T(g['pe_k_gt']).to(dev)
    metas = [dict(flip=False, ori_shape=(64, 96, 3))] * 2
    model.eval()
    with torch.no_grad():
        depth = model.encode_decode(img, metas)
    ref = T(g['depth_eval'])
    err = ((depth.cpu() - ref).abs() / ref.abs().clamp_min(1e-3)).max().item()
    assert err < 1e-3, f'eval depth rel err {err:.3e}'      # conv/GEMM library accumulation order differs
    model.train()
    kw = dict(pe_k_gt=kgt) if adaptive else {}
    out = model.train_step(dict(img=img, img_metas=metas, depth_gt=gt, **kw), None)
    names = json.loads(str(g['loss_names']))
    for n, v in zip(names, g['loss_values']):
        assert abs(out['log_vars'][n] - v) <= 1e-5 * abs(v) + 1e-6, (n, out['log_vars'][n], v)
    out['loss'].backward()
    params = dict(model.named_parameters())
    assert all(p.grad is not None for p in params.values())
    for k in g.files:
        if k.startswith('grad::'):
            gr = params[k[6:]].grad.flatten()
            gr = gr[::max(1, gr.numel() // 50000)].cpu()
            refg = T(g[k])
            l2rel = ((gr - refg).norm() / (refg.norm() + 1e-30)).item()
            # Swin-T: <= 5e-5 everywhere except four stage-0 tensors at ~4e-4 since LayerNorm runs as the HIP kernel (2.5e-5
            # with ATen's): qkv.bias / relative_position_bias_table of blocks 0-1 — directions the softmax is invariant to,
            # whose gradients are sums that cancel to rounding level — and the patch-embed weight behind them.  The kernel is
            # closer to float64 than ATen's on every LayerNorm of this model (7e-8 vs 8e-8 rel. l2, forward and backward,
            # scratch/dbg_ln3.py); the fixture comes from CPU kernels that round like ATen's.
            # Swin-L (24 blocks, fill-rule weights): ~6e-3 on the backbone gradients although the losses agree to 1e-7 and
            # every conv / linear / norm layer's own backward agrees to 1e-6 with a float64 CPU recomputation (scratch
            # probes, DESIGN.md "open items") — the same sensitivity; bounded here, tracked there.
            tol = 1e-3 if 'T' in tag else 1.5e-2
            assert l2rel <= tol, (k, l2rel)
    total = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params.values())).item()
    assert abs(total - float(g['grad_norm_total'])) <= (2e-4 if 'T' in tag else 5e-3) * float(g['grad_norm_total'])


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
    assert rel.max().item() < 2e-3, rel.max().item()
    assert rel.mean().item() < 1e-4, rel.mean().item()


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
