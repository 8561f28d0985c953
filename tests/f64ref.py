"""float64 run of the CPU oracle on an end-to-end fixture's inputs (test infrastructure).

fp32 gradients of a 24-block Swin differ between any two correct implementations by far more than fp32 epsilon on a
few tensors (softmax-invariant directions whose gradients cancel to rounding level), so "who is right" is settled
against the same algorithm evaluated in float64: ``run(tag)`` returns the float64 losses, eval depth and gradients,
and ``l2rel`` measures a candidate against them.  Used by tests/test_oracle_golden.py (reference fixture vs float64)
and tests/test_model_gpu.py (HIP path vs float64)."""
import json
import os

import numpy as np
import torch

from oracle import gedepth_oracle as O
from oracle.fill import fill_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
_STATS = ('running_mean', 'running_var')
_INT = ('relative_position_index', 'num_batches_tracked')


def sample(t, cap=50000):
    """the strided sample rule of tests/golden/make_golden.py::grad_sample"""
    t = t.detach().flatten()
    return t[::max(1, t.numel() // cap)]


def l2rel(a, ref):
    a, ref = a.double().flatten().cpu(), ref.double().flatten().cpu()
    return ((a - ref).norm() / (ref.norm() + 1e-300)).item()


def run(tag, dtype=torch.float64, salt='e2e', kink=None, force=None, record=False):
    """kink = (tau_px, tau_act, sign): resolve sampling locations within tau_px pixels of a bilinear kink and pre-activations
    within tau_act of zero to one side (oracle.KINK_NUDGE)."""
    g = np.load(os.path.join(GOLDEN, tag + '.npz'), allow_pickle=False)
    spec = json.loads(str(g['spec']))
    P32 = fill_state_dict([(k, s) for k, s in spec], salt)
    P = {}
    for k, v in P32.items():
        if not v.is_floating_point():
            P[k] = v
        elif k.endswith(_STATS):
            P[k] = v.to(dtype)
        else:
            P[k] = v.to(dtype).requires_grad_(True)
    arch = dict(O.SWIN_L if '_L_' in tag else O.SWIN_T, adaptive=tag.endswith('A'))
    img = torch.from_numpy(g['img']).to(dtype)
    gt = torch.from_numpy(g['depth_gt']).to(dtype)
    kgt = torch.from_numpy(g['pe_k_gt'])
    O.KINK_NUDGE, O.KINK_COUNT[:] = kink, [0, 0, 0, 0]
    # force = dict(act={site: mask}, floors=[cells per msda call]): take another implementation's side at every kink
    O.KINK_FORCE = None if force is None else dict(act=dict(force.get('act', {})), floors=list(force.get('floors', [])))
    for k in O.FORCE_STATS:
        O.FORCE_STATS[k] = 0 if k.endswith(('flipped', 'seen')) else 0.0
    O.KINK_RECORD = dict(act={}, floors=[]) if record else None
    try:
        losses, _ = O.forward_train(img, gt, kgt, P, arch, train_bn=True)
    finally:
        O.KINK_NUDGE = None
        O.KINK_FORCE = None
        decisions, O.KINK_RECORD = O.KINK_RECORD, None
    kink_count = tuple(O.KINK_COUNT)
    loss, _ = O.parse_losses(losses)
    log_vars = {k: v.item() for k, v in losses.items()}
    log_vars['loss'] = loss.item()
    loss.backward()
    with torch.no_grad():
        depth_eval = O.encode_decode(img, {k: v.detach() for k, v in P.items()}, arch)
    grads = {k: v.grad for k, v in P.items() if v.is_floating_point() and v.requires_grad}
    return dict(log_vars=log_vars, depth_eval=depth_eval, grads=grads, fixture=g, kink_count=kink_count, force_stats=dict(O.FORCE_STATS), decisions=decisions)


def kink_spread(tag, tau_px=1e-4, tau_act=2e-5):
    """Per tensor, how far apart two float64 evaluations are that only differ in the side taken at gradient kinks within
    fp32 rounding reach (bilinear sampling coordinates within tau_px pixels of an integer, ReLU / LeakyReLU pre-activations
    within tau_act of zero): the part of the gradient the algorithm leaves undefined for any fp32 implementation."""
    up, down = run(tag, kink=(tau_px, tau_act, +1.0)), run(tag, kink=(tau_px, tau_act, -1.0))
    spread = {k: l2rel(sample(up['grads'][k]), sample(down['grads'][k])) for k in up['grads']}
    return spread, up['kink_count']
