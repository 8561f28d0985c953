"""mmcv-full 1.3.13 is not vendored in the reference, so ``ConvModule`` / ``FFN`` / ``build_norm_layer`` / ``DropPath`` reach the fixture
generator as THIS repository's restatement (gedepth_amd/mmrt/bricks.py): fixtures of modules built on them cannot pin them.  This file is
the independent check the round-4 review asked for: the documented behaviour of that mmcv release, written out
below in plain torch.nn.functional WITHOUT using any code of ``bricks``, against the bricks modules on the same parameters:

  ConvModule (mmcv/cnn/bricks/conv_module.py): order ('conv', 'norm', 'act'); ``bias='auto'`` means bias iff there is no norm layer; the
    norm layer is built on out_channels and registered under the abbreviation of its type ('bn', 'gn', 'ln'); activations are in place
    unless Tanh / PReLU / Sigmoid / HSigmoid / Swish (/ GELU); init = kaiming_normal(fan_out, nonlinearity of the activation, a =
    negative_slope for LeakyReLU) on the conv weight, zeros on its bias, ones / zeros on the norm.
  FFN (mmcv/cnn/bricks/transformer.py): Sequential(Linear, act, Dropout) x (num_fcs - 1), Linear, Dropout; forward(x, identity=None):
    out = layers(x); without add_identity: dropout_layer(out); else (identity if given else x) + dropout_layer(out); state-dict keys
    layers.{i}.0.* and layers.{num_fcs - 1}.*.
  build_norm_layer: ('bn', BatchNorm2d(eps 1e-5)) / ('ln', LayerNorm) / ('gn', GroupNorm(num_groups)); requires_grad copied to the parameters.
  DropPath (mmcv/cnn/bricks/drop.py): x / keep * floor(keep + U[0,1)) per SAMPLE in training, identity in eval or at drop_prob 0.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from gedepth_amd.mmrt import bricks


def _doc_conv_module(x, P, stride, padding, dilation, groups, norm, act, train, slope=0.01, num_groups=None, eps=1e-5):
    """The documented ConvModule.forward on a dict of parameters (mmcv 1.3.13 semantics, no bricks code)."""
    y = F.conv2d(x, P['conv.weight'], P.get('conv.bias'), stride, padding, dilation, groups)
    if norm == 'BN':
        y = F.batch_norm(y, P['bn.running_mean'].clone(), P['bn.running_var'].clone(), P['bn.weight'], P['bn.bias'], train, 0.1, eps)
    elif norm == 'GN':
        y = F.group_norm(y, num_groups, P['gn.weight'], P['gn.bias'], eps)
    if act == 'ReLU':
        y = F.relu(y)
    elif act == 'LeakyReLU':
        y = F.leaky_relu(y, slope)
    elif act == 'GELU':
        y = F.gelu(y)
    return y


@pytest.mark.parametrize('norm', [None, 'BN', 'GN'])
@pytest.mark.parametrize('act', [None, 'ReLU', 'LeakyReLU'])
@pytest.mark.parametrize('geom', [(6, 8, 1, 1, 0, 1, 1), (8, 12, 3, 1, 1, 1, 1), (8, 8, 3, 2, 2, 2, 4)])
def test_conv_module_is_mmcv_documented_behaviour(norm, act, geom):
    cin, cout, k, stride, pad, dil, groups = geom
    torch.manual_seed(3)
    norm_cfg = None if norm is None else (dict(type='BN', requires_grad=True) if norm == 'BN' else dict(type='GN', num_groups=4))
    act_cfg = None if act is None else dict(type=act)
    m = bricks.ConvModule(cin, cout, k, stride=stride, padding=pad, dilation=dil, groups=groups, norm_cfg=norm_cfg, act_cfg=act_cfg)
    # structure: bias iff no norm; the norm's registered name; in-place activation; key set
    keys = set(m.state_dict().keys())
    want = {'conv.weight'} | ({'conv.bias'} if norm is None else set())
    if norm == 'BN':
        want |= {'bn.weight', 'bn.bias', 'bn.running_mean', 'bn.running_var', 'bn.num_batches_tracked'}
    if norm == 'GN':
        want |= {'gn.weight', 'gn.bias'}
    assert keys == want, (keys, want)
    assert (m.conv.bias is None) == (norm is not None)
    if act is not None:
        assert m.activate.inplace is True
    if norm == 'BN':
        assert m.bn.num_features == cout and m.bn.eps == 1e-5 and m.bn.momentum == 0.1
    # init: kaiming normal, mode fan_out -> std = gain / sqrt(cout * k * k / groups ... fan_out of the conv weight), zero bias, unit norm
    fan_out = cout * k * k // groups
    gain = math.sqrt(2.0 / (1 + 0.01 ** 2)) if act == 'LeakyReLU' else math.sqrt(2.0)
    big = bricks.ConvModule(64, 256, 3, padding=1, norm_cfg=norm_cfg if norm != 'GN' else dict(type='GN', num_groups=4), act_cfg=act_cfg)
    std = big.conv.weight.detach().std().item()
    assert abs(std - gain / math.sqrt(256 * 9)) <= 0.03 * gain / math.sqrt(256 * 9), (std, gain / math.sqrt(256 * 9))
    if big.conv.bias is not None:
        assert torch.count_nonzero(big.conv.bias) == 0
    if norm is not None:
        assert bool((big.norm.weight == 1).all()) and bool((big.norm.bias == 0).all())
    # forward, training and eval mode, on perturbed parameters
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.3 * torch.randn_like(p))
        if norm == 'BN':
            m.bn.running_mean.normal_()
            m.bn.running_var.uniform_(0.5, 2.0)
    x = torch.randn(3, cin, 9, 7)
    for train in (True, False):
        m.train(train)
        P = {k_: v.detach().clone() for k_, v in m.state_dict().items()}
        want_y = _doc_conv_module(x, P, stride, pad, dil, groups, norm, act, train, num_groups=4)
        got = m(x.clone())
        assert got.shape == want_y.shape and torch.allclose(got, want_y, rtol=1e-5, atol=1e-6), (norm, act, train, (got - want_y).abs().max().item())
    # gradients flow to exactly the parameters the documented module has
    m.train()
    m(x).sum().backward()
    assert all(p.grad is not None for p in m.parameters())


def test_build_norm_layer_documented_names_and_flags():
    for cfg, name, cls in ((dict(type='BN'), 'bn', torch.nn.BatchNorm2d), (dict(type='LN'), 'ln', torch.nn.LayerNorm),
                           (dict(type='GN', num_groups=2), 'gn', torch.nn.GroupNorm)):
        n, layer = bricks.build_norm_layer(cfg, 8)
        assert n == name and isinstance(layer, cls)
        assert all(p.requires_grad for p in layer.parameters())
    n, layer = bricks.build_norm_layer(dict(type='BN', requires_grad=False), 8)
    assert not any(p.requires_grad for p in layer.parameters())
    n, layer = bricks.build_norm_layer(dict(type='LN'), 8, postfix=1)
    assert n == 'ln1' and layer.eps == 1e-5


@pytest.mark.parametrize('num_fcs', [2, 3])
@pytest.mark.parametrize('add_identity', [True, False])
def test_ffn_is_mmcv_documented_behaviour(num_fcs, add_identity):
    torch.manual_seed(5)
    m = bricks.FFN(embed_dims=12, feedforward_channels=40, num_fcs=num_fcs, act_cfg=dict(type='GELU'), ffn_drop=0.0, dropout_layer=None,
                   add_identity=add_identity).eval()
    keys = set(m.state_dict().keys())
    want = {f'layers.{i}.0.{p}' for i in range(num_fcs - 1) for p in ('weight', 'bias')} | {f'layers.{num_fcs - 1}.{p}' for p in ('weight', 'bias')}
    assert keys == want, (keys, want)
    P = m.state_dict()
    x, ident = torch.randn(2, 5, 12), torch.randn(2, 5, 12)
    h = x
    for i in range(num_fcs - 1):
        h = F.gelu(F.linear(h, P[f'layers.{i}.0.weight'], P[f'layers.{i}.0.bias']))
    out = F.linear(h, P[f'layers.{num_fcs - 1}.weight'], P[f'layers.{num_fcs - 1}.bias'])
    assert torch.allclose(m(x), (x + out) if add_identity else out, rtol=1e-5, atol=1e-6)
    assert torch.allclose(m(x, identity=ident), (ident + out) if add_identity else out, rtol=1e-5, atol=1e-6)
    # the Swin blocks' configuration (depthformer_swin.py:451-459): dropout_layer = DropPath; in eval it is the identity
    d = bricks.FFN(embed_dims=12, feedforward_channels=40, num_fcs=2, act_cfg=dict(type='GELU'), ffn_drop=0.0,
                   dropout_layer=dict(type='DropPath', drop_prob=0.4), add_identity=True)
    d.load_state_dict({k: v for k, v in bricks.FFN(embed_dims=12, feedforward_channels=40, act_cfg=dict(type='GELU')).state_dict().items()})
    d.eval()
    Pd = d.state_dict()
    o = F.linear(F.gelu(F.linear(x, Pd['layers.0.0.weight'], Pd['layers.0.0.bias'])), Pd['layers.1.weight'], Pd['layers.1.bias'])
    assert torch.allclose(d(x), x + o, rtol=1e-5, atol=1e-6)


def test_drop_path_is_per_sample_bernoulli_scaled_by_keep():
    torch.manual_seed(0)
    dp = bricks.DropPath(0.25).train()
    x = torch.ones(4000, 3, 2)
    y = dp(x)
    per_sample = y.view(4000, -1)
    assert bool(((per_sample == 0).all(1) | (per_sample == 1 / 0.75).all(1)).all())          # one draw per sample, kept samples scaled by 1 / keep
    assert abs((per_sample[:, 0] != 0).float().mean().item() - 0.75) < 0.03
    assert torch.equal(dp.eval()(x), x) and torch.equal(bricks.DropPath(0.0).train()(x), x)
