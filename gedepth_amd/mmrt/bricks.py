"""Module bricks the reference takes from mmcv 1.3.13 (not vendored there).

Each brick restates the mmcv semantics listed in SURVEY.md Appendix A so that
state-dict key names, default hyper-parameters and initialisation of the
reference's modules are reproduced without mmcv:
``BaseModule/ModuleList/Sequential`` (depth/models/backbones/depthformer_swin.py:12),
``ConvModule`` (necks/hahi.py:3, decode_heads/densedepth_head.py:4),
``build_norm_layer`` (depthformer_swin.py:8,437,1040), ``FFN``/``build_dropout``
(depthformer_swin.py:9,283,451-459), weight-init helpers.
"""
import math
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from .registry import Registry, build_from_cfg

MODELS = Registry('model')
ATTENTION = Registry('attention')
POSITIONAL_ENCODING = Registry('position encoding')
DROPOUT_LAYERS = Registry('drop out layers')


# ----------------------------------------------------------------------------------- init
def constant_init(module, val, bias=0):
    if hasattr(module, 'weight') and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    assert distribution in ['uniform', 'normal']
    if hasattr(module, 'weight') and module.weight is not None:
        if distribution == 'uniform':
            nn.init.xavier_uniform_(module.weight, gain=gain)
        else:
            nn.init.xavier_normal_(module.weight, gain=gain)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def kaiming_init(module, a=0, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
    assert distribution in ['uniform', 'normal']
    if hasattr(module, 'weight') and module.weight is not None:
        if distribution == 'uniform':
            nn.init.kaiming_uniform_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
        else:
            nn.init.kaiming_normal_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def trunc_normal_init(module, mean=0., std=1., a=-2., b=2., bias=0.):
    """Accepts a module (weight/bias) or a bare parameter, as the reference
    calls it both ways (depthformer_swin.py:182,1049-1053)."""
    if isinstance(module, torch.Tensor):
        nn.init.trunc_normal_(module, mean, std, a, b)
        return
    if hasattr(module, 'weight') and module.weight is not None:
        nn.init.trunc_normal_(module.weight, mean, std, a, b)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


# ----------------------------------------------------------------------------------- base
class BaseModule(nn.Module):
    """nn.Module + ``init_cfg`` + one-shot recursive ``init_weights``."""

    def __init__(self, init_cfg=None):
        super().__init__()
        self._is_init = False
        self.init_cfg = init_cfg

    @property
    def is_init(self):
        return self._is_init

    def init_weights(self):
        if self._is_init:
            warnings.warn(f'init_weights of {self.__class__.__name__} has been called more than once.')
            return
        for m in self.children():
            if hasattr(m, 'init_weights'):
                m.init_weights()
        self._is_init = True


class Sequential(BaseModule, nn.Sequential):

    def __init__(self, *args, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.Sequential.__init__(self, *args)


class ModuleList(BaseModule, nn.ModuleList):

    def __init__(self, modules=None, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.ModuleList.__init__(self, modules)


# ----------------------------------------------------------------------------------- layers
def _split_k(tokens):
    """Number of K-chunks for the weight gradient of a Linear over ``tokens`` rows (0: leave it to the library)."""
    # measured (tools/ubench/wgrad_splitk2.py, MI355X): 12320 tokens x (1536 x 384) 70 us plain -> 44 us with 8 - 28 chunks, (384 x 384) 46 -> 29;
    # 3080 tokens: the plain GEMM is fastest (27 - 39 us); >= 49280 tokens: 2 - 3x from 32 - 44 chunks
    if tokens < 8192:
        return 0
    for s in range(64, 3, -1):                    # the largest divisor <= 64 that leaves chunks of >= 1024 tokens
        if tokens % s == 0 and tokens // s >= 1024:
            return s
    return 0


def _linear_fwd(xc, wc, weight, bias, dt):
    """y = xc wc^T + bias for the token Linears: ge_gemm_nt (csrc/gemm.hip) for the shapes where it beats the tuned library solution
    (kernels.GEMM_OWN), the library GEMM with its bias epilogue otherwise.  `bias` is the fp32 master parameter: the own kernel adds it in
    fp32 before the single rounding, the library epilogue adds the bf16 shadow."""
    from .. import kernels
    from .optim import lowp
    K = xc.shape[-1]
    M = xc.numel() // K
    # ge_gemm_nt wants 16-byte-aligned operands (a contiguous view at an odd storage offset, a bias slice of an arena built with align < 4,
    # would be refused with GE_ERR_UNSUPPORTED in the middle of training): such calls take the library
    if (dt == torch.bfloat16 and xc.is_cuda and xc.is_contiguous() and wc.is_contiguous() and kernels.gemm_own(M, K, wc.shape[0])
            and (bias is None or (bias.dtype == torch.float32 and bias.is_contiguous() and bias.data_ptr() % 16 == 0))
            and xc.data_ptr() % 16 == 0 and wc.data_ptr() % 16 == 0):
        return kernels.gemm_nt(xc.reshape(M, K), wc, None if bias is None else bias.detach()).view(*xc.shape[:-1], wc.shape[0])
    return F.linear(xc, wc, None if bias is None else lowp(bias, dt))


def _linear_dx(dy2, wc):
    """dx = dy2 wc (the input gradient of a token Linear): ge_gemm_nt on the transposed weight where it wins, else the library GEMM."""
    from .. import kernels
    M, N = dy2.shape
    K = wc.shape[1]
    if dy2.dtype == torch.bfloat16 and dy2.is_contiguous() and dy2.data_ptr() % 16 == 0 and kernels.gemm_own(M, N, K):
        return kernels.gemm_nt(dy2, wc.t().contiguous(), None)          # (K, N) weight image: <= 1.2 MB, one small transpose kernel
    return dy2 @ wc


def _sum_partials(part, weight, w_dtype):
    """fp32 sum of the split-K partial products — written straight into the parameter's slice of the gradient arena when this is the
    parameter's first gradient of the step (optim.grad_target), else into a new tensor."""
    if weight is not None and w_dtype == torch.float32:
        from .optim import grad_target
        out = grad_target(weight)
        if out is not None and out.shape == part.shape[1:]:
            return torch.sum(part, 0, dtype=torch.float32, out=out)
    return part.sum(0, dtype=torch.float32).to(w_dtype)


def _plain_dw(dy2, x2, weight, w_dtype):
    """dW = dY^T X as ONE library GEMM (token counts too small for the split-K form).  bf16 operands with an fp32 master weight: the GEMM writes
    its fp32 accumulators straight into the parameter's slice of the gradient arena (``torch.mm(..., out_dtype=float32, out=...)``) — no bf16
    rounding of the result and no widening copy (Swin-L at 2 images per GPU: 96 copy launches per step)."""
    if dy2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16 and w_dtype == torch.float32:
        from .. import kernels
        if 'mm_fp32_out' not in kernels.DISABLED:
            from .optim import grad_target
            out = grad_target(weight) if weight is not None else None
            if out is not None and tuple(out.shape) == (dy2.shape[1], x2.shape[1]):
                return torch.mm(dy2.t(), x2, out_dtype=torch.float32, out=out)
            return torch.mm(dy2.t(), x2, out_dtype=torch.float32)
    return _finished_grad(dy2.t() @ x2, weight, w_dtype)


def _finished_grad(dw, weight, w_dtype):
    from .optim import grad_into_arena
    return grad_into_arena(weight, dw, w_dtype) if weight is not None else dw.to(w_dtype)


class _LinearTokens(torch.autograd.Function):
    """``F.linear`` whose weight gradient is computed split-K.  dW = dY^T X has an output of at most 768 x 768 and a
    reduction over 5e4 - 8e5 tokens: the libraries run it as one GEMM on a handful of output tiles (hipBLASLt / rocBLAS,
    best tuned solution: 18 - 365 TFLOP/s on MI355X); as a batched GEMM over S token chunks plus an fp32 sum of the S
    partial products it fills the chip (2 - 4x faster, tools/ubench/wgrad_splitk.py) and the result is accumulated in fp32
    instead of being rounded to bf16 first.  Forward and input gradient are the library GEMMs unchanged."""

    @staticmethod
    def forward(ctx, x, weight, bias, splits):
        dt = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled() else x.dtype
        with torch.autocast('cuda', enabled=False):
            from .optim import lowp
            xc, wc = x.to(dt), lowp(weight, dt)
            y = _linear_fwd(xc, wc, weight, bias, dt)
        ctx.save_for_backward(xc, wc)
        ctx.meta = (splits, x.dtype, weight.dtype, None if bias is None else bias.dtype)
        ctx.weight_ref = weight if isinstance(weight, nn.Parameter) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, wc = ctx.saved_tensors
        splits, x_dtype, w_dtype, b_dtype = ctx.meta
        K = xc.numel() // xc.shape[-1]
        with torch.autocast('cuda', enabled=False):
            dy2 = dy.to(wc.dtype).reshape(K, -1)
            x2 = xc.reshape(K, -1)
            dx = dw = db = None
            if not dy2.is_contiguous():
                dy2 = dy2.contiguous()
            if ctx.needs_input_grad[0]:
                dx = _linear_dx(dy2, wc).reshape(xc.shape).to(x_dtype)
            if ctx.needs_input_grad[1]:
                if splits:
                    M, N = dy2.shape[1], x2.shape[1]
                    part = torch.bmm(dy2.view(splits, K // splits, M).transpose(1, 2), x2.view(splits, K // splits, N))
                    dw = _sum_partials(part, ctx.weight_ref, w_dtype)
                else:
                    dw = _plain_dw(dy2, x2, ctx.weight_ref, w_dtype)
            if b_dtype is not None and ctx.needs_input_grad[2]:
                if dy2.shape[1] % (4 if dy2.dtype == torch.float32 else 8) == 0:      # 16-byte channel vectors: one streaming pass
                    from .. import kernels
                    db = kernels.colsum(dy2 if dy2.is_contiguous() else dy2.contiguous()).to(b_dtype)
                else:                                                                 # e.g. the 2-wide reference-point Linear of HAHI
                    db = dy2.sum(0, dtype=torch.float32).to(b_dtype)
        return dx, dw, db, None


class _LinearBiasGelu(torch.autograd.Function):
    """``gelu(F.linear(x, W, b))`` (the first half of mmcv's FFN, exact erf GELU) as: library GEMM with its bias epilogue -> one HIP
    pass ``gelu(y)`` (csrc/nhwc.hip ge_bias_gelu_fwd); backward: ONE HIP pass that multiplies by GELU' and accumulates the bias
    gradient (ge_bias_gelu_bwd: no separate column-sum read of d_y), then the library input-gradient GEMM and the split-K weight
    gradient of ``_LinearTokens``.  Saves x, W and y (as autograd did)."""

    @staticmethod
    def forward(ctx, x, weight, bias, splits):
        from .. import kernels
        dt = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled() else x.dtype
        with torch.autocast('cuda', enabled=False):
            from .optim import lowp
            xc, wc = x.to(dt), lowp(weight, dt)
            # the bias rides in the library GEMM's epilogue (same addmm problem as before, i.e. the SAME tuned hipBLASLt solution of
            # gedepth_amd/tuning/tunableop_gfx950.csv: the bias-free mm variant of these shapes is not in the table and measured
            # 0.4 ms/step slower, tools/ubench/ab_bench.sh), so the kernels run with bias = NULL on y = x W^T + b
            y0 = _linear_fwd(xc, wc, weight, bias, dt)
            g = kernels.bias_gelu_fwd(y0, None)
        ctx.save_for_backward(xc, wc, y0)
        ctx.meta = (splits, x.dtype, weight.dtype, bias.dtype)
        ctx.weight_ref = weight if isinstance(weight, nn.Parameter) else None
        return g

    @staticmethod
    def backward(ctx, dg):
        from .. import kernels
        xc, wc, y0 = ctx.saved_tensors
        splits, x_dtype, w_dtype, b_dtype = ctx.meta
        K = xc.numel() // xc.shape[-1]
        with torch.autocast('cuda', enabled=False):
            dg2 = dg.to(wc.dtype).reshape(K, -1)
            if not dg2.is_contiguous():
                dg2 = dg2.contiguous()
            dy2, db = kernels.bias_gelu_bwd(dg2, y0.reshape(K, -1), None)
            x2 = xc.reshape(K, -1)
            dx = dw = None
            if ctx.needs_input_grad[0]:
                dx = _linear_dx(dy2, wc).reshape(xc.shape).to(x_dtype)
            if ctx.needs_input_grad[1]:
                if splits:
                    M, N = dy2.shape[1], x2.shape[1]
                    part = torch.bmm(dy2.view(splits, K // splits, M).transpose(1, 2), x2.view(splits, K // splits, N))
                    dw = _sum_partials(part, ctx.weight_ref, w_dtype)
                else:
                    dw = _plain_dw(dy2, x2, ctx.weight_ref, w_dtype)
        return dx, dw, (db.to(b_dtype) if ctx.needs_input_grad[2] else None), None


def linear_bias_gelu(x, weight, bias):
    """``F.gelu(F.linear(x, weight, bias))`` (exact GELU) with the bias + GELU epilogue kernels; eager composition off the GPU /
    outside training / for channel counts that are not whole 16-byte vectors."""
    C = weight.shape[0]
    ok = (x.is_cuda and torch.is_grad_enabled() and weight.requires_grad and bias is not None and x.dim() >= 2 and x.numel() > 0
          and x.dtype in (torch.float32, torch.bfloat16) and C % 8 == 0)
    if ok:
        from .. import kernels
        ok = 'bias_gelu' not in kernels.DISABLED
    if not ok:
        return F.gelu(linear_tokens(x, weight, bias))
    return _LinearBiasGelu.apply(x, weight, bias, _split_k(x.numel() // x.shape[-1]))


def linear_tokens(x, weight, bias=None):
    """``F.linear`` for token matrices: split-K weight gradient when the token count is large (see _LinearTokens)."""
    if not (x.is_cuda and torch.is_grad_enabled() and weight.requires_grad) or x.dtype not in (torch.float32, torch.bfloat16) or x.dim() < 2 or x.numel() == 0:
        return F.linear(x, weight, bias)
    # every training-mode token Linear takes this route (splits == 0: plain library dW): weights come from the optimizer's bf16
    # shadow arena instead of a cast kernel each, and the bias gradient is one column-sum kernel
    return _LinearTokens.apply(x, weight, bias, _split_k(x.numel() // x.shape[-1]))


class Linear(nn.Linear):
    """``nn.Linear`` (same parameters / state-dict keys) over token matrices, see ``linear_tokens``."""

    def forward(self, x):
        return linear_tokens(x, self.weight, self.bias)


class LayerNorm(nn.LayerNorm):
    """``nn.LayerNorm`` (same parameters / state-dict keys) running as one mixed-precision HIP kernel on MI355X: bf16 or
    fp32 tokens in, fp32 statistics, and — under autocast — bf16 out for the Linear that follows, instead of ATen's
    copy-to-fp32 / fp32 LayerNorm / copy-to-bf16 (gedepth_amd/csrc/norm.hip).  ``autocast_out = False`` keeps the fp32 output
    autocast would give (the patch-embed norm, whose output is the fp32 residual stream of stage 0)."""
    autocast_out = True

    def _hip_ok(self, x):
        C = x.shape[-1]
        return (x.is_cuda and self.elementwise_affine and len(self.normalized_shape) == 1 and C % 4 == 0 and C <= 3072
                and x.dtype in (torch.float32, torch.bfloat16))

    def _out_dtype(self, x):
        if torch.is_autocast_enabled():
            return torch.get_autocast_dtype('cuda') if self.autocast_out else torch.float32
        return x.dtype

    def forward_with_skip(self, x):
        """-> (self(x), x') where x' is x for the skip connection of a pre-norm block: in training on MI355X the two gradients of x meet
        inside the LayerNorm backward kernel (kernels.layer_norm_res); otherwise x' is x."""
        if self._hip_ok(x) and torch.is_grad_enabled() and x.requires_grad:
            from .. import kernels
            if 'ln_res' not in kernels.DISABLED:
                return kernels.layer_norm_res(x, self.weight, self.bias, self.eps, self._out_dtype(x))
        return self(x), x

    def forward(self, x):
        C = x.shape[-1]
        if self._hip_ok(x):
            from .. import kernels
            return kernels.layer_norm(x, self.weight, self.bias, self.eps, self._out_dtype(x))
        if x.is_cuda:
            from .. import kernels
            kernels.note_fallback('LayerNorm', f'C={C} dtype={x.dtype}')
        return super().forward(x)


_NORM = {
    'BN': ('bn', nn.BatchNorm2d), 'BN1d': ('bn', nn.BatchNorm1d), 'BN2d': ('bn', nn.BatchNorm2d),
    'SyncBN': ('bn', nn.SyncBatchNorm), 'GN': ('gn', nn.GroupNorm), 'LN': ('ln', LayerNorm),
    'IN': ('in', nn.InstanceNorm2d),
}
_ACT = {
    'ReLU': nn.ReLU, 'LeakyReLU': nn.LeakyReLU, 'PReLU': nn.PReLU, 'ReLU6': nn.ReLU6, 'ELU': nn.ELU,
    'Sigmoid': nn.Sigmoid, 'Tanh': nn.Tanh, 'GELU': nn.GELU,
}
_CONV = {'Conv1d': nn.Conv1d, 'Conv2d': nn.Conv2d, 'Conv3d': nn.Conv3d, 'Conv': nn.Conv2d}


def build_norm_layer(cfg, num_features, postfix=''):
    if not isinstance(cfg, dict) or 'type' not in cfg:
        raise KeyError('the cfg dict must contain the key "type"')
    cfg_ = dict(cfg)
    layer_type = cfg_.pop('type')
    if layer_type not in _NORM:
        raise KeyError(f'Unrecognized norm type {layer_type}')
    abbr, norm_layer = _NORM[layer_type]
    name = abbr + str(postfix)
    requires_grad = cfg_.pop('requires_grad', True)
    cfg_.setdefault('eps', 1e-5)
    if layer_type == 'GN':
        assert 'num_groups' in cfg_
        layer = norm_layer(num_channels=num_features, **cfg_)
    else:
        layer = norm_layer(num_features, **cfg_)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return name, layer


def build_activation_layer(cfg):
    cfg_ = dict(cfg)
    t = cfg_.pop('type')
    if t not in _ACT:
        raise KeyError(f'Unrecognized activation type {t}')
    return _ACT[t](**cfg_)


def build_conv_layer(cfg, *args, **kwargs):
    cfg_ = dict(type='Conv2d') if cfg is None else dict(cfg)
    t = cfg_.pop('type')
    if t not in _CONV:
        raise KeyError(f'Unrecognized conv type {t}')
    return _CONV[t](*args, **kwargs, **cfg_)


class ConvModule(nn.Module):
    """conv -> norm -> act block; sub-module names ``conv``, ``bn``/``ln``…, ``activate``."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True,
                 order=('conv', 'norm', 'act')):
        super().__init__()
        assert order == ('conv', 'norm', 'act'), 'only the conv-norm-act order is on the GEDepth path'
        self.conv_cfg, self.norm_cfg, self.act_cfg = conv_cfg, norm_cfg, act_cfg
        self.inplace = inplace
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        self.with_bias = bias
        self.conv = build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, stride=stride,
                                     padding=padding, dilation=dilation, groups=groups, bias=bias)
        self.in_channels, self.out_channels = in_channels, out_channels
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        else:
            self.norm_name = None
        if self.with_activation:
            act_cfg_ = dict(act_cfg)
            if act_cfg_['type'] not in ['Tanh', 'PReLU', 'Sigmoid', 'GELU']:
                act_cfg_.setdefault('inplace', inplace)
            self.activate = build_activation_layer(act_cfg_)
        self.init_weights()

    @property
    def norm(self):
        return getattr(self, self.norm_name) if self.norm_name else None

    def init_weights(self):
        if self.with_activation and self.act_cfg['type'] == 'LeakyReLU':
            nonlinearity, a = 'leaky_relu', self.act_cfg.get('negative_slope', 0.01)
        else:
            nonlinearity, a = 'relu', 0
        kaiming_init(self.conv, a=a, nonlinearity=nonlinearity)
        if self.with_norm:
            constant_init(self.norm, 1, bias=0)

    def _fused_slope(self):
        """leaky slope when conv bias + activation can run as one in-place HIP pass over the conv output, else None."""
        if self.with_norm or self.conv.bias is None or type(self.conv) is not nn.Conv2d:
            return None
        if not self.with_activation:
            return 1.0
        if isinstance(self.activate, nn.LeakyReLU):
            return float(self.activate.negative_slope)
        if isinstance(self.activate, nn.ReLU):
            return 0.0
        return None

    def _fused_bn_slope(self):
        """leaky slope when norm + activation can run as the fused training-mode BatchNorm kernels, else None."""
        bn = self.norm
        if (type(bn) is not nn.BatchNorm2d or not bn.training or not bn.affine or not bn.track_running_stats
                or bn.momentum is None or self.conv.bias is not None):
            return None
        if not self.with_activation:
            return 1.0
        if isinstance(self.activate, nn.ReLU):
            return 0.0
        if isinstance(self.activate, nn.LeakyReLU):
            return float(self.activate.negative_slope)
        return None

    def forward(self, x):
        slope = self._fused_slope() if x.is_cuda else None
        if slope is not None:
            return conv_bias_act(self.conv, x, slope)
        if x.is_cuda and self.conv.bias is None:
            from .. import kernels
            if kernels.conv3x3_ok(self.conv, x):              # hand-written MFMA implicit GEMM (forward + data gradient)
                x = kernels.conv3x3(self.conv, x)
            else:
                x = kernels.conv_lib(self.conv, x)            # library convolution, weight from the bf16 shadow arena
        else:
            x = self.conv(x)
        if x.is_cuda and self.with_norm and x.dtype in (torch.float32, torch.bfloat16):
            slope = self._fused_bn_slope()
            if slope is not None:                       # training-mode BatchNorm2d (+ ReLU): the fused HIP kernels
                from .. import kernels
                return kernels.bn_act(x, self.norm, slope)
        if x.is_cuda and (self.with_norm or self.with_activation) and self.training:
            from .. import kernels
            kernels.note_fallback('ConvModule', f'norm={type(self.norm).__name__ if self.with_norm else None} act={self.act_cfg}')
        if self.with_norm:
            x = self.norm(x)
        if self.with_activation:
            x = self.activate(x)
        return x


def conv_bias_act(conv, x, slope=1.0):
    """``act(conv(x) + bias)`` with the bias add and the (leaky-)ReLU fused into one in-place HIP kernel over the
    bias-free convolution output (MIOpen has no fused epilogue for these shapes; ATen would launch a broadcast add
    and an activation kernel, and two more for their gradients)."""
    from .. import kernels
    if kernels.conv3x3_ok(conv, x):                            # MFMA 3x3 convolution with the bias / activation in its epilogue
        return kernels.conv3x3(conv, x, conv.bias, act=slope != 1.0, slope=slope)
    if slope == 1.0 and kernels.conv3x3_c1_ok(conv, x):        # one output channel: streaming kernel, one backward pass for dx / dw / db
        return kernels.conv3x3_c1(conv, x)
    y = kernels.conv_lib(conv, x)
    if not (y.is_contiguous() or kernels._cl_ok(y)):           # dense NCHW or (vector-sized) channels-last both run in place
        y = y.contiguous()
    return kernels.bias_act_(y, conv.bias, slope)


def drop_path(x, drop_prob=0., training=False):
    if drop_prob == 0. or not training:
        return x
    keep_prob = 1 - drop_prob
    shape = (x.shape[0],) + (1,) * (x.ndim - 1)
    random_tensor = keep_prob + torch.rand(shape, dtype=x.dtype, device=x.device)
    return x.div(keep_prob) * random_tensor.floor()


@DROPOUT_LAYERS.register_module()
class DropPath(nn.Module):
    """Per-sample stochastic depth."""

    def __init__(self, drop_prob=0.1):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        return drop_path(x, self.drop_prob, self.training)

    def residual(self, identity, branch):
        """``identity + self(branch)`` — on MI355X one mixed-precision pass (kernels.residual_drop_path) instead of ATen's
        divide / multiply / add over the token tensor.  Same random draw as ``drop_path`` (one uniform per sample)."""
        if (not identity.is_cuda or self.drop_prob == 0. or not self.training or identity.shape != branch.shape
                or (identity.dtype, branch.dtype) not in _RESIDUAL_DTYPES):
            if identity.is_cuda and self.training and self.drop_prob > 0.:
                from .. import kernels
                kernels.note_fallback('DropPath.residual', f'{identity.dtype}+{branch.dtype} {tuple(identity.shape)} vs {tuple(branch.shape)}')
            return identity + self(branch)
        from .. import kernels
        return kernels.residual_drop_path(identity, branch, _DROP_PATH_BANK.scale(self, branch.shape[0], branch.dtype, branch.device))


class _DropPathBank:
    """Per-sample keep / (1 - p) scales of ALL DropPath layers of a step from ONE uniform draw: the 22 - 46 stochastic-depth layers of a
    Swin encoder each cost five tiny launches (rand, add, floor, divide, cast) per step; here layers register on first use and the bank
    refills every row at once — ``floor(keep + U(L, B)) / keep`` — whenever a layer asks for a row it has already consumed (or the
    generator was re-seeded since: a run after ``torch.manual_seed`` does not see rows drawn before it).  Same law as
    ``drop_path`` (one uniform per sample and layer; drawn in fp32, so the keep probability is exact to 2^-24)."""

    def __init__(self):
        # layers are held by WEAK reference (a bank entry must not keep a discarded model alive); the keep probabilities are re-read
        # from the live layers at every refill, so a later change of ``layer.drop_prob`` takes effect.  A recomputed forward
        # (activation checkpointing) would draw a fresh row — ``with_cp`` is refused by the Swin blocks for that reason.
        self.layers, self.rows, self.used, self.key, self.keep = [], None, [], None, None
        self.probs = None
        self.seed, self.offset = None, 0

    def scale(self, layer, batch, dtype, device):
        idx = getattr(layer, '_bank_index', None)
        key = (batch, device)
        if idx is None or idx >= len(self.layers) or self.layers[idx]() is not layer:
            import weakref
            self.layers = [r for r in self.layers if r() is not None]         # drop the entries of models that are gone
            for i, r in enumerate(self.layers):
                r()._bank_index = i
            idx = layer._bank_index = len(self.layers)
            self.layers.append(weakref.ref(layer))
            self.rows = self.keep = None
        gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()] if device.type == 'cuda' else None
        # a step being captured in a hipGraph (mmrt/graph.py) may not query the generator; its refill is captured with torch's graph-safe
        # Philox offsets, so every replay draws fresh rows
        capturing = gen is not None and torch.cuda.is_current_stream_capturing()
        reseeded = gen is not None and not capturing and (gen.initial_seed(), gen.get_offset() >= self.offset) != (self.seed, True)     # torch.manual_seed since the draw
        if self.rows is None or self.key != key or self.used[idx] or reseeded:
            probs = [1.0 - (r().drop_prob if r() is not None else 0.0) for r in self.layers]
            if self.keep is None or self.keep.device != device or probs != self.probs:
                self.keep, self.probs = torch.tensor(probs, dtype=torch.float32).view(-1, 1).to(device), probs
            u = torch.rand((len(self.layers), batch), dtype=torch.float32, device=device)
            self.rows = (self.keep + u).floor() / self.keep
            self.used = [False] * len(self.layers)
            self.key = key
            if gen is not None and not capturing:
                self.seed, self.offset = gen.initial_seed(), gen.get_offset()
        self.used[idx] = True
        return self.rows[idx]


_DROP_PATH_BANK = _DropPathBank()


_RESIDUAL_DTYPES = {(torch.float32, torch.float32), (torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16)}


@DROPOUT_LAYERS.register_module()
class Dropout(nn.Dropout):

    def __init__(self, drop_prob=0.5, inplace=False):
        super().__init__(p=drop_prob, inplace=inplace)


def build_dropout(cfg, default_args=None):
    return build_from_cfg(cfg, DROPOUT_LAYERS, default_args)


def build_positional_encoding(cfg, default_args=None):
    return build_from_cfg(cfg, POSITIONAL_ENCODING, default_args)


class FFN(BaseModule):
    """Linear(C,4C)-act-drop-Linear(4C,C)-drop with identity add; keys
    ``layers.0.0.*`` / ``layers.1.*`` (consistent with models/utils/ckpt_convert.py:29-32)."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=dict(type='ReLU', inplace=True), ffn_drop=0., dropout_layer=None,
                 add_identity=True, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        assert num_fcs >= 2
        self.embed_dims, self.feedforward_channels, self.num_fcs = embed_dims, feedforward_channels, num_fcs
        self.act_cfg = act_cfg
        self.activate = build_activation_layer(act_cfg)
        layers, in_channels = [], embed_dims
        for _ in range(num_fcs - 1):
            layers.append(Sequential(Linear(in_channels, feedforward_channels), self.activate,
                                     nn.Dropout(ffn_drop)))
            in_channels = feedforward_channels
        layers.append(Linear(feedforward_channels, embed_dims))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = Sequential(*layers)
        self.dropout_layer = build_dropout(dropout_layer) if dropout_layer else nn.Identity()
        self.add_identity = add_identity

    def _fused_first(self):
        """Linear -> exact GELU -> Dropout(0 or eval): the pattern of every Swin block (depthformer_swin.py:451-459)."""
        first = self.layers[0]
        return (self.num_fcs == 2 and isinstance(first, nn.Sequential) and len(first) == 3 and type(first[0]) is Linear
                and type(first[1]) is nn.GELU and getattr(first[1], 'approximate', 'none') == 'none'
                and isinstance(first[2], nn.Dropout) and (first[2].p == 0 or not self.training))

    def forward(self, x, identity=None):
        if x.is_cuda and self._fused_first():
            lin = self.layers[0][0]
            h = linear_bias_gelu(x, lin.weight, lin.bias)          # bias + GELU epilogue kernels (kernels.bias_gelu_*)
            out = self.layers[2](self.layers[1](h))
        else:
            out = self.layers(x)
        if not self.add_identity:
            return self.dropout_layer(out)
        if identity is None:
            identity = x
        if hasattr(self.dropout_layer, 'residual'):
            return self.dropout_layer.residual(identity, out)
        return identity + self.dropout_layer(out)


def msda_offset_bias(num_heads, num_levels, num_points):
    """Initial ``sampling_offsets.bias`` of mmcv's MultiScaleDeformableAttention
    (unit directions / max-abs component, scaled by point index + 1)."""
    thetas = torch.arange(num_heads, dtype=torch.float32) * (2.0 * math.pi / num_heads)
    grid = torch.stack([thetas.cos(), thetas.sin()], -1)
    grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(num_heads, 1, 1, 2).repeat(
        1, num_levels, num_points, 1)
    for i in range(num_points):
        grid[:, :, i, :] *= i + 1
    return grid.view(-1)
