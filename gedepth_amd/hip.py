"""ctypes binding of libgedepth_hip.so (C ABI: include/gedepth_hip.h).

There is deliberately NO fallback: if the shared library is missing, or a tensor is not a
contiguous CUDA(HIP) tensor of the expected dtype, the call raises.  Build the library with
``gedepth_amd/csrc/build.sh`` (or ``python -c 'import __graft_entry__ as g; g.build()'``).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GE_LIB') or os.path.join(_HERE, 'csrc', 'libgedepth_hip.so')     # GE_LIB: a differently built library (A/B timing)
GE_F32, GE_BF16 = 0, 1

_c = ctypes
_vp, _i, _f, _l, _d, _sz, _u64 = _c.c_void_p, _c.c_int, _c.c_float, _c.c_long, _c.c_double, _c.c_size_t, _c.c_ulonglong

# name -> (restype, argtypes): mirrors include/gedepth_hip.h one to one
SIGNATURES = {
    'ge_abi_version': (_i, []),
    'ge_window_attn_fwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _vp]),
    'ge_window_attn_bwd_workspace': (_sz, [_i, _i, _i, _i]),
    'ge_window_attn_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _vp]),
    'ge_msda_fwd': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_msda_mode': (_i, [_i]),
    'ge_msda_bwd_workspace': (_sz, [_vp, _i, _i, _i, _i, _i, _i]),
    'ge_msda_bwd': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_msda_bwd_plan': (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_msda_bwd_timing': (_i, [_i]),
    'ge_msda_bwd_timing_read': (_i, [_i, _vp, _vp, _vp, _i]),
    'ge_msda_prep_fwd': (_i, [_vp, _l, _vp, _l, _vp, _l, _l, _l, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_msda_prep_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _l, _vp, _l, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_msda_raw_supported': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i]),
    'ge_msda_fwd_raw': (_i, [_vp, _vp, _vp, _i, _vp, _l, _vp, _l, _vp, _l, _l, _l, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_msda_bwd_raw': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _l, _vp, _l, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_msda_mm_supported': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i]),
    'ge_msda_fwd_mm': (_i, [_vp, _vp, _vp, _l, _vp, _l, _vp, _l, _l, _l, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_msda_bwd_lw_mm': (_i, [_vp, _vp, _vp, _l, _vp, _l, _vp, _l, _l, _l, _vp, _vp, _vp, _l, _vp, _l, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_msda_bwd_mm_workspace': (_sz, [_i, _i, _i, _i]),
    'ge_msda_bwd_mm_stats_offset': (_sz, [_i, _i, _i, _i]),
    'ge_msda_bwd_value_mm': (_i, [_vp, _vp, _l, _vp, _l, _vp, _l, _l, _l, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_msda_fwd_mm_part': (_i, [_vp, _vp, _vp, _l, _vp, _l, _vp, _l, _l, _l, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_msda_bwd_lw_mm_part': (_i, [_vp, _vp, _vp, _l, _vp, _l, _vp, _l, _l, _l, _vp, _vp, _vp, _l, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_msda_bwd_vs_workspace': (_sz, [_vp, _i, _i, _i, _i, _i, _i]),
    'ge_msda_bwd_vs_stats_offset': (_sz, [_vp, _i, _i, _i, _i, _i, _i]),
    'ge_msda_bwd_value_vs': (_i, [_vp, _vp, _l, _vp, _l, _vp, _l, _l, _l, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_msda_bwd_value_raw_levels': (_i, [_vp, _vp, _l, _vp, _l, _vp, _l, _l, _l, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_msda_bwd_value': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_msda_bwd_value_raw': (_i, [_vp, _vp, _l, _vp, _l, _vp, _l, _l, _l, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_msda_dref': (_i, [_vp, _l, _vp, _vp, _l, _i, _i, _i, _i, _vp]),
    'ge_rng_salt': (_i, [_vp]),
    'ge_tokens_from_map': (_i, [_vp, _l, _vp, _vp, _l, _i, _i, _l, _f, _u64, _i, _vp]),
    'ge_map_from_tokens': (_i, [_vp, _l, _vp, _l, _vp, _l, _i, _i, _l, _f, _u64, _i, _vp]),
    'ge_bilinear_fwd': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_bilinear_bwd': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_layernorm_fwd': (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _l, _i, _f, _vp]),
    'ge_layernorm_bwd': (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _vp]),
    'ge_layernorm_bwd_res': (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _vp]),
    'ge_layernorm_bwd_multi': (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _l, _i, _vp]),
    'ge_layernorm_fold': (_i, [_vp, _vp, _i, _i, _vp]),
    'ge_residual_scale_add': (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _l, _vp]),
    'ge_scale_rows': (_i, [_vp, _i, _vp, _vp, _i, _i, _l, _vp]),
    'ge_bn_workspace': (_sz, [_i]),
    'ge_bn_act_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _l, _f, _f, _f, _i, _vp]),
    'ge_bn_act_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _l, _f, _i, _vp]),
    'ge_bias_act_fwd': (_i, [_vp, _vp, _i, _i, _l, _f, _i, _vp]),
    'ge_bias_act_bwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _l, _f, _i, _vp]),
    'ge_ground_embed_fwd': (_i, [_vp, _vp, _vp, _l, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'ge_ground_embed_bwd': (_i, [_vp, _vp, _vp, _l, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'ge_ground_vanilla_fwd': (_i, [_vp, _vp, _l, _f, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'ge_ground_vanilla_bwd': (_i, [_vp, _l, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'ge_depth_fuse_fwd': (_i, [_vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'ge_depth_fuse_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'ge_ground_plane': (_i, [_vp, _d, _vp, _vp, _i, _i, _vp]),
    'ge_slope_class': (_i, [_vp, _vp, _d, _i, _vp, _i, _i, _vp]),
    'ge_slope_class_ddad': (_i, [_vp, _vp, _d, _vp, _i, _i, _vp]),
    'ge_pe_channels': (_i, [_vp, _vp, _f, _l, _vp]),
    'ge_bn_act_nhwc_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _f, _f, _f, _i, _vp]),
    'ge_bn_act_nhwc_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _f, _i, _vp]),
    'ge_bias_act_nhwc_fwd': (_i, [_vp, _vp, _l, _i, _f, _i, _vp]),
    'ge_bias_act_nhwc_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _l, _i, _f, _i, _vp]),
    'ge_bilinear_nhwc_fwd': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_bilinear_nhwc_bwd': (_i, [_vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_concat_rows_fwd': (_i, [_vp, _l, _l, _vp, _vp, _vp, _l, _i, _i, _i, _f, _u64, _i, _vp]),
    'ge_slice_rows_drop': (_i, [_vp, _vp, _l, _i, _i, _i, _f, _u64, _i, _vp]),
    'ge_add_rows': (_i, [_vp, _vp, _vp, _i, _l, _i, _i, _vp]),
    'ge_colsum': (_i, [_vp, _l, _i, _vp, _vp, _i, _i, _vp]),
    'ge_bias_gelu_fwd': (_i, [_vp, _vp, _vp, _l, _i, _i, _vp]),
    'ge_bias_gelu_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _vp]),
    'ge_upcat_nhwc_fwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_upcat_nhwc_bwd': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_upsum_nhwc_fwd': (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_conv3x3_nhwc_fwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    'ge_conv3x3_nhwc_wgrad': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_conv1x1_nhwc_wgrad': (_i, [_vp, _vp, _vp, _l, _i, _i, _i, _vp]),
    'ge_conv1x1_bn_workspace': (_sz, [_i, _i]),
    'ge_conv1x1_bn_stats': (_i, [_vp, _l, _i, _vp, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    'ge_conv1x1_bn_act_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _l, _i, _i, _f, _vp]),
    'ge_conv1x1_bn_bwd_mask': (_i, [_vp, _l, _vp, _l, _vp, _vp, _vp, _vp, _l, _i, _f, _vp]),
    'ge_conv1x1_bn_bwd_finalize': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'ge_conv1x1_bn_dgrad': (_i, [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, _vp]),
    'ge_conv3x3_c1_fwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'ge_conv3x3_c1_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'ge_gemm_nt': (_i, [_vp, _l, _vp, _l, _vp, _vp, _l, _l, _i, _i, _i, _vp]),
    'ge_nhwc_workspace': (_sz, [_i, _i]),
    'ge_aug_load': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'ge_aug_depth': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'ge_aug_resize': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'ge_aug_rotate': (_i, [_vp, _vp, _i, _i, _i, _vp, _f, _i, _vp]),
    'ge_aug_window': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'ge_aug_color_normalize': (_i, [_vp, _vp, _i, _i, _i, _f, _f, _vp, _vp, _vp, _f, _i, _vp]),
    'ge_aug_area_u8': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'ge_aug_splat': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'ge_silog_stats': (_i, [_vp, _vp, _f, _vp, _l, _vp]),
    'ge_silog_bwd': (_i, [_vp, _vp, _f, _vp, _vp, _vp, _l, _vp]),
    'ge_sumsq': (_i, [_vp, _l, _vp, _vp]),
    'ge_adamw_step': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _vp]),
    'ge_adamw_step_shadow': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _vp, _vp]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library; raises HipLibraryError if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise HipLibraryError(
                f'{LIB_PATH} is missing: the gfx950 HIP kernels are not built. '
                f'Run gedepth_amd/csrc/build.sh (hipcc --offload-arch=gfx950). There is no CPU/eager fallback.')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)       # AttributeError here == ABI mismatch: fail loudly
            fn.restype, fn.argtypes = res, args
        _lib = handle
        if os.environ.get('GE_MSDA_MODE'):           # kernel-selection knob of the deformable attention (kernels.msda_mode), e.g.
            handle.ge_msda_mode(int(os.environ['GE_MSDA_MODE']))      # 60 = default without the bf16-tap-weight window forward
    return _lib


def is_built():
    return os.path.isfile(LIB_PATH)


def check(code, what):
    if code != 0:
        raise RuntimeError(f'{what} failed with code {code} '
                           f'({"bad argument" if code == 10001 else "unsupported" if code == 10002 else "hipError_t"})')


def dtype_code(t):
    if t.dtype == torch.float32:
        return GE_F32
    if t.dtype == torch.bfloat16:
        return GE_BF16
    raise TypeError(f'gedepth_amd kernels take float32 or bfloat16 tensors, got {t.dtype}')


def ptr(t, dtype=None, name='tensor'):
    """Device pointer of a contiguous HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f'{name}: gedepth_amd ops run on MI355X only; got a {t.device} tensor '
                           '(the CPU oracle lives in oracle/ and is test infrastructure)')
    if not t.is_contiguous():
        raise RuntimeError(f'{name} must be contiguous')
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f'{name} must be {dtype}, got {t.dtype}')
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream
