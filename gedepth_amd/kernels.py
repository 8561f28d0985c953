"""torch.autograd bindings of the HIP kernels (one Function per C-ABI op pair).

PyTorch is used for device memory, streams and autograd plumbing only; every forward and
backward below is a call into libgedepth_hip.so through gedepth_amd.hip (no eager fallback).
"""
import ctypes
import os

import torch

from . import hip

_f32 = torch.float32


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class _Profiler:
    """Per-kernel HIP-event timing on the launch stream (bench.py's roofline object).  Off by default."""

    def __init__(self):
        self.on, self.records, self.stage_bytes = False, {}, {}

    def enable(self):
        self.on, self.records, self.stage_bytes = True, {}, {}
        hip.lib().ge_msda_bwd_timing(1)

    def disable(self):
        self.on = False

    def add_stage_bytes(self, per_stage):
        """algorithmic bytes of one composite ge_msda_bwd call, per kernel (stage order of ge_msda_bwd_timing_read)"""
        for i, b in enumerate(per_stage):
            self.stage_bytes[i] = self.stage_bytes.get(i, 0) + int(b)

    def msda_bwd_stages(self):
        """Per-kernel records of the composite MSDA backward, timed by HIP events inside the library."""
        out = []
        total, n, name = ctypes.c_double(), ctypes.c_long(), ctypes.create_string_buffer(64)
        for i in range(5):
            hip.check(hip.lib().ge_msda_bwd_timing_read(i, ctypes.addressof(total), ctypes.addressof(n),
                                                        ctypes.addressof(name), 64), 'ge_msda_bwd_timing_read')
            if n.value:
                out.append(dict(name=name.value.decode(), launches=n.value, avg_us=1e3 * total.value / n.value,
                                total_ms=total.value, bytes_per_launch=self.stage_bytes.get(i, 0) / n.value))
        return out

    def run(self, name, nbytes, call, flops=0):
        if not self.on:
            return call()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = call()
        e.record()
        rec = self.records.setdefault(name, dict(bytes=int(nbytes), flops=int(flops), events=[]))
        rec['events'].append((s, e))
        return r

    def summary(self):
        if not self.records:
            return []
        torch.cuda.synchronize()
        out = []
        for name, rec in self.records.items():
            ms = [s.elapsed_time(e) for s, e in rec['events']]
            out.append(dict(name=name, launches=len(ms), avg_us=1e3 * sum(ms) / len(ms), total_ms=sum(ms),
                            bytes_per_launch=rec['bytes'], flops_per_launch=rec.get('flops', 0)))
        return out


PROFILER = _Profiler()

# Ledger of the places where a module took ATen's generic kernels for a CUDA tensor although this library has a HIP kernel
# for the op (shape / dtype / mode outside what the kernel covers).  Plumbing, not an error — but bench.py and the GPU tests
# assert that the measured training step has none (DESIGN.md §1 "no eager fallback on the hot path").
FALLBACKS = {}


def note_fallback(site, why=''):
    key = f'{site}: {why}' if why else site
    FALLBACKS[key] = FALLBACKS.get(key, 0) + 1


def _es(t):
    return t.element_size()


# A/B switch for the fused passes added in round 3: GE_DISABLE=upcat,upsum,bias_gelu,msda_raw makes the named entry points take their
# two-pass composition (still HIP kernels / library calls: a measurement aid for same-box comparisons, not a fall-back)
DISABLED = {t for t in os.environ.get('GE_DISABLE', '').split(',') if t}
ENABLED = {t for t in os.environ.get('GE_ENABLE', '').split(',') if t}        # opt-in paths that measured slower than their alternative


def _tag(t):
    return 'bf16' if t.dtype == torch.bfloat16 else 'f32'


# ------------------------------------------------------------------------ window attention
class _WindowAttention(torch.autograd.Function):

    @staticmethod
    def forward(ctx, qkv, qkv_bias, bias_table, H, W, num_heads, shift, scale, variant):
        qkv = _c(qkv)
        B, L, C3 = qkv.shape
        assert L == H * W and C3 == 3 * num_heads * 32, (qkv.shape, H, W, num_heads)
        qkv_bias = _c(qkv_bias.detach().to(_f32))
        bias_table = _c(bias_table.detach().to(_f32))
        out = torch.empty(B, L, C3 // 3, device=qkv.device, dtype=qkv.dtype)
        nbytes = 4 * B * L * (C3 // 3) * _es(qkv) + 169 * num_heads * 4
        n_wh = B * ((H + 6) // 7) * ((W + 6) // 7) * num_heads            # (window, head) pairs; 2 contractions of 2*49*49*32 flops each
        PROFILER.run(f'window_attn_fwd[{B}x{H}x{W} nH{num_heads} s{shift} {_tag(qkv)} v{variant}]', nbytes, lambda: hip.check(
            hip.lib().ge_window_attn_fwd(
                hip.ptr(qkv, name='qkv'), hip.ptr(qkv_bias, _f32), hip.ptr(bias_table, _f32), hip.ptr(out),
                B, H, W, num_heads, shift, scale, hip.dtype_code(qkv), variant, hip.stream()), 'ge_window_attn_fwd'),
            flops=n_wh * 2 * 2 * 49 * 49 * 32)
        ctx.save_for_backward(qkv, qkv_bias, bias_table)
        ctx.geom = (B, H, W, num_heads, shift, scale, variant)
        return out

    @staticmethod
    def backward(ctx, d_out):
        qkv, qkv_bias, bias_table = ctx.saved_tensors
        B, H, W, nH, shift, scale, variant = ctx.geom
        d_out = _c(d_out.to(qkv.dtype))
        d_qkv = torch.empty_like(qkv)
        d_qb = torch.empty_like(qkv_bias)
        d_tab = torch.empty_like(bias_table)
        lib = hip.lib()
        ws = torch.empty(max(int(lib.ge_window_attn_bwd_workspace(B, H, W, nH)), 4) // 4, device=qkv.device, dtype=_f32)
        nbytes = 7 * qkv.numel() // 3 * _es(qkv)
        n_wh = B * ((H + 6) // 7) * ((W + 6) // 7) * nH                     # backward: S, dP, dQ, dK, dV = 5 contractions
        PROFILER.run(f'window_attn_bwd[{B}x{H}x{W} nH{nH} s{shift} {_tag(qkv)} v{variant}]', nbytes, lambda: hip.check(
            lib.ge_window_attn_bwd(
                hip.ptr(qkv), hip.ptr(qkv_bias), hip.ptr(bias_table), hip.ptr(d_out), hip.ptr(d_qkv), hip.ptr(d_qb),
                hip.ptr(d_tab), hip.ptr(ws), B, H, W, nH, shift, scale, hip.dtype_code(qkv), variant, hip.stream()),
            'ge_window_attn_bwd'), flops=n_wh * 5 * 2 * 49 * 49 * 32)
        return d_qkv, d_qb, d_tab, None, None, None, None, None, None


def window_attention(qkv, qkv_bias, bias_table, H, W, num_heads, shift, scale, variant=0):
    """qkv (B, H*W, 3C) -> (B, H*W, C); see ge_window_attn_fwd in include/gedepth_hip.h."""
    return _WindowAttention.apply(qkv, qkv_bias, bias_table, int(H), int(W), int(num_heads), int(shift),
                                  float(scale), int(variant))


# ------------------------------------------------------------------------------------ MSDA
MSDA_BINNED_BACKWARD = True     # False: single-pass fp32-atomic scatter (no workspace)

def _levels(spatial_shapes):
    flat = [int(v) for hw in spatial_shapes for v in hw]
    return (ctypes.c_int * len(flat))(*flat), len(flat) // 2


def _query_grid(query_shapes, Nq):
    if query_shapes is None:
        return None, 0
    flat = [int(v) for hw in query_shapes for v in hw]
    assert sum(flat[i] * flat[i + 1] for i in range(0, len(flat), 2)) == Nq, (query_shapes, Nq)
    return ctypes.cast((ctypes.c_int * len(flat))(*flat), ctypes.c_void_p), len(flat) // 2


def msda_mode(mode=-1):
    """Kernel-selection knob of the deformable-attention ops; returns the previous mode.  Bits: 0 window (LDS-staged) forward,
    1 window d_loc / d_attw, 2 owner-lane tap arithmetic in the window kernels, 3 head-major work order of the streaming kernels,
    4 bf16 d_value drain on MFMA, 5 its operand reads through ds_read_b64_tr_b16, 6 drain work order grouped by query range (opt-in: less re-fetch traffic, same time).
    Default 61; ``GE_MSDA_MODE`` in the environment sets it at library load (e.g. 60: streaming forward with exact fp32 tap weights for accuracy-parity runs in bf16)."""
    return int(hip.lib().ge_msda_mode(int(mode)))


class _MSDeformAttn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, value, loc, attw, spatial_shapes, query_shapes):
        value = _c(value)
        loc = _c(loc.to(_f32))
        attw = _c(attw.to(_f32))
        B, Nv, nH, D = value.shape
        assert D == 64, 'ge_msda: 64 channels per head'
        _, Nq, _, L, P, _ = loc.shape
        arr, nl = _levels(spatial_shapes)
        assert nl == L
        qarr, nq = _query_grid(query_shapes, Nq)
        out = torch.empty(B, Nq, nH * D, device=value.device, dtype=value.dtype)
        nbytes = value.numel() * _es(value) + loc.numel() * 4 + attw.numel() * 4 + out.numel() * _es(out)
        PROFILER.run(f'msda_fwd[B{B} Nq{Nq} Nv{Nv} {_tag(value)}{" win" if nq else ""}]', nbytes, lambda: hip.check(
            hip.lib().ge_msda_fwd(hip.ptr(value, name='value'), ctypes.cast(arr, ctypes.c_void_p), qarr, nq, hip.ptr(loc), hip.ptr(attw),
                                  hip.ptr(out), B, Nv, Nq, nH, L, P, hip.dtype_code(value), hip.stream()), 'ge_msda_fwd'))
        ctx.save_for_backward(value, loc, attw)
        ctx.shapes = tuple(tuple(int(v) for v in hw) for hw in spatial_shapes)
        ctx.qshapes = None if query_shapes is None else tuple(tuple(int(v) for v in hw) for hw in query_shapes)
        return out

    @staticmethod
    def backward(ctx, d_out):
        value, loc, attw = ctx.saved_tensors
        B, Nv, nH, D = value.shape
        _, Nq, _, L, P, _ = loc.shape
        d_out = _c(d_out.to(value.dtype))
        arr, _ = _levels(ctx.shapes)
        qarr, nq = _query_grid(ctx.qshapes, Nq)
        d_value = torch.zeros(B, Nv, nH, D, device=value.device, dtype=_f32)
        d_loc = torch.empty_like(loc)
        d_attw = torch.empty_like(attw)
        nbytes = (value.numel() * _es(value) + 2 * loc.numel() * 4 + 2 * attw.numel() * 4 + d_out.numel() * _es(d_out)
                  + d_value.numel() * 4)
        lib = hip.lib()
        shapes_p = ctypes.cast(arr, ctypes.c_void_p)
        ws_bytes = int(lib.ge_msda_bwd_workspace(shapes_p, B, Nv, Nq, nH, L, P)) if MSDA_BINNED_BACKWARD else 0
        ws = torch.empty(ws_bytes, device=value.device, dtype=torch.uint8) if ws_bytes else None
        if PROFILER.on and ws_bytes:
            lw_b = value.numel() * _es(value) + 2 * (loc.numel() + attw.numel()) * 4 + d_out.numel() * _es(d_out)
            la_b = (loc.numel() + attw.numel()) * 4
            PROFILER.add_stage_bytes((lw_b, la_b, 0, la_b, d_out.numel() * _es(d_out) + d_value.numel() * 4))
        PROFILER.run(f'msda_bwd[B{B} Nq{Nq} Nv{Nv} {_tag(value)}{" binned" if ws_bytes else ""}]', nbytes, lambda: hip.check(
            lib.ge_msda_bwd(hip.ptr(value), shapes_p, qarr, nq, hip.ptr(loc), hip.ptr(attw), hip.ptr(d_out),
                            hip.ptr(d_value), hip.ptr(d_loc), hip.ptr(d_attw), hip.ptr(ws), ws_bytes, B, Nv, Nq, nH, L, P,
                            hip.dtype_code(value), hip.stream()), 'ge_msda_bwd'))
        return d_value.to(value.dtype), d_loc, d_attw, None, None


def ms_deform_attn(value, spatial_shapes, sampling_locations, attention_weights, query_shapes=None):
    """value (B,Nv,nH,64), loc (B,Nq,nH,L,P,2) in [0,1], attw (B,Nq,nH,L,P) -> (B,Nq,nH*64).
    ``query_shapes``: the queries as a list of (H, W) maps in raster order (sum H*W == Nq) — lets the kernels tile them 2-D
    and sample from LDS-staged value windows (csrc/msda_win.hip); None = streaming kernels.
    fp32: the same results either way (to the order of the lane reductions, 2e-5 of the tensor scale).  bf16 storage: the window
    forward rounds the four tap weights (already multiplied by the attention weight) to bf16 — 2^-9 relative per tap, below the
    bf16 rounding of the output it ends in (measured: within one bf16 ulp of the streaming kernel, which keeps fp32 weights like
    mmcv) — and the MFMA drain rounds the d_value coefficients the same way; d_loc / d_attw always use exact fp32 weights.
    ``msda_mode`` / ``GE_MSDA_MODE`` select the exact-weight kernels (tests compare both)."""
    return _MSDeformAttn.apply(value, sampling_locations, attention_weights, spatial_shapes, query_shapes)


class _MSDAPrep(torch.autograd.Function):

    @staticmethod
    def forward(ctx, raw, ref, spatial_shapes, nH, L, P):
        raw = _c(raw)
        B, Nq, ld = raw.shape
        n_off, n_log = nH * L * P * 2, nH * L * P
        assert ld == n_off + n_log, 'raw = [sampling_offsets | attention_weights] columns of one GEMM'
        ref = ref.to(_f32)
        assert ref.is_cuda and tuple(ref.shape) == (B, Nq, L, 2), 'reference points (B, Nq, L, 2), may be an expanded view'
        if ref.stride(3) != 1:
            ref = ref.contiguous()
        arr, nl = _levels(spatial_shapes)
        assert nl == L
        loc = torch.empty(B, Nq, nH, L, P, 2, device=raw.device, dtype=_f32)
        attw = torch.empty(B, Nq, nH, L, P, device=raw.device, dtype=_f32)
        es = _es(raw)
        base = hip.ptr(raw, name='raw')
        PROFILER.run(f'msda_prep_fwd[B{B} Nq{Nq} {_tag(raw)}]', raw.numel() * es + (loc.numel() + attw.numel()) * 4, lambda: hip.check(
            hip.lib().ge_msda_prep_fwd(base, ld, base + n_off * es, ld, ref.data_ptr(), ref.stride(0), ref.stride(1), ref.stride(2),
                                       ctypes.cast(arr, ctypes.c_void_p), hip.ptr(loc), hip.ptr(attw), B, Nq, nH, L, P,
                                       hip.dtype_code(raw), hip.stream()), 'ge_msda_prep_fwd'))
        ctx.save_for_backward(attw)
        ctx.meta = (tuple(tuple(int(v) for v in hw) for hw in spatial_shapes), nH, L, P, ld, raw.dtype)
        return loc, attw

    @staticmethod
    def backward(ctx, d_loc, d_attw):
        attw, = ctx.saved_tensors
        shapes, nH, L, P, ld, dtype = ctx.meta
        B, Nq = attw.shape[:2]
        d_loc, d_attw = _c(d_loc.to(_f32)), _c(d_attw.to(_f32))
        arr, _ = _levels(shapes)
        d_raw = torch.empty(B, Nq, ld, device=attw.device, dtype=dtype)
        fused_ref = ctx.needs_input_grad[1] and nH in (1, 2, 4, 8, 16)
        d_ref = torch.empty(B, Nq, L, 2, device=attw.device, dtype=_f32) if fused_ref else None
        es = _es(d_raw)
        base = hip.ptr(d_raw)
        n_off = nH * L * P * 2
        PROFILER.run(f'msda_prep_bwd[B{B} Nq{Nq} {_tag(d_raw)}]', d_raw.numel() * es + (d_loc.numel() + 2 * attw.numel()) * 4,
                     lambda: hip.check(hip.lib().ge_msda_prep_bwd(
                         hip.ptr(d_loc), hip.ptr(d_attw), hip.ptr(attw), ctypes.cast(arr, ctypes.c_void_p), base, ld,
                         base + n_off * es, ld, hip.ptr(d_ref), B, Nq, nH, L, P, hip.dtype_code(d_raw), hip.stream()),
                         'ge_msda_prep_bwd'))
        if ctx.needs_input_grad[1] and not fused_ref:
            d_ref = d_loc.sum((2, 4))
        return d_raw, d_ref, None, None, None, None


def msda_prepare(raw, reference_points, spatial_shapes, num_heads, num_levels, num_points):
    """raw (B,Nq,[nH*L*P*2 offsets | nH*L*P logits]) + reference points (B,Nq,L,2) -> (loc, attw) fp32 for ms_deform_attn."""
    return _MSDAPrep.apply(raw, reference_points, spatial_shapes, num_heads, num_levels, num_points)


class _MSDeformAttnRaw(torch.autograd.Function):
    """prepare + sampling in one kernel each way (ge_msda_fwd_raw / ge_msda_bwd_raw)."""

    @staticmethod
    def forward(ctx, value, raw, ref, spatial_shapes, query_shapes, nH, L, P):
        value, raw = _c(value), _c(raw)
        B, Nv, _, D = value.shape
        _, Nq, ld = raw.shape
        n_off, n_log = nH * L * P * 2, nH * L * P
        assert D == 64 and ld == n_off + n_log and raw.dtype == value.dtype
        ref = ref.to(_f32)
        assert ref.is_cuda and tuple(ref.shape) == (B, Nq, L, 2)
        if ref.stride(3) != 1:
            ref = ref.contiguous()
        arr, _ = _levels(spatial_shapes)
        qarr, nq = _query_grid(query_shapes, Nq)
        loc = torch.empty(B, Nq, nH, L, P, 2, device=raw.device, dtype=_f32)
        attw = torch.empty(B, Nq, nH, L, P, device=raw.device, dtype=_f32)
        out = torch.empty(B, Nq, nH * D, device=value.device, dtype=value.dtype)
        es = _es(raw)
        base = hip.ptr(raw, name='raw')
        nbytes = value.numel() * _es(value) + raw.numel() * es + (loc.numel() + attw.numel()) * 4 + out.numel() * _es(out)
        PROFILER.run(f'msda_fwd_raw[B{B} Nq{Nq} Nv{Nv} {_tag(value)}]', nbytes, lambda: hip.check(hip.lib().ge_msda_fwd_raw(
            hip.ptr(value, name='value'), ctypes.cast(arr, ctypes.c_void_p), qarr, nq, base, ld, base + n_off * es, ld, ref.data_ptr(),
            ref.stride(0), ref.stride(1), ref.stride(2), hip.ptr(loc), hip.ptr(attw), hip.ptr(out), B, Nv, Nq, nH, L, P,
            hip.dtype_code(value), hip.stream()), 'ge_msda_fwd_raw'))
        # bf16: d_value is binned straight from the raw projections with 8-byte records (ge_msda_bwd_value_raw: half the record bytes
        # of the fp32 loc / attw path in fill and drain), so the backward needs them too
        ctx.value_from_raw = (value.dtype == torch.bfloat16 and L == 4 and P == 8 and 'msda_value_raw' not in DISABLED
                              and max(max(hw) for hw in spatial_shapes) <= 8191)
        if ctx.value_from_raw:
            ctx.save_for_backward(value, loc, attw, raw, ref)
        else:
            ctx.save_for_backward(value, loc, attw)
        ctx.meta = (tuple(tuple(int(v) for v in hw) for hw in spatial_shapes), tuple(tuple(int(v) for v in hw) for hw in query_shapes),
                    nH, L, P, ld, raw.dtype)
        ctx.last_loc = loc
        return out

    @staticmethod
    def backward(ctx, d_out):
        value, loc, attw = ctx.saved_tensors[:3]
        split = ctx.value_from_raw
        shapes, qshapes, nH, L, P, ld, dtype = ctx.meta
        B, Nv, _, D = value.shape
        Nq = loc.shape[1]
        d_out = _c(d_out.to(value.dtype))
        arr, _ = _levels(shapes)
        qarr, nq = _query_grid(qshapes, Nq)
        shapes_p = ctypes.cast(arr, ctypes.c_void_p)
        lib = hip.lib()
        d_value = torch.zeros(B, Nv, nH, D, device=value.device, dtype=_f32)
        d_raw = torch.empty(B, Nq, ld, device=value.device, dtype=dtype)
        d_ref = torch.empty(B, Nq, L, 2, device=value.device, dtype=_f32) if ctx.needs_input_grad[2] else None
        ws_bytes = int(lib.ge_msda_bwd_workspace(shapes_p, B, Nv, Nq, nH, L, P))
        ws = torch.empty(ws_bytes, device=value.device, dtype=torch.uint8)
        es = _es(d_raw)
        base = hip.ptr(d_raw)
        n_off = nH * L * P * 2
        if PROFILER.on:
            lw_b = value.numel() * _es(value) + (loc.numel() + attw.numel()) * 4 + d_raw.numel() * es + d_out.numel() * _es(d_out)
            la_b = (loc.numel() + attw.numel()) * 4
            PROFILER.add_stage_bytes((lw_b, 0, 0, 0, 0) if split else (lw_b, la_b, 0, la_b, d_out.numel() * _es(d_out) + d_value.numel() * 4))
        nbytes = (value.numel() * _es(value) + (loc.numel() + attw.numel()) * 4 + d_raw.numel() * es + d_out.numel() * _es(d_out)
                  + (0 if split else d_value.numel() * 4))
        PROFILER.run(f'msda_bwd_raw[B{B} Nq{Nq} Nv{Nv} {_tag(value)}{" lw" if split else ""}]', nbytes, lambda: hip.check(lib.ge_msda_bwd_raw(
            hip.ptr(value), shapes_p, qarr, nq, hip.ptr(loc), hip.ptr(attw), hip.ptr(d_out), None if split else hip.ptr(d_value), base, ld,
            base + n_off * es, ld, hip.ptr(d_ref), hip.ptr(ws), ws_bytes, B, Nv, Nq, nH, L, P, hip.dtype_code(value), hip.stream()),
            'ge_msda_bwd_raw'))
        if split:
            raw, ref = ctx.saved_tensors[3:]
            rbase = hip.ptr(raw)
            if PROFILER.on:
                PROFILER.add_stage_bytes((0, B * Nq * n_off * 2, 0, raw.numel() * 2, d_out.numel() * 2 + d_value.numel() * 4))
            PROFILER.run(f'msda_bwd_value_raw[B{B} Nq{Nq} Nv{Nv}]', B * Nq * n_off * 2 + raw.numel() * 2 + d_out.numel() * 2 + d_value.numel() * 4,
                         lambda: hip.check(lib.ge_msda_bwd_value_raw(
                             shapes_p, rbase, ld, rbase + n_off * 2, ld, ref.data_ptr(), ref.stride(0), ref.stride(1), ref.stride(2),
                             hip.ptr(d_out), hip.ptr(d_value), hip.ptr(ws), ws_bytes, B, Nv, Nq, nH, L, P, hip.dtype_code(value),
                             hip.stream()), 'ge_msda_bwd_value_raw'))
        return d_value.to(value.dtype), d_raw, d_ref, None, None, None, None, None


def ms_deform_attn_raw(value, raw, reference_points, spatial_shapes, query_shapes, num_heads, num_levels, num_points):
    """mmcv MultiScaleDeformableAttention's sampling half from the RAW projection outputs:
    ``ms_deform_attn(value, shapes, *msda_prepare(raw, reference_points, ...), query_shapes)`` — one kernel each way when the
    geometry allows (4 levels, 8 points, a query grid, workspace backward: ``ge_msda_raw_supported``), else exactly that composition.
    value (B,Nv,nH,64), raw (B,Nq,[nH*L*P*2 offsets | nH*L*P logits]) same dtype, reference points (B,Nq,L,2) -> (B,Nq,nH*64)."""
    B, Nv = value.shape[:2]
    Nq = raw.shape[1]
    fused = ('msda_raw' not in DISABLED and query_shapes is not None and MSDA_BINNED_BACKWARD and value.is_cuda and raw.dtype == value.dtype
             and value.dtype in (_f32, torch.bfloat16))
    if fused:
        arr, _ = _levels(spatial_shapes)
        qarr, nq = _query_grid(query_shapes, Nq)
        fused = bool(hip.lib().ge_msda_raw_supported(ctypes.cast(arr, ctypes.c_void_p), qarr, nq, B, Nv, Nq, num_heads, num_levels, num_points))
        fused = fused and int(hip.lib().ge_msda_bwd_workspace(ctypes.cast(arr, ctypes.c_void_p), B, Nv, Nq, num_heads, num_levels, num_points)) > 0
    if not fused:
        loc, attw = msda_prepare(raw, reference_points, spatial_shapes, num_heads, num_levels, num_points)
        return ms_deform_attn(value, spatial_shapes, loc, attw, query_shapes)
    return _MSDeformAttnRaw.apply(value, raw, reference_points, spatial_shapes, query_shapes, int(num_heads), int(num_levels), int(num_points))


# ---- deformable attention as MFMA contractions (csrc/msda_mm.hip): query orders + forward
_TILE_ORDER_CACHE = {}


def msda_tile_order(query_shapes, device, th=4, tw=8):
    """Query order for grid queries (self-attention: the queries ARE the token maps): each (H, W) segment is cut into th x tw
    tiles and the queries are listed tile by tile, so 32 consecutive entries are a compact 2-D patch.  int32 (Nq,), cached."""
    key = (tuple(tuple(int(v) for v in hw) for hw in query_shapes), str(device), th, tw)
    if key not in _TILE_ORDER_CACHE:
        parts, start = [], 0
        for h, w in key[0]:
            idx = torch.arange(h * w, dtype=torch.int64).view(h, w)
            ph, pw = (-h) % th, (-w) % tw
            idx = torch.nn.functional.pad(idx, (0, pw, 0, ph), value=-1)
            hh, ww = idx.shape
            t = idx.view(hh // th, th, ww // tw, tw).permute(0, 2, 1, 3).reshape(-1)
            parts.append(t[t >= 0] + start)
            start += h * w
        _TILE_ORDER_CACHE[key] = torch.cat(parts).to(torch.int32).to(device)
    return _TILE_ORDER_CACHE[key]


def msda_ref_order(ref_xy, level0_hw):
    """Query order for content-independent reference points (cross-attention, hahi.py:294-302): sort the queries by the Morton
    code of the level-0 cell (quarter-cell resolution) of their reference point.  ref_xy (Nq, 2) in [0, 1] -> int32 (Nq,)."""
    h, w = level0_hw
    x = (ref_xy[:, 0].float() * (4 * w)).clamp_(0, 4 * w - 1).to(torch.int64)
    y = (ref_xy[:, 1].float() * (4 * h)).clamp_(0, 4 * h - 1).to(torch.int64)

    def spread(v):
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        return (v | (v << 1)) & 0x55555555
    return torch.argsort((spread(y) << 1) | spread(x)).to(torch.int32)


def msda_fwd_mm(value, raw, ref, spatial_shapes, order=None, want_loc=False, nH=8, L=4, P=8):
    """ge_msda_fwd_mm: value (B,Nv,nH,64) bf16, raw (B,Nq,nH*L*P*3) bf16, ref (B,Nq,L,2) f32 (may be an expanded view)."""
    value, raw = _c(value), _c(raw)
    B, Nv, _, D = value.shape
    _, Nq, ld = raw.shape
    n_off = nH * L * P * 2
    assert D == 64 and ld == n_off + nH * L * P and raw.dtype == value.dtype == torch.bfloat16
    ref = ref.to(_f32)
    assert ref.is_cuda and tuple(ref.shape) == (B, Nq, L, 2)
    if ref.stride(3) != 1:
        ref = ref.contiguous()
    arr, _ = _levels(spatial_shapes)
    out = torch.empty(B, Nq, nH * D, device=value.device, dtype=value.dtype)
    loc = torch.empty(B, Nq, nH, L, P, 2, device=raw.device, dtype=_f32) if want_loc else None
    attw = torch.empty(B, Nq, nH, L, P, device=raw.device, dtype=_f32) if want_loc else None
    if order is not None:
        assert order.dtype == torch.int32 and order.is_cuda and order.numel() == Nq and order.is_contiguous()
    base = hip.ptr(raw, name='raw')
    nbytes = value.numel() * 2 + raw.numel() * 2 + out.numel() * 2 + ((loc.numel() + attw.numel()) * 4 if want_loc else 0)
    PROFILER.run(f'msda_mm_fwd_k[B{B} Nq{Nq} Nv{Nv}]', nbytes, lambda: hip.check(hip.lib().ge_msda_fwd_mm(
        hip.ptr(value, name='value'), ctypes.cast(arr, ctypes.c_void_p), base, ld, base + n_off * 2, ld, ref.data_ptr(),
        ref.stride(0), ref.stride(1), ref.stride(2), hip.ptr(order), hip.ptr(loc), hip.ptr(attw), hip.ptr(out), B, Nv, Nq, nH, L, P,
        hip.dtype_code(value), hip.stream()), 'ge_msda_fwd_mm'))
    return (out, loc, attw) if want_loc else out


class _MMValueChoice:
    """Which d_value kernel the MFMA deformable attention uses for one problem shape (as a level mask: all levels or none by default).  The transposed-contraction kernel
    (ge_msda_bwd_value_mm) costs ~2.1 ns per tile pass + ~0.28 ns per flushed window row (MI355X, tools/ubench/msda_mm/dv_time.py), i.e. it
    depends on how many consecutive query tiles share a window: at the KITTI shape with the model's reference points (8 queries per level-0
    cell) level 0 would take 5.3 ms, level 1 1.7, level 2 0.74, level 3 0.50; the record pipeline (ge_msda_bwd_value_raw_levels) costs
    ~0.70 ns per (query, head) over the four levels whatever the geometry (1.1 ms per level).  The run cutter leaves {rows, passes} per level
    in the workspace; they are copied to pinned host memory asynchronously and read by a LATER call (never a synchronisation), so the choice
    follows the geometry with a lag of a step.  GE_MSDA_VALUE = mm | records | <level bit mask of the MFMA kernel> pins it."""
    NS_PASS, NS_ROW, NS_QH = 2.1, 0.28, 0.70
    EVERY = 16                                        # steady state: look at the statistics every 16th call

    def __init__(self):
        # start on the record pipeline: its cost does not depend on the geometry, and the statistics-only call (~50 us) still runs, so a geometry that favours
        # the MFMA kernel switches over within a few calls.  (Until round 6 the first calls ran the MFMA kernel: 9 ms each at the bench model's geometry
        # during warm-up, and a graph captured before the statistics arrived would have kept it.)
        self.mm_mask, self.calls, self.pending, self.last = 0, 0, None, None
        env = os.environ.get('GE_MSDA_VALUE', '')
        self.forced = {'mm': 15, 'records': 0, 'vs': 0}.get(env, int(env) if env.isdigit() else None)
        self.vs = env == 'vs'                             # the value-stationary kernel (ge_msda_bwd_value_vs): opt-in, see _MSDeformAttnMM.backward

    @property
    def use_mm(self):
        return self.mm_mask != 0

    def wants_stats(self):
        return self.forced is None and (self.calls <= 4 or self.calls % self.EVERY == 0) and not torch.cuda.is_current_stream_capturing()

    def observe(self, ws, offset, n_qh):
        if self.forced is not None or torch.cuda.is_current_stream_capturing() or not self.wants_stats() or self.pending is not None:
            return
        host = torch.empty(10, dtype=torch.int32).pin_memory()
        host.copy_(ws[offset:offset + 40].view(torch.int32), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending = (host, ev, n_qh)

    def update(self):
        self.calls += 1
        if torch.cuda.is_current_stream_capturing():      # a captured step keeps the choice it was captured with
            return
        if self.pending is not None and self.pending[1].query():
            host, _, n_qh = self.pending
            self.pending = None
            if int(host[1]) > 0:
                rows, passes = int(host[0]), int(host[1])
                t_mm, t_rec = self.NS_PASS * passes + self.NS_ROW * rows, self.NS_QH * n_qh
                # all levels on one kernel or the other.  A per-level split (coarse levels on the MFMA kernel, fine ones through the
                # records: GE_MSDA_VALUE=12) is supported and tested but measured SLOWER at the KITTI shape: the record pipeline restricted
                # to levels 0-1 still costs 3.3 of its 4.7 ms (its count / fill passes are bound by the per-point loads and tap arithmetic,
                # not by the records they emit) and the MFMA kernel on levels 2-3 alone 2.0 ms (latency of a run's serial tile passes with half
                # the runs to overlap): 5.4 vs 4.7 ms (tools/ubench/msda_mm/dv_time.py, round 5)
                on = self.mm_mask == 15
                self.mm_mask = 15 if t_mm < (1.1 if on else 0.9) * t_rec else 0          # hysteresis: change only for a 10 % predicted gain
                self.last = dict(rows=rows, passes=passes, mm_ms=round(t_mm * 1e-6, 3), records_ms=round(t_rec * 1e-6, 3),
                                 per_level=[dict(rows=int(host[2 + 2 * l]), passes=int(host[3 + 2 * l])) for l in range(4)])


_MM_VALUE_CHOICE = {}


def _mm_value_choice(key, mm_ok, rec_ok):
    """-> (choice object, level mask of the MFMA kernel for this call)."""
    c = _MM_VALUE_CHOICE.get(key)
    if c is None:
        c = _MM_VALUE_CHOICE[key] = _MMValueChoice()
    c.update()
    mask = c.mm_mask if c.forced is None else c.forced
    mask = 0 if not mm_ok else 15 if not rec_ok else mask            # only one of the two covers this call: no choice to make
    return c, mask


class _MSDeformAttnMM(torch.autograd.Function):
    """Deformable attention from the raw projections on the MFMA decomposition (csrc/msda_mm.hip): forward ge_msda_fwd_mm, backward
    ge_msda_bwd_lw_mm (d_raw) + ge_msda_dref (d_ref) + the binned d_value scatter (ge_msda_bwd_value)."""

    @staticmethod
    def forward(ctx, value, raw, ref, order, spatial_shapes, nH, L, P):
        out = msda_fwd_mm(value, raw, ref, spatial_shapes, order, want_loc=False, nH=nH, L=L, P=P)    # no fp32 loc / attw tensors exist
        ctx.save_for_backward(value, raw, ref, order)
        ctx.meta = (tuple(tuple(int(v) for v in hw) for hw in spatial_shapes), nH, L, P)
        return out

    @staticmethod
    def backward(ctx, d_out):
        value, raw, ref, order = ctx.saved_tensors
        shapes, nH, L, P = ctx.meta
        value, raw = _c(value), _c(raw)
        B, Nv, _, D = value.shape
        _, Nq, ld = raw.shape
        n_off = nH * L * P * 2
        d_out = _c(d_out.to(value.dtype))
        ref = ref.to(_f32)
        if ref.stride(3) != 1:
            ref = ref.contiguous()
        arr, _ = _levels(shapes)
        shapes_p = ctypes.cast(arr, ctypes.c_void_p)
        lib = hip.lib()
        d_raw = torch.empty(B, Nq, ld, device=value.device, dtype=raw.dtype)
        base, dbase = hip.ptr(raw), hip.ptr(d_raw)
        nb_lw = value.numel() * 2 + raw.numel() * 2 + d_raw.numel() * 2 + d_out.numel() * 2
        # round 5: d_value as the transposed contraction dV_window = C^T dO (ge_msda_bwd_value_mm); the d_raw kernel leaves the per-tile
        # tap boxes in the shared workspace.  Its cost depends on the geometry (how compact the windows of consecutive query tiles are), the
        # record pipeline's (ge_msda_bwd_value_raw) does not: _mm_value_choice picks per call from the run statistics of earlier calls
        want_dv = ctx.needs_input_grad[0]
        # round 6: GE_MSDA_VALUE=vs routes d_value through the value-stationary kernel (ge_msda_bwd_value_vs: a workgroup owns a 24 x 16 super-block
        # of value rows and walks the query tiles that reach it; no records, one write per block).  Built as the review's route (i) and MEASURED
        # SLOWER at the bench shapes — cross 6.3 vs 4.9 ms, self 4.8 vs 2.0 ms (tools/ubench/msda_mm/vs_time.py): 1.4 M tile visits x 4 waves x ~390
        # instructions are ~3.5 ms of pure issue time before any latency — so it stays opt-in; it is correct for any geometry (stray tiles fall
        # back per tile to the atomic kernel) and tested like the other two
        # (its kernel addresses rows with 32-bit byte offsets: tensors of 4 GB and more keep the other two kernels)
        use_vs = (want_dv and os.environ.get('GE_MSDA_VALUE') == 'vs' and B * Nq * ld * 2 < 2 ** 32 and B * Nq * nH * D * 2 < 2 ** 32
                  and 0 <= ref.stride(1) and Nq * ref.stride(1) * 4 < 2 ** 32)
        vs_ws_bytes = int(lib.ge_msda_bwd_vs_workspace(shapes_p, B, Nv, Nq, nH, L, P)) if use_vs else 0
        mm_ws_bytes = int(lib.ge_msda_bwd_mm_workspace(B, Nq, nH, L)) if (want_dv and 'msda_value_mm' not in DISABLED) else 0
        if vs_ws_bytes:
            mm_ws_bytes = vs_ws_bytes                 # its head is the ge_msda_bwd_mm_workspace layout (tap boxes, run lists)
        mm_ws = torch.empty(mm_ws_bytes, device=value.device, dtype=torch.uint8) if mm_ws_bytes else None
        PROFILER.run(f'msda_mm_bwd_lw_k[B{B} Nq{Nq} Nv{Nv}]', nb_lw, lambda: hip.check(lib.ge_msda_bwd_lw_mm(
            hip.ptr(value), shapes_p, base, ld, base + n_off * 2, ld, ref.data_ptr(), ref.stride(0), ref.stride(1), ref.stride(2),
            hip.ptr(order), hip.ptr(d_out), dbase, ld, dbase + n_off * 2, ld, hip.ptr(mm_ws), B, Nv, Nq, nH, L, P, hip.dtype_code(value),
            hip.stream()), 'ge_msda_bwd_lw_mm'))
        d_ref = None
        if ctx.needs_input_grad[2]:
            d_ref = torch.empty(B, Nq, L, 2, device=value.device, dtype=_f32)
            hip.check(lib.ge_msda_dref(dbase, ld, shapes_p, hip.ptr(d_ref), B * Nq, nH, L, P, hip.dtype_code(value), hip.stream()), 'ge_msda_dref')
        d_value = None
        if want_dv:
            d_value = torch.zeros(B, Nv, nH, D, device=value.device, dtype=_f32)
            if vs_ws_bytes:
                PROFILER.run(f'msda_mm_bwd_vs_k[B{B} Nq{Nq} Nv{Nv}]', raw.numel() * 2 + d_out.numel() * 2 + d_value.numel() * 4, lambda: hip.check(
                    lib.ge_msda_bwd_value_vs(shapes_p, base, ld, base + n_off * 2, ld, ref.data_ptr(), ref.stride(0), ref.stride(1), ref.stride(2), hip.ptr(order),
                                             hip.ptr(d_out), hip.ptr(d_value), hip.ptr(mm_ws), mm_ws_bytes, B, Nv, Nq, nH, L, P, hip.dtype_code(value), hip.stream()),
                    'ge_msda_bwd_value_vs'))
                return d_value.to(value.dtype), d_raw, d_ref, None, None, None, None, None
            rec_ws_bytes = int(lib.ge_msda_bwd_workspace(shapes_p, B, Nv, Nq, nH, L, P))
            choice, mm_mask = _mm_value_choice((B, Nq, Nv, nH, shapes, str(value.device)), mm_ws is not None, rec_ws_bytes > 0)
            rec_mask = 15 & ~mm_mask

            def value_mm(dv, mask):
                return hip.check(lib.ge_msda_bwd_value_mm(
                    shapes_p, base, ld, base + n_off * 2, ld, ref.data_ptr(), ref.stride(0), ref.stride(1), ref.stride(2), hip.ptr(order),
                    hip.ptr(d_out), hip.ptr(dv), hip.ptr(mm_ws), mm_ws_bytes, mask, B, Nv, Nq, nH, L, P, hip.dtype_code(value), hip.stream()),
                    'ge_msda_bwd_value_mm')
            if mm_mask:
                # algorithmic bytes: the raw projections + d_out read once per level, those levels' d_value rows written once
                lv_rows = sum(h * w for l, (h, w) in enumerate(shapes) if (mm_mask >> l) & 1)
                PROFILER.run(f'msda_mm_bwd_v_k[B{B} Nq{Nq} Nv{Nv} levels {mm_mask:04b}]',
                             bin(mm_mask).count('1') * (raw.numel() // 2 + d_out.numel() * 2 // 4) + B * lv_rows * nH * D * 4, lambda: value_mm(d_value, mm_mask))
            elif mm_ws is not None and choice.wants_stats():
                value_mm(None, 0)                   # run statistics only (~50 us): keeps the choice informed while the record path runs
            if rec_mask:
                if rec_ws_bytes <= 0:
                    raise RuntimeError('ms_deform_attn_mm backward: neither the MFMA d_value kernel nor the binned path covers this geometry')
                ws = torch.empty(rec_ws_bytes, device=value.device, dtype=torch.uint8)
                frac = bin(rec_mask).count('1') / 4
                if PROFILER.on:       # algorithmic bytes per stage: count reads the offsets, fill offsets + logits, drain d_out + d_value
                    PROFILER.add_stage_bytes((0, B * Nq * n_off * 2, 0, raw.numel() * 2, int(frac * d_out.numel() * 2) + d_value.numel() * 4))
                PROFILER.run(f'msda_bwd_value_raw[B{B} Nq{Nq} Nv{Nv}{"" if rec_mask == 15 else f" levels {rec_mask:04b}"}]',
                             B * Nq * n_off * 2 + raw.numel() * 2 + d_out.numel() * 2 + d_value.numel() * 4,
                             lambda: hip.check(lib.ge_msda_bwd_value_raw_levels(
                                 shapes_p, base, ld, base + n_off * 2, ld, ref.data_ptr(), ref.stride(0), ref.stride(1), ref.stride(2),
                                 hip.ptr(d_out), hip.ptr(d_value), hip.ptr(ws), rec_ws_bytes, rec_mask, B, Nv, Nq, nH, L, P, hip.dtype_code(value),
                                 hip.stream()), 'ge_msda_bwd_value_raw_levels'))
            if mm_ws is not None:
                choice.observe(mm_ws, int(lib.ge_msda_bwd_mm_stats_offset(B, Nq, nH, L)), B * Nq * nH)
            d_value = d_value.to(value.dtype)
        return d_value, d_raw, d_ref, None, None, None, None, None


def msda_mm_supported(value, raw, spatial_shapes, nH, L, P):
    if not (value.is_cuda and value.dtype == torch.bfloat16 and raw.dtype == torch.bfloat16 and value.shape[-1] == 64):
        return False
    arr, _ = _levels(spatial_shapes)
    B, Nv = value.shape[:2]
    return bool(hip.lib().ge_msda_mm_supported(ctypes.cast(arr, ctypes.c_void_p), B, Nv, raw.shape[1], nH, L, P, hip.GE_BF16))


def ms_deform_attn_mm(value, raw, reference_points, spatial_shapes, order, num_heads=8, num_levels=4, num_points=8):
    """``ms_deform_attn_raw`` on the MFMA decomposition: queries are processed in ``order`` (int32 permutation, or None)."""
    return _MSDeformAttnMM.apply(value, raw, reference_points, order, spatial_shapes, int(num_heads), int(num_levels), int(num_points))


class _MSDeformAttnSelfSplit(torch.autograd.Function):
    """The SELF-attention's sampling (queries = the token maps of the value levels, reference points = their own pixel centres) split by
    query level (round 6): the level-0 queries — 75 % of them, whose 4 x 8 query patches see compact value windows — run on the MFMA
    decomposition (ge_msda_fwd_mm_part / ge_msda_bwd_lw_mm_part, rows addressed in place through the row pitch), the coarse-level queries —
    whose 32-query patches span 16 - 64 level-0 cells, i.e. windows of many chunks — stay on the LDS-window gather kernels (ge_msda_fwd_raw /
    ge_msda_bwd_raw on contiguous copies of their rows: 8 085 of 32 725).  d_value comes from the record pipeline over ALL queries as before
    (splitting it costs more than it saves: its count / fill passes are bound by per-launch work).  Measured at 8 x 32 725 queries
    (tools/ubench/msda_mm/self_split_time.py): forward 1.64 -> 0.51 + 0.51 ms, d_raw 1.73 -> 0.62 + 0.52 ms."""

    @staticmethod
    def forward(ctx, value, raw, ref, spatial_shapes, n_fine, order_fine, nH, L, P):
        value, raw = _c(value), _c(raw)
        B, Nv, _, D = value.shape
        _, Nq, ld = raw.shape
        n_off = nH * L * P * 2
        assert D == 64 and ld == n_off + nH * L * P and raw.dtype == value.dtype == torch.bfloat16 and 0 < n_fine < Nq
        ref = ref.to(_f32)
        if ref.stride(3) != 1:
            ref = ref.contiguous()
        shapes = tuple(tuple(int(v) for v in hw) for hw in spatial_shapes)
        arr, _ = _levels(shapes)
        shapes_p = ctypes.cast(arr, ctypes.c_void_p)
        lib = hip.lib()
        out = torch.empty(B, Nq, nH * D, device=value.device, dtype=value.dtype)
        base = hip.ptr(raw, name='raw')
        nc = Nq - n_fine
        PROFILER.run(f'msda_mm_fwd_k[B{B} Nq{n_fine}/{Nq} Nv{Nv}]', value.numel() * 2 + B * n_fine * (ld + nH * D) * 2, lambda: hip.check(lib.ge_msda_fwd_mm_part(
            hip.ptr(value, name='value'), shapes_p, base, ld, base + n_off * 2, ld, ref.data_ptr(), ref.stride(0), ref.stride(1), ref.stride(2),
            hip.ptr(order_fine), hip.ptr(out), B, Nv, n_fine, Nq, nH, L, P, hip.GE_BF16, hip.stream()), 'ge_msda_fwd_mm_part'))
        # coarse-level queries: contiguous copies of their rows for the window kernels (whose row addressing is B * Nq)
        coarse_shapes = []
        acc = 0
        for hw in shapes:
            acc += hw[0] * hw[1]
            if acc > n_fine:
                coarse_shapes.append(hw)
        assert sum(h * w for h, w in coarse_shapes) == nc, 'n_fine must end on a level boundary'
        raw_c = raw[:, n_fine:].contiguous()
        ref_c = ref[:, n_fine:]
        qarr, nq = _query_grid(coarse_shapes, nc)
        loc = torch.empty(B, nc, nH, L, P, 2, device=raw.device, dtype=_f32)
        attw = torch.empty(B, nc, nH, L, P, device=raw.device, dtype=_f32)
        out_c = torch.empty(B, nc, nH * D, device=value.device, dtype=value.dtype)
        cbase = hip.ptr(raw_c)
        PROFILER.run(f'msda_fwd_raw[B{B} Nq{nc} Nv{Nv} bf16]', value.numel() * 2 + raw_c.numel() * 2 + (loc.numel() + attw.numel()) * 4 + out_c.numel() * 2,
                     lambda: hip.check(lib.ge_msda_fwd_raw(hip.ptr(value), shapes_p, qarr, nq, cbase, ld, cbase + n_off * 2, ld, ref_c.data_ptr(), ref_c.stride(0),
                                                           ref_c.stride(1), ref_c.stride(2), hip.ptr(loc), hip.ptr(attw), hip.ptr(out_c), B, Nv, nc, nH, L, P,
                                                           hip.GE_BF16, hip.stream()), 'ge_msda_fwd_raw'))
        out[:, n_fine:] = out_c
        ctx.save_for_backward(value, raw, ref, order_fine, loc, attw)
        ctx.meta = (shapes, tuple(coarse_shapes), n_fine, nH, L, P)
        return out

    @staticmethod
    def backward(ctx, d_out):
        value, raw, ref, order_fine, loc, attw = ctx.saved_tensors
        shapes, coarse_shapes, n_fine, nH, L, P = ctx.meta
        B, Nv, _, D = value.shape
        _, Nq, ld = raw.shape
        nc = Nq - n_fine
        n_off = nH * L * P * 2
        d_out = _c(d_out.to(value.dtype))
        arr, _ = _levels(shapes)
        shapes_p = ctypes.cast(arr, ctypes.c_void_p)
        lib = hip.lib()
        d_raw = torch.empty(B, Nq, ld, device=value.device, dtype=raw.dtype)
        base, dbase = hip.ptr(raw), hip.ptr(d_raw)
        PROFILER.run(f'msda_mm_bwd_lw_k[B{B} Nq{n_fine}/{Nq} Nv{Nv}]', value.numel() * 2 + B * n_fine * (2 * ld + nH * D) * 2, lambda: hip.check(lib.ge_msda_bwd_lw_mm_part(
            hip.ptr(value), shapes_p, base, ld, base + n_off * 2, ld, ref.data_ptr(), ref.stride(0), ref.stride(1), ref.stride(2), hip.ptr(order_fine),
            hip.ptr(d_out), dbase, ld, dbase + n_off * 2, ld, B, Nv, n_fine, Nq, nH, L, P, hip.GE_BF16, hip.stream()), 'ge_msda_bwd_lw_mm_part'))
        d_out_c = d_out[:, n_fine:].contiguous()
        d_raw_c = torch.empty(B, nc, ld, device=value.device, dtype=raw.dtype)
        qarr, nq = _query_grid(coarse_shapes, nc)
        ws_bytes = int(lib.ge_msda_bwd_workspace(shapes_p, B, Nv, nc, nH, L, P))
        ws = torch.empty(ws_bytes, device=value.device, dtype=torch.uint8)
        cb = hip.ptr(d_raw_c)
        PROFILER.run(f'msda_bwd_raw[B{B} Nq{nc} Nv{Nv} bf16 lw]', value.numel() * 2 + (loc.numel() + attw.numel()) * 4 + d_raw_c.numel() * 2 + d_out_c.numel() * 2,
                     lambda: hip.check(lib.ge_msda_bwd_raw(hip.ptr(value), shapes_p, qarr, nq, hip.ptr(loc), hip.ptr(attw), hip.ptr(d_out_c), None, cb, ld, cb + n_off * 2, ld,
                                                           None, hip.ptr(ws), ws_bytes, B, Nv, nc, nH, L, P, hip.GE_BF16, hip.stream()), 'ge_msda_bwd_raw'))
        d_raw[:, n_fine:] = d_raw_c
        d_value = None
        if ctx.needs_input_grad[0]:
            d_value = torch.zeros(B, Nv, nH, D, device=value.device, dtype=_f32)
            ws_bytes = int(lib.ge_msda_bwd_workspace(shapes_p, B, Nv, Nq, nH, L, P))
            ws = torch.empty(ws_bytes, device=value.device, dtype=torch.uint8)
            if PROFILER.on:
                PROFILER.add_stage_bytes((0, B * Nq * n_off * 2, 0, raw.numel() * 2, d_out.numel() * 2 + d_value.numel() * 4))
            PROFILER.run(f'msda_bwd_value_raw[B{B} Nq{Nq} Nv{Nv}]', B * Nq * n_off * 2 + raw.numel() * 2 + d_out.numel() * 2 + d_value.numel() * 4,
                         lambda: hip.check(lib.ge_msda_bwd_value_raw(shapes_p, base, ld, base + n_off * 2, ld, ref.data_ptr(), ref.stride(0), ref.stride(1), ref.stride(2),
                                                                     hip.ptr(d_out), hip.ptr(d_value), hip.ptr(ws), ws_bytes, B, Nv, Nq, nH, L, P, hip.GE_BF16, hip.stream()),
                                           'ge_msda_bwd_value_raw'))
            d_value = d_value.to(value.dtype)
        return d_value, d_raw, None, None, None, None, None, None, None


def msda_self_split_ok(value, raw, spatial_shapes, query_shapes, nH, L, P):
    """The level split of the self-attention's sampling applies: bf16 HIP tensors, queries = the value levels themselves (>= 2 of them), every
    kernel involved supports the geometry.  ``GE_DISABLE=msda_self_split`` keeps all queries on the window kernels."""
    if 'msda_self_split' in DISABLED or 'msda_mm' in DISABLED or 'msda_raw' in DISABLED or 'msda_value_raw' in DISABLED or not MSDA_BINNED_BACKWARD:
        return False
    shapes = [tuple(int(v) for v in hw) for hw in spatial_shapes]
    if query_shapes is None or [tuple(int(v) for v in hw) for hw in query_shapes] != shapes or len(shapes) < 2 or L != 4 or P != 8:
        return False
    if not (value.is_cuda and value.dtype == torch.bfloat16 and raw.dtype == torch.bfloat16 and value.shape[-1] == 64 and max(max(hw) for hw in shapes) <= 8191):
        return False
    B, Nv = value.shape[:2]
    Nq = raw.shape[1]
    n_fine = shapes[0][0] * shapes[0][1]
    if Nq != Nv or not msda_mm_supported(value, raw, shapes, nH, L, P):
        return False
    lib = hip.lib()
    arr, _ = _levels(shapes)
    sp = ctypes.cast(arr, ctypes.c_void_p)
    qarr, nq = _query_grid(shapes[1:], Nq - n_fine)
    return (bool(lib.ge_msda_raw_supported(sp, qarr, nq, B, Nv, Nq - n_fine, nH, L, P)) and int(lib.ge_msda_bwd_workspace(sp, B, Nv, Nq - n_fine, nH, L, P)) > 0
            and int(lib.ge_msda_bwd_workspace(sp, B, Nv, Nq, nH, L, P)) > 0)


def ms_deform_attn_self_split(value, raw, reference_points, spatial_shapes, num_heads=8, num_levels=4, num_points=8):
    """``ms_deform_attn_raw`` for the self-attention (queries = the value levels) with the level split of ``_MSDeformAttnSelfSplit``."""
    shapes = [tuple(int(v) for v in hw) for hw in spatial_shapes]
    n_fine = shapes[0][0] * shapes[0][1]
    order = msda_tile_order(shapes[:1], value.device)
    return _MSDeformAttnSelfSplit.apply(value, raw, reference_points, shapes, n_fine, order, int(num_heads), int(num_levels), int(num_points))


# ---------------------------------------------------------------------------- channels-last helpers
_CL = torch.channels_last


def _is_cl(x):
    """A 4-D map stored channels-last (B, H, W, C) with C > 1 — i.e. the row matrix (B*H*W, C) the NHWC kernels take."""
    return x.dim() == 4 and x.shape[1] > 1 and x.stride(1) == 1 and x.is_contiguous(memory_format=_CL)


def _cl_ok(x):
    """NHWC kernels apply: channels-last storage, C a multiple of the 16-byte vector and at most 256 vectors wide."""
    vn = 8 if x.dtype == torch.bfloat16 else 4
    return (x.is_cuda and _is_cl(x) and x.shape[1] % vn == 0 and x.shape[1] // vn <= 256 and x.dtype in (_f32, torch.bfloat16))


def _cl(x):
    return x if _is_cl(x) else x.contiguous(memory_format=_CL)


def _rows_of(x):
    """(rows, C) of a channels-last map"""
    B, C, H, W = x.shape
    return B * H * W, C


# ------------------------------------------------------------------- feature map <-> token sequence
def _planes(x):
    """(B,C,H,W) with contiguous (H,W) planes packed over C; the batch stride is free (channel slices of a concat)."""
    _, C, H, W = x.shape
    if x.stride(3) == 1 and x.stride(2) == W and x.stride(1) == H * W:
        return x
    return x.contiguous()


def _rows(t):
    """(B,N,C) with packed rows; the batch stride is free (token ranges of a longer sequence)."""
    if t.stride(2) == 1 and t.stride(1) == t.shape[2]:
        return t
    return t.contiguous()


def _raw_ptr(t, name):
    if not t.is_cuda:
        raise RuntimeError(f'{name}: gedepth_amd ops run on MI355X only; got a {t.device} tensor')
    return t.data_ptr()


class _TokensFromMap(torch.autograd.Function):

    @staticmethod
    def forward(ctx, fmap, pos):
        fmap = _planes(fmap)
        B, C, H, W = fmap.shape
        N = H * W
        if pos is not None:
            assert not pos.requires_grad and pos.numel() == C * N
            pos = _c(pos.to(_f32))
        tok = torch.empty(B, N, C, device=fmap.device, dtype=fmap.dtype)
        PROFILER.run(f'tokens_from_map[{B}x{C}x{N} {_tag(fmap)}]', 2 * tok.numel() * _es(tok) + (C * N * 4 if pos is not None else 0),
                     lambda: hip.check(hip.lib().ge_tokens_from_map(
                         _raw_ptr(fmap, 'map'), fmap.stride(0), hip.ptr(pos), hip.ptr(tok), N * C, B, C, N, 0.0, 0,
                         hip.dtype_code(fmap), hip.stream()), 'ge_tokens_from_map'))
        ctx.geom = (B, C, H, W)
        return tok

    @staticmethod
    def backward(ctx, d_tok):
        B, C, H, W = ctx.geom
        N = H * W
        d_tok = _rows(d_tok)
        d_map = torch.empty(B, C, H, W, device=d_tok.device, dtype=d_tok.dtype)
        PROFILER.run(f'map_from_tokens[{B}x{C}x{N} {_tag(d_tok)}]', 2 * d_map.numel() * _es(d_map), lambda: hip.check(
            hip.lib().ge_map_from_tokens(_raw_ptr(d_tok, 'd_tok'), d_tok.stride(0), None, 0, hip.ptr(d_map), C * N, B, C, N, 0.0, 0,
                                         hip.dtype_code(d_tok), hip.stream()), 'ge_map_from_tokens'))
        return d_map, None


class _AddRows(torch.autograd.Function):
    """tokens (B,N,C) + pos (N,C) fp32, one pass, one rounding (channels-last path of ``tokens_from_map``)."""

    @staticmethod
    def forward(ctx, tok, pos_rows):
        tok = _c(tok)
        B, N, C = tok.shape
        out = torch.empty_like(tok)
        PROFILER.run(f'add_rows[{B}x{N}x{C} {_tag(tok)}]', 2 * tok.numel() * _es(tok) + N * C * 4, lambda: hip.check(
            hip.lib().ge_add_rows(hip.ptr(tok, name='tokens'), hip.ptr(pos_rows, _f32), hip.ptr(out), B, N, C, hip.dtype_code(tok),
                                  hip.stream()), 'ge_add_rows'))
        return out

    @staticmethod
    def backward(ctx, d):
        d_pos = None
        if ctx.needs_input_grad[1]:                         # rows that carry a Parameter (HAHI's level embedding): batch sum in fp32
            d_pos = d.sum(0, dtype=_f32)
        return d, d_pos


def add_rows(tokens, rows):
    """tokens (B,N,C) + rows (N,C) fp32 broadcast over the batch, one pass and one rounding; gradient to both."""
    vn = 8 if tokens.dtype == torch.bfloat16 else 4
    if tokens.is_cuda and tokens.dtype in (_f32, torch.bfloat16) and tokens.shape[2] % vn == 0 and rows.dtype == _f32:
        return _AddRows.apply(tokens, _c(rows))
    return tokens + rows.to(tokens.dtype)


class _ResidualDropout(torch.autograd.Function):
    """``identity + dropout(tokens)`` over token rows in one pass (the concat-rows kernel with an empty second part), backward: the mask is
    recomputed from the seed (ge_slice_rows_drop) — replaces native_dropout + add and native_dropout_backward."""

    @staticmethod
    def forward(ctx, tok, identity, p, seed):
        tok = _rows(tok)
        B, N, C = tok.shape
        res = _c(identity.to(tok.dtype))
        out = torch.empty(B, N, C, device=tok.device, dtype=tok.dtype)
        PROFILER.run(f'residual_dropout[{B}x{N}x{C} {_tag(tok)}]', 3 * tok.numel() * _es(tok), lambda: hip.check(hip.lib().ge_concat_rows_fwd(
            _raw_ptr(tok, 'tokens'), N, tok.stride(0), _raw_ptr(res, 'identity'), None, _raw_ptr(out, 'out'), B * N, C, 0, 1, p, seed,
            hip.dtype_code(tok), hip.stream()), 'ge_concat_rows_fwd'))
        ctx.meta = (B, N, C, p, seed, identity.dtype)
        return out

    @staticmethod
    def backward(ctx, d_out):
        B, N, C, p, seed, id_dtype = ctx.meta
        d_out = _c(d_out)
        d_tok = torch.empty_like(d_out)
        PROFILER.run(f'slice_rows_drop[{B}x{N}x{C} {_tag(d_out)}]', 2 * d_tok.numel() * _es(d_out), lambda: hip.check(
            hip.lib().ge_slice_rows_drop(_raw_ptr(d_out, 'd_out'), hip.ptr(d_tok), B * N, C, C, 0, p, seed, hip.dtype_code(d_out), hip.stream()),
            'ge_slice_rows_drop'))
        return d_tok, d_out.to(id_dtype), None, None


def residual_dropout(identity, tokens, p, seed=None):
    """``identity + F.dropout(tokens, p)`` for token matrices (B,N,C) in training mode (p > 0)."""
    vn = 8 if tokens.dtype == torch.bfloat16 else 4
    if not (tokens.is_cuda and tokens.dim() == 3 and tokens.dtype in (_f32, torch.bfloat16) and tokens.shape[2] % vn == 0 and identity.shape == tokens.shape
            and identity.dtype == tokens.dtype and 0.0 < p < 1.0):       # mixed dtypes: the eager composition and its type promotion
        return identity + torch.nn.functional.dropout(tokens, p, True)
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())          # host generator: follows torch.manual_seed, no device sync
    return _ResidualDropout.apply(tokens, identity, float(p), int(seed))


_POS_ROWS = {}


def _pos_rows(pos):
    """(1,C,H,W) cached positional map -> (H*W, C) fp32 rows, cached per source tensor."""
    key = (pos.data_ptr(), tuple(pos.shape))
    if key not in _POS_ROWS:
        if len(_POS_ROWS) > 64:
            _POS_ROWS.clear()
        _POS_ROWS[key] = pos.detach().to(_f32).flatten(2)[0].t().contiguous()
    return _POS_ROWS[key]


def tokens_from_map(fmap, pos=None):
    """(B,C,H,W) [+ pos (1,C,H,W) fp32] -> (B,H*W,C): ``fmap.flatten(2).transpose(1,2) + pos`` in one transposing pass;
    for a channels-last map the token matrix is a VIEW of it and only the position add is a kernel."""
    if _cl_ok(fmap):
        B, C, H, W = fmap.shape
        tok = fmap.permute(0, 2, 3, 1).reshape(B, H * W, C)
        return tok if pos is None else _AddRows.apply(tok, _pos_rows(pos))
    return _TokensFromMap.apply(fmap, pos)


class _ConcatTokensMap(torch.autograd.Function):

    @staticmethod
    def forward(ctx, tok, fmap, identity, tokens_first, p, seed):
        tok = _rows(tok)
        B, N, C = tok.shape
        _, Cm, H, W = fmap.shape
        assert H * W == N and fmap.shape[0] == B
        out = torch.empty(B, C + Cm, H, W, device=tok.device, dtype=tok.dtype)
        t0, m0 = (0, C) if tokens_first else (Cm, 0)
        res = None
        if identity is not None:
            res = _planes(identity.to(tok.dtype))
            assert tuple(res.shape) == (B, C, H, W)
        es = _es(tok)
        PROFILER.run(f'map_from_tokens[{B}x{C}x{N} {_tag(tok)}{" +res" if res is not None else ""}{" drop" if p > 0 else ""}]',
                     (2 + (res is not None)) * tok.numel() * es, lambda: hip.check(hip.lib().ge_map_from_tokens(
                         _raw_ptr(tok, 'tokens'), tok.stride(0), None if res is None else _raw_ptr(res, 'identity'),
                         0 if res is None else res.stride(0), hip.ptr(out) + t0 * N * es, out.stride(0), B, C, N, p, seed,
                         hip.dtype_code(tok), hip.stream()), 'ge_map_from_tokens'))
        out[:, m0:m0 + Cm].copy_(fmap)
        ctx.meta = (B, N, C, Cm, t0, m0, p, seed, identity is not None)
        return out

    @staticmethod
    def backward(ctx, d_out):
        B, N, C, Cm, t0, m0, p, seed, has_id = ctx.meta
        d_out = _c(d_out)
        es = _es(d_out)
        d_tok = torch.empty(B, N, C, device=d_out.device, dtype=d_out.dtype)
        PROFILER.run(f'tokens_from_map[{B}x{C}x{N} {_tag(d_out)}{" drop" if p > 0 else ""}]', 2 * d_tok.numel() * es,
                     lambda: hip.check(hip.lib().ge_tokens_from_map(
                         hip.ptr(d_out) + t0 * N * es, d_out.stride(0), None, hip.ptr(d_tok), N * C, B, C, N, p, seed,
                         hip.dtype_code(d_out), hip.stream()), 'ge_tokens_from_map'))
        d_slice = d_out[:, t0:t0 + C]
        return d_tok, d_out[:, m0:m0 + Cm], (d_slice if has_id else None), None, None, None


class _ConcatRows(torch.autograd.Function):
    """Channels-last ``concat_tokens_map``: rows of [dropout(tokens) + identity | fmap] written once (csrc/nhwc.hip)."""

    @staticmethod
    def forward(ctx, tok, fmap, identity, tokens_first, p, seed):
        tok = _rows(tok)
        B, N, C = tok.shape
        _, Cm, H, W = fmap.shape
        assert H * W == N and fmap.shape[0] == B
        fmap = _cl(fmap)
        res = None if identity is None else _cl(identity.to(tok.dtype))
        out = torch.empty((B, C + Cm, H, W), device=tok.device, dtype=tok.dtype, memory_format=_CL)
        PROFILER.run(f'concat_rows[{B}x{N}x({C}+{Cm}) {_tag(tok)}{" +res" if res is not None else ""}{" drop" if p > 0 else ""}]',
                     ((2 + (res is not None)) * tok.numel() + 2 * fmap.numel()) * _es(tok), lambda: hip.check(hip.lib().ge_concat_rows_fwd(
                         _raw_ptr(tok, 'tokens'), N, tok.stride(0), None if res is None else _raw_ptr(res, 'identity'), _raw_ptr(fmap, 'map'),
                         _raw_ptr(out, 'out'), B * N, C, Cm, int(tokens_first), p, seed, hip.dtype_code(tok), hip.stream()), 'ge_concat_rows_fwd'))
        ctx.meta = (B, N, C, Cm, tokens_first, p, seed, identity is not None)
        return out

    @staticmethod
    def backward(ctx, d_out):
        B, N, C, Cm, tokens_first, p, seed, has_id = ctx.meta
        d_out = _cl(d_out)
        t0, m0 = (0, C) if tokens_first else (Cm, 0)
        d_slice = d_out[:, t0:t0 + C]
        if p > 0:
            d_tok = torch.empty(B, N, C, device=d_out.device, dtype=d_out.dtype)
            PROFILER.run(f'slice_rows_drop[{B}x{N}x{C} {_tag(d_out)}]', 2 * d_tok.numel() * _es(d_out), lambda: hip.check(
                hip.lib().ge_slice_rows_drop(_raw_ptr(d_out, 'd_out'), hip.ptr(d_tok), B * N, C, C + Cm, t0, p, seed, hip.dtype_code(d_out),
                                             hip.stream()), 'ge_slice_rows_drop'))
        else:
            d_tok = d_slice.permute(0, 2, 3, 1).reshape(B, N, C)               # a strided view: the consumer packs it
        return d_tok, d_out[:, m0:m0 + Cm], (d_slice if has_id else None), None, None, None


def concat_tokens_map(tokens, fmap, identity=None, tokens_first=True, p_drop=0.0, seed=None):
    """``torch.cat([to_map(dropout(tokens)) + identity, fmap], 1)`` (or fmap first) with the token part transposed,
    dropped and added straight into the concat buffer.  tokens (B,H*W,C), fmap (B,Cm,H,W), identity (B,C,H,W)."""
    p_drop = float(p_drop)
    if p_drop > 0.0 and seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())          # host generator: follows torch.manual_seed, no device sync
    vn = 8 if tokens.dtype == torch.bfloat16 else 4
    if (_cl_ok(fmap) and tokens.shape[2] % vn == 0 and tokens.is_cuda and tokens.dtype == fmap.dtype and tokens.stride(2) == 1
            and tokens.stride(1) == tokens.shape[2] and tokens.stride(0) % vn == 0):
        return _ConcatRows.apply(tokens, fmap, identity, bool(tokens_first), p_drop, int(seed or 0))
    return _ConcatTokensMap.apply(tokens, fmap, identity, bool(tokens_first), p_drop, int(seed or 0))


# -------------------------------------------------------------------------------- bilinear
class _Bilinear(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, Ho, Wo, align_corners):
        N, C, Hi, Wi = x.shape
        ctx.cl = _cl_ok(x)
        ctx.geom = (N, C, Hi, Wi, Ho, Wo, int(align_corners))
        if ctx.cl:
            out = torch.empty((N, C, Ho, Wo), device=x.device, dtype=x.dtype, memory_format=_CL)
            PROFILER.run(f'bilinear_nhwc_fwd[{N}x{C} {Hi}x{Wi}->{Ho}x{Wo} {_tag(x)}]', (x.numel() + out.numel()) * _es(x), lambda: hip.check(
                hip.lib().ge_bilinear_nhwc_fwd(_raw_ptr(x, 'input'), _raw_ptr(out, 'out'), N, C, Hi, Wi, Ho, Wo, int(align_corners),
                                               hip.dtype_code(x), hip.stream()), 'ge_bilinear_nhwc_fwd'))
            return out
        x = _c(x)
        out = torch.empty(N, C, Ho, Wo, device=x.device, dtype=x.dtype)
        PROFILER.run(f'bilinear_fwd[{N}x{C} {Hi}x{Wi}->{Ho}x{Wo} {_tag(x)}]', (x.numel() + out.numel()) * _es(x), lambda: hip.check(
            hip.lib().ge_bilinear_fwd(hip.ptr(x, name='input'), hip.ptr(out), N, C, Hi, Wi, Ho, Wo, int(align_corners),
                                      hip.dtype_code(x), hip.stream()), 'ge_bilinear_fwd'))
        return out

    @staticmethod
    def backward(ctx, d_out):
        N, C, Hi, Wi, Ho, Wo, ac = ctx.geom
        if ctx.cl:
            d_out = _cl(d_out)
            d_in = torch.empty((N, C, Hi, Wi), device=d_out.device, dtype=d_out.dtype, memory_format=_CL)
            ws = torch.empty(N * Ho * Wi * C, device=d_out.device, dtype=_f32) if (Ho > 3 * Hi or Wo > 3 * Wi) else None
            PROFILER.run(f'bilinear_nhwc_bwd[{N}x{C} {Hi}x{Wi}<-{Ho}x{Wo} {_tag(d_out)}]', (d_out.numel() + d_in.numel()) * _es(d_out),
                         lambda: hip.check(hip.lib().ge_bilinear_nhwc_bwd(_raw_ptr(d_out, 'd_out'), _raw_ptr(d_in, 'd_in'), hip.ptr(ws),
                                                                          0 if ws is None else ws.numel() * 4, N, C, Hi, Wi, Ho, Wo, ac,
                                                                          hip.dtype_code(d_out), hip.stream()), 'ge_bilinear_nhwc_bwd'))
            return d_in, None, None, None
        d_out = _c(d_out)
        d_in = torch.empty(N, C, Hi, Wi, device=d_out.device, dtype=d_out.dtype)
        PROFILER.run(f'bilinear_bwd[{N}x{C} {Hi}x{Wi}<-{Ho}x{Wo} {_tag(d_out)}]', (d_out.numel() + d_in.numel()) * _es(d_out),
                     lambda: hip.check(hip.lib().ge_bilinear_bwd(hip.ptr(d_out), hip.ptr(d_in), N, C, Hi, Wi, Ho, Wo, ac,
                                                                 hip.dtype_code(d_out), hip.stream()), 'ge_bilinear_bwd'))
        return d_in, None, None, None


def bilinear_resize(x, size, align_corners=False):
    """F.interpolate(x, size=size, mode='bilinear', align_corners=...) on the HIP kernel."""
    Ho, Wo = int(size[0]), int(size[1])
    if x.shape[2] == Ho and x.shape[3] == Wo:
        return x
    return _Bilinear.apply(x, Ho, Wo, bool(align_corners))


# ------------------------------------------------------------------------ 3x3 convolution on MFMA (csrc/conv3x3.hip)
def _conv3x3_launch(x, w_ohwi, bias, act, slope):
    N, C, H, W = x.shape
    Co = w_ohwi.shape[0]
    y = torch.empty((N, Co, H, W), device=x.device, dtype=torch.bfloat16, memory_format=_CL)
    flops = 2 * N * H * W * C * Co * 9
    PROFILER.run(f'conv3x3[{N}x{C}->{Co} {H}x{W}{" +b" if bias is not None else ""}{" act" if act else ""}]',
                 (x.numel() + y.numel() + w_ohwi.numel()) * 2, lambda: hip.check(hip.lib().ge_conv3x3_nhwc_fwd(
                     _raw_ptr(x, 'x'), hip.ptr(w_ohwi, torch.bfloat16), hip.ptr(bias, _f32), _raw_ptr(y, 'y'), N, H, W, C, Co, int(act), float(slope),
                     hip.GE_BF16, hip.stream()), 'ge_conv3x3_nhwc_fwd'), flops=flops)
    return y


class _Conv3x3(torch.autograd.Function):
    """3x3 / stride 1 / pad 1 convolution (+ bias, + leaky-ReLU) on the hand-written MFMA kernels: forward, data gradient (the same
    kernel on the flipped / transposed weights) and weight gradient (csrc/conv3x3_wgrad.hip, fp32 accumulation).  bf16 channels-last
    maps; the fp32 master weight is read through a bf16 copy."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, slope):
        x = _cl(x.to(torch.bfloat16))
        from .mmrt.optim import lowp
        wb = lowp(weight, torch.bfloat16).detach()                             # the optimizer's bf16 shadow when current: no cast kernel
        if not wb.is_contiguous(memory_format=_CL):
            wb = wb.contiguous(memory_format=_CL)
        w_ohwi = wb.permute(0, 2, 3, 1)                                      # a view: the channels-last storage IS (O, H, W, I)
        b32 = None if bias is None else _c(bias.detach().to(_f32))
        y = _conv3x3_launch(x, w_ohwi, b32, act, slope)
        ctx.save_for_backward(x, wb, y if act else None)
        ctx.meta = (act, slope, bias is not None, weight.dtype, None if bias is None else bias.dtype)
        ctx.weight_ref = weight if isinstance(weight, torch.nn.Parameter) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wb, y = ctx.saved_tensors
        act, slope, has_bias, w_dtype, b_dtype = ctx.meta
        N, Co, H, W = dy.shape
        dy = _cl(dy.to(torch.bfloat16))
        db = None
        with torch.autocast('cuda', enabled=False):
            if act:                                                             # d(pre-activation) and the bias gradient in one pass
                dyp = torch.empty_like(dy)
                dbf = torch.empty(Co, device=dy.device, dtype=_f32)
                ws = torch.empty(int(hip.lib().ge_nhwc_workspace(Co, 1)), device=dy.device, dtype=torch.uint8)
                PROFILER.run(f'bias_act_nhwc_bwd[{N}x{Co}x{H}x{W} bf16]', 3 * dy.numel() * 2, lambda: hip.check(hip.lib().ge_bias_act_nhwc_bwd(
                    _raw_ptr(dy, 'dy'), _raw_ptr(y, 'y'), _raw_ptr(dyp, 'dx'), hip.ptr(dbf), hip.ptr(ws), N * H * W, Co, slope, hip.GE_BF16,
                    hip.stream()), 'ge_bias_act_nhwc_bwd'))
                dy = dyp
                db = dbf if has_bias else None
            elif has_bias and ctx.needs_input_grad[2]:
                db = colsum(dy.permute(0, 2, 3, 1).reshape(N * H * W, Co))
            dx = dw = None
            if ctx.needs_input_grad[0]:
                wt = wb.permute(1, 2, 3, 0).flip(1, 2).contiguous()             # (I, 3, 3, O): w'[ci, r, s, co] = w[co, 2 - r, 2 - s, ci]
                dx = _conv3x3_launch(dy, wt, None, 0, 1.0)
            if ctx.needs_input_grad[1]:
                Ci = x.shape[1]
                # measured against MIOpen's wrw kernels (tools/ubench/conv_time.py, CONV_SWINL=1 for the 2-image Swin-L shapes;
                # profiles/r4_conv_time.txt).  Round 4 (K split = one resident round, staging indices hoisted out of the tile loop, operand
                # reads software-pipelined): 576 -> 64 @176x560 x 8: 630 vs 1322 us, 608 -> 96 @88x280: 302 vs 446, 704 -> 192 @44x140: 165 vs
                # 245, 1152 -> 384 @22x70: 167 vs 190; it still loses on the 11 x 35 maps (1280 -> 768: 151 vs 118: 32 pixel tiles for 480
                # output blocks).  At 2 images per GPU (config #3) MIOpen's kernels lose their batch parallelism: ours wins 15 of 17 layers
                # by 1.1 - 3.6x (576 -> 64 @176x560: 162 vs 489 us)
                px, gflop = N * H * W, 18e-9 * N * H * W * Ci * Co
                if 'conv3x3_wgrad' not in DISABLED and w_dtype == _f32 and (px >= 10000 or (N <= 4 and px >= 2000 and gflop >= 25.0)):
                    # MFMA weight gradient, fp32 accumulation straight into an (O, H, W, I) tensor = a channels-last (O, I, 3, 3) gradient
                    from .mmrt.optim import grad_target_ohwi
                    dw_ohwi = grad_target_ohwi(ctx.weight_ref) if ctx.weight_ref is not None else None      # the arena slice itself: no copy later
                    if dw_ohwi is None or tuple(dw_ohwi.shape) != (Co, 3, 3, Ci):
                        dw_ohwi = torch.empty(Co, 3, 3, Ci, device=dy.device, dtype=_f32)
                    dw_ohwi.zero_()
                    PROFILER.run(f'conv3x3_wgrad[{N}x{Ci}->{Co} {H}x{W}]', (x.numel() + dy.numel()) * 2 + dw_ohwi.numel() * 4, lambda: hip.check(
                        hip.lib().ge_conv3x3_nhwc_wgrad(_raw_ptr(x, 'x'), _raw_ptr(dy, 'dy'), hip.ptr(dw_ohwi), N, H, W, Ci, Co, hip.GE_BF16, hip.stream()),
                        'ge_conv3x3_nhwc_wgrad'), flops=2 * N * H * W * Ci * Co * 9)
                    dw = dw_ohwi.permute(0, 3, 1, 2)
                else:
                    dw = torch.ops.aten.convolution_backward(dy, x, wb, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False))[1]
                    from .mmrt.optim import grad_into_arena
                    dw = grad_into_arena(ctx.weight_ref, dw, w_dtype)
        return dx, dw, (None if db is None else db.to(b_dtype)), None, None


def conv3x3_ok(conv, x):
    """The MFMA convolution applies: plain 3x3 / stride 1 / pad 1 nn.Conv2d, bf16 execution (bf16 input or bf16 autocast), channel
    counts the kernel and its data-gradient instance take (multiples of 32 both ways)."""
    if 'conv3x3' in DISABLED or type(conv) is not torch.nn.Conv2d or not x.is_cuda or x.dim() != 4:
        return False
    if conv.kernel_size != (3, 3) or conv.stride != (1, 1) or conv.padding != (1, 1) or conv.dilation != (1, 1) or conv.groups != 1:
        return False
    if conv.padding_mode != 'zeros' or conv.in_channels % 32 or conv.out_channels % 32:
        return False
    bf16 = x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16)
    # channels-last execution only (depth.models.utils.to_channels_last): an NCHW model keeps its layout end to end on the library path
    # grid.z of the forward launch AND of the data-gradient launch (the same kernel with the channel roles swapped)
    nt = max((conv.out_channels + 63) // 64, (conv.in_channels + 63) // 64)
    return bf16 and _is_cl(x) and x.shape[0] * nt <= 65535


def conv3x3(conv, x, bias=None, act=False, slope=1.0):
    """``act(conv(x) + bias)`` for a 3x3 nn.Conv2d module on the MFMA kernel (``conv3x3_ok`` must hold)."""
    return _Conv3x3.apply(x, conv.weight, bias, bool(act), float(slope))


class _ConvLib(torch.autograd.Function):
    """Library (MIOpen / CK) convolution without bias under bf16 autocast, with the weight read from the optimizer's bf16 shadow arena
    (mmrt.optim.lowp) instead of autocast's per-step cast kernel; backward = the library's data / weight gradients, the latter widened to
    the master dtype."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding, dilation, groups):
        from .mmrt.optim import lowp
        dt = torch.get_autocast_dtype('cuda')
        with torch.autocast('cuda', enabled=False):
            xc, wc = x.to(dt), lowp(weight, dt).detach()
            y = torch.nn.functional.conv2d(xc, wc, None, stride, padding, dilation, groups)
        ctx.save_for_backward(xc, wc)
        ctx.meta = (stride, padding, dilation, groups, x.dtype, weight.dtype)
        ctx.weight_ref = weight if isinstance(weight, torch.nn.Parameter) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, wc = ctx.saved_tensors
        stride, padding, dilation, groups, x_dtype, w_dtype = ctx.meta
        dy = dy.to(wc.dtype)
        own_dw = ctx.needs_input_grad[1] and conv1x1_wgrad_ok(xc, dy, wc, stride, padding, dilation, groups)
        with torch.autocast('cuda', enabled=False):
            dx, dw, _ = torch.ops.aten.convolution_backward(dy, xc, wc, None, stride, padding, dilation, False, (0, 0), groups,
                                                            (ctx.needs_input_grad[0], ctx.needs_input_grad[1] and not own_dw, False))
        if own_dw:
            dw = conv1x1_wgrad(xc, dy).view(wc.shape)
        if dw is not None:
            from .mmrt.optim import grad_into_arena
            dw = grad_into_arena(ctx.weight_ref, dw, w_dtype)
        return (None if dx is None else dx.to(x_dtype)), dw, None, None, None, None


def conv1x1_wgrad_ok(x, dy, w, stride, padding, dilation, groups):
    """The streaming MFMA weight gradient of a 1x1 convolution applies (csrc/conv1x1_wgrad.hip): bf16 channels-last maps, plain 1x1 / stride 1
    geometry, channel counts the kernel tiles (Cin a multiple of 64 or 96, Cout of 32), enough rows to stream.
    OPT-IN (``GE_ENABLE=conv1x1_wgrad``): measured on MI355X it ties the library on the one large problem (64 -> 512 @176x560 x 8: 193 vs 180 us,
    both at the HBM rate) and loses on the small ones (zero fill + atomic flush + one round of 256 workgroups for 25 - 60 us of work): 687 vs
    498 us over the ten 1x1 layers, step 50.03 vs 49.77 ms same-session (profiles/r4_conv1x1_wgrad_time.txt) — so the default stays MIOpen / CK."""
    if 'conv1x1_wgrad' not in ENABLED or 'conv1x1_wgrad' in DISABLED or tuple(w.shape[2:]) != (1, 1) or groups != 1:
        return False
    if tuple(stride) != (1, 1) or tuple(padding) != (0, 0) or tuple(dilation) != (1, 1):
        return False
    Co, Ci = w.shape[:2]
    if x.dtype != torch.bfloat16 or dy.dtype != torch.bfloat16 or Co % 32 or (Ci % 64 and Ci % 96):
        return False
    return (x.is_cuda and _is_cl(x) and _is_cl(dy) and x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0
            and x.shape[0] * x.shape[2] * x.shape[3] >= 2048)


def conv1x1_wgrad(x, dy):
    """dW (Cout, Cin) fp32 = sum over the N H W pixels of dy x^T for channels-last bf16 maps x (N, Cin, H, W), dy (N, Cout, H, W)."""
    N, Ci, H, W = x.shape
    Co = dy.shape[1]
    M = N * H * W
    dw = torch.zeros(Co, Ci, device=x.device, dtype=_f32)
    PROFILER.run(f'conv1x1_wgrad[{N}x{Ci}->{Co} {H}x{W}]', (x.numel() + dy.numel()) * 2 + dw.numel() * 4, lambda: hip.check(
        hip.lib().ge_conv1x1_nhwc_wgrad(_raw_ptr(x, 'x'), _raw_ptr(dy, 'dy'), hip.ptr(dw), M, Ci, Co, hip.GE_BF16, hip.stream()),
        'ge_conv1x1_nhwc_wgrad'), flops=2 * M * Ci * Co)
    return dw


# rows (N H W) up to which a 1x1 convolution runs as token GEMMs: measured same-session on MI355X — every 1x1 layer through the GEMM path costs the
# 8-image step +0.3 ms (MIOpen's tuned implicit-GEMM kernels win on the 2e5 - 8e5-row maps) and saves the 2-image Swin-L step 0.33 ms (on maps
# of 770 - 49 280 rows MIOpen spends 50 - 110 us per layer backward for 10 - 30 us of work)
_CONV1X1_GEMM_ROWS = int(os.environ.get('GE_CONV1X1_GEMM_ROWS', '65536'))


class _Conv1x1Gemm(torch.autograd.Function):
    """A bias-free 1x1 / stride-1 convolution of a channels-last bf16 map as the token GEMM it is (rows = N H W): forward and data gradient
    through the token-Linear path of mmrt.bricks (tuned hipBLASLt solution or ge_gemm_nt), weight gradient split-K / plain with fp32
    accumulators written into the gradient arena — instead of MIOpen's implicit-GEMM kernels, their zero-fill launches and the widening copy of
    their bf16 weight gradient (the lateral / trans_proj blocks of the HAHI neck, reference necks/hahi.py:120-137)."""

    @staticmethod
    def forward(ctx, x, weight):
        from .mmrt import bricks
        from .mmrt.optim import lowp
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        wc = lowp(weight, torch.bfloat16).detach().reshape(Cout, Cin)
        x2 = x.permute(0, 2, 3, 1).reshape(B * H * W, Cin)
        with torch.autocast('cuda', enabled=False):
            y2 = bricks._linear_fwd(x2, wc, None, None, torch.bfloat16)
        ctx.save_for_backward(x, wc)
        ctx.w_dtype = weight.dtype
        ctx.weight_ref = weight if isinstance(weight, torch.nn.Parameter) else None
        return y2.view(B, H, W, Cout).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        from .mmrt import bricks
        from .mmrt.optim import grad_target
        x, wc = ctx.saved_tensors
        B, Cin, H, W = x.shape
        Cout = wc.shape[0]
        rows = B * H * W
        dy2 = _cl(dy.to(torch.bfloat16)).permute(0, 2, 3, 1).reshape(rows, Cout)
        x2 = x.permute(0, 2, 3, 1).reshape(rows, Cin)
        dx = dw = None
        with torch.autocast('cuda', enabled=False):
            if ctx.needs_input_grad[0]:
                dx = bricks._linear_dx(dy2, wc).view(B, H, W, Cin).permute(0, 3, 1, 2)
            if ctx.needs_input_grad[1]:
                # two aliases of the parameter's arena slice: the (Cout, Cin) one the GEMM writes, the parameter-shaped one autograd adopts
                tgt = grad_target(ctx.weight_ref) if (ctx.weight_ref is not None and ctx.w_dtype == _f32) else None
                out2 = tgt.view(Cout, Cin) if tgt is not None else None
                splits = bricks._split_k(rows)
                if splits:
                    part = torch.bmm(dy2.view(splits, rows // splits, Cout).transpose(1, 2), x2.view(splits, rows // splits, Cin))
                    dw2 = torch.sum(part, 0, dtype=_f32, out=out2) if out2 is not None else part.sum(0, dtype=_f32)
                else:
                    dw2 = torch.mm(dy2.t(), x2, out_dtype=_f32, out=out2) if out2 is not None else torch.mm(dy2.t(), x2, out_dtype=_f32)
                dw = tgt if tgt is not None else dw2.view(Cout, Cin, 1, 1).to(ctx.w_dtype)
                del out2
        return dx, dw


def conv1x1_gemm_ok(conv, x):
    """A ConvModule's bias-free 1x1 convolution can run as a token GEMM (``_Conv1x1Gemm``): bf16 autocast training, channels-last bf16 input,
    plain 1x1 geometry, channel counts that are whole 16-byte vectors.  ``GE_DISABLE=conv1x1_gemm`` keeps MIOpen."""
    return ('conv1x1_gemm' not in DISABLED and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.bias is None and x.dtype == torch.bfloat16 and _is_cl(x) and conv.in_channels % 8 == 0
            and conv.out_channels % 8 == 0 and x.data_ptr() % 16 == 0
            and x.shape[0] * x.shape[2] * x.shape[3] <= _CONV1X1_GEMM_ROWS)


def conv_lib(conv, x):
    """``conv._conv_forward(x, conv.weight, None)`` (no bias) — through ``_ConvLib`` when bf16 autocast training applies, else unchanged."""
    if (x.is_cuda and type(conv) is torch.nn.Conv2d and conv.padding_mode == 'zeros' and torch.is_grad_enabled() and conv.weight.requires_grad
            and torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16 and conv.weight.dtype == _f32
            and not isinstance(conv.padding, str) and 'conv_lib' not in DISABLED):
        if conv1x1_gemm_ok(conv, x):
            return _Conv1x1Gemm.apply(x, conv.weight)
        return _ConvLib.apply(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)
    return conv._conv_forward(x, conv.weight, None)


class _Conv3x3C1(torch.autograd.Function):
    """3x3 / stride 1 / pad 1 convolution to ONE output channel (+ bias) on the streaming kernels of csrc/conv3x3_c1.hip: forward, and a
    single backward pass for dx, dw and db.  The fp32 master weight is read directly (rounded to bf16 in the kernel, as autocast's cast
    does): no cast, flip or zero-fill kernels around it."""

    @staticmethod
    def forward(ctx, x, weight, bias, out_fp32):
        x = _cl(x.to(torch.bfloat16))
        N, C, H, W = x.shape
        w = weight.detach().permute(0, 2, 3, 1)                              # (1, 3, 3, I): a view of a channels-last weight
        if not w.is_contiguous():
            w = w.contiguous()
        b = None if bias is None else bias.detach()
        y = torch.empty((N, 1, H, W), device=x.device, dtype=_f32 if out_fp32 else torch.bfloat16)
        PROFILER.run(f'conv3x3_c1[{N}x{C}->1 {H}x{W}]', x.numel() * 2 + y.numel() * y.element_size(), lambda: hip.check(hip.lib().ge_conv3x3_c1_fwd(
            _raw_ptr(x, 'x'), hip.ptr(w, _f32), hip.ptr(b, _f32), hip.ptr(y), N, H, W, C, hip.GE_F32 if out_fp32 else hip.GE_BF16, hip.stream()),
            'ge_conv3x3_c1_fwd'))
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        N, C, H, W = x.shape
        if dy.dtype not in (_f32, torch.bfloat16):
            dy = dy.float()
        dy = _c(dy)
        dx = torch.empty_like(x)
        dw = torch.empty((1, 3, 3, C), device=x.device, dtype=_f32)
        db = torch.empty(1, device=x.device, dtype=_f32) if ctx.has_bias else None
        PROFILER.run(f'conv3x3_c1_bwd[{N}x{C}->1 {H}x{W}]', 2 * x.numel() * 2 + dy.numel() * dy.element_size(), lambda: hip.check(hip.lib().ge_conv3x3_c1_bwd(
            _raw_ptr(x, 'x'), hip.ptr(dy), hip.ptr(w, _f32), _raw_ptr(dx, 'dx'), hip.ptr(dw), hip.ptr(db, _f32), N, H, W, C,
            hip.GE_F32 if dy.dtype == _f32 else hip.GE_BF16, hip.stream()), 'ge_conv3x3_c1_bwd'))
        return dx, dw.permute(0, 3, 1, 2), db, None


def conv3x3_c1_ok(conv, x):
    """The one-output-channel streaming convolution applies (3x3 / s1 / p1, bf16 channels-last execution, fp32 master weights)."""
    if 'conv3x3_c1' in DISABLED or type(conv) is not torch.nn.Conv2d or not x.is_cuda or x.dim() != 4 or conv.out_channels != 1:
        return False
    if conv.kernel_size != (3, 3) or conv.stride != (1, 1) or conv.padding != (1, 1) or conv.dilation != (1, 1) or conv.groups != 1:
        return False
    if conv.padding_mode != 'zeros' or conv.in_channels % 8 or conv.in_channels > 1024 or conv.weight.dtype != _f32:
        return False
    bf16 = x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16)
    return bf16 and _is_cl(x)


def conv3x3_c1(conv, x, out_fp32=False):
    """``conv(x)`` (bias included) for a 3x3 nn.Conv2d with one output channel (``conv3x3_c1_ok`` must hold); ``out_fp32``: the result in
    fp32 instead of the autocast dtype (for a consumer that would cast it up anyway)."""
    return _Conv3x3C1.apply(x, conv.weight, conv.bias, bool(out_fp32))


# -------------------------------------------------------------- decoder glue: up-sample + concat, sum of up-sampled maps
class _UpCat(torch.autograd.Function):

    @staticmethod
    def forward(ctx, coarse, skip, align_corners):
        N, Cu, Hc, Wc = coarse.shape
        _, Cs, H, W = skip.shape
        coarse, skip = _cl(coarse), _cl(skip)
        out = torch.empty((N, Cu + Cs, H, W), device=coarse.device, dtype=coarse.dtype, memory_format=_CL)
        PROFILER.run(f'upcat_fwd[{N}x({Cu}^+{Cs}) {Hc}x{Wc}->{H}x{W} {_tag(coarse)}]', (coarse.numel() + skip.numel() + out.numel()) * _es(out),
                     lambda: hip.check(hip.lib().ge_upcat_nhwc_fwd(_raw_ptr(coarse, 'coarse'), _raw_ptr(skip, 'skip'), _raw_ptr(out, 'out'), N, Cu, Hc, Wc,
                                                                    Cs, H, W, int(align_corners), hip.dtype_code(out), hip.stream()), 'ge_upcat_nhwc_fwd'))
        ctx.geom = (N, Cu, Hc, Wc, Cs, H, W, int(align_corners))
        return out

    @staticmethod
    def backward(ctx, d_out):
        N, Cu, Hc, Wc, Cs, H, W, ac = ctx.geom
        d_out = _cl(d_out)
        d_coarse = torch.empty((N, Cu, Hc, Wc), device=d_out.device, dtype=d_out.dtype, memory_format=_CL)
        PROFILER.run(f'upcat_bwd[{N}x{Cu} {Hc}x{Wc}<-{H}x{W} {_tag(d_out)}]', (N * H * W * Cu + d_coarse.numel()) * _es(d_out),
                     lambda: hip.check(hip.lib().ge_upcat_nhwc_bwd(_raw_ptr(d_out, 'd_out'), _raw_ptr(d_coarse, 'd_coarse'), N, Cu, Hc, Wc, Cs, H, W, ac,
                                                                    hip.dtype_code(d_out), hip.stream()), 'ge_upcat_nhwc_bwd'))
        return d_coarse, d_out[:, Cu:], None


def upcat(coarse, skip, align_corners=True):
    """``torch.cat([F.interpolate(coarse, size=skip.shape[2:], mode='bilinear', align_corners=...), skip], 1)`` in one pass on channels-last
    maps (csrc/decoder.hip); the plain composition of the HIP bilinear kernel and ATen's cat for other layouts / channel counts."""
    vn = 8 if skip.dtype == torch.bfloat16 else 4
    if ('upcat' not in DISABLED and _cl_ok(skip) and coarse.is_cuda and coarse.dtype == skip.dtype and coarse.shape[1] % vn == 0 and coarse.shape[1] > 1
            and coarse.shape[0] * skip.shape[2] <= 65535 and tuple(coarse.shape[2:]) != tuple(skip.shape[2:])):
        return _UpCat.apply(coarse, skip, bool(align_corners))
    return torch.cat([bilinear_resize(coarse, skip.shape[2:], align_corners), skip], 1)


class _UpSum(torch.autograd.Function):

    @staticmethod
    def forward(ctx, align_corners, fine, *srcs):
        N, C, H, W = fine.shape
        fine = _cl(fine)
        srcs = [_cl(t) for t in srcs]
        out = torch.empty_like(fine)
        ptrs = (ctypes.c_void_p * len(srcs))(*[_raw_ptr(t, 'src') for t in srcs])
        hw = (ctypes.c_int * (2 * len(srcs)))(*[v for t in srcs for v in t.shape[2:]])
        PROFILER.run(f'upsum_fwd[{N}x{C}x{H}x{W} <- {len(srcs)} maps {_tag(fine)}]', (2 * fine.numel() + sum(t.numel() for t in srcs)) * _es(fine),
                     lambda: hip.check(hip.lib().ge_upsum_nhwc_fwd(ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(hw, ctypes.c_void_p), len(srcs),
                                                                    _raw_ptr(fine, 'fine'), _raw_ptr(out, 'out'), N, C, H, W, int(align_corners),
                                                                    hip.dtype_code(fine), hip.stream()), 'ge_upsum_nhwc_fwd'))
        ctx.geom = (N, C, H, W, int(align_corners), [tuple(t.shape[2:]) for t in srcs])
        return out

    @staticmethod
    def backward(ctx, d_out):
        N, C, H, W, ac, sizes = ctx.geom
        d_out = _cl(d_out)
        grads = []
        for (Hi, Wi) in sizes:                               # the transpose of each interpolation reads the same d_out
            d_in = torch.empty((N, C, Hi, Wi), device=d_out.device, dtype=d_out.dtype, memory_format=_CL)
            ws = torch.empty(N * H * Wi * C, device=d_out.device, dtype=_f32) if (H > 3 * Hi or W > 3 * Wi) else None
            PROFILER.run(f'bilinear_nhwc_bwd[{N}x{C} {Hi}x{Wi}<-{H}x{W} {_tag(d_out)}]', (d_out.numel() + d_in.numel()) * _es(d_out),
                         lambda: hip.check(hip.lib().ge_bilinear_nhwc_bwd(_raw_ptr(d_out, 'd_out'), _raw_ptr(d_in, 'd_in'), hip.ptr(ws),
                                                                          0 if ws is None else ws.numel() * 4, N, C, Hi, Wi, H, W, ac,
                                                                          hip.dtype_code(d_out), hip.stream()), 'ge_bilinear_nhwc_bwd'))
            grads.append(d_in)
        return (None, d_out) + tuple(grads)


def upsum(fine, coarse_maps, align_corners=True):
    """``((up(c0) + up(c1)) + ...) + fine`` with ``up = F.interpolate(., size=fine.shape[2:], bilinear)``: one pass over the fine map
    (csrc/decoder.hip) instead of one up-sampled tensor and one add per coarse map."""
    vn = 8 if fine.dtype == torch.bfloat16 else 4
    ok = ('upsum' not in DISABLED and _cl_ok(fine) and 1 <= len(coarse_maps) <= 4 and fine.shape[0] * fine.shape[2] <= 65535
          and all(t.is_cuda and t.dtype == fine.dtype and t.shape[1] == fine.shape[1] and t.shape[1] % vn == 0 and tuple(t.shape[2:]) != tuple(fine.shape[2:])
                  for t in coarse_maps))
    if not ok:
        acc = None
        for t in coarse_maps:
            t = bilinear_resize(t, fine.shape[2:], align_corners)
            acc = t if acc is None else acc + t
        return fine if acc is None else acc + fine
    return _UpSum.apply(bool(align_corners), fine, *coarse_maps)


# ------------------------------------------------------------------------------ layer norm
_LN_COPIES = max(1, min(64, int(os.environ.get('GE_LN_COPIES', '8'))))      # accumulators of the d_gamma / d_beta column sums (same-address fp32 atomics serialise in L2)


_LN_ACC = {}


def _ln_accumulators(device, C):
    """The (copies, 2, C) fp32 accumulators of a LayerNorm backward: one persistent buffer per (device, stream, width), zero on entry — the fold
    kernel that sums the copies clears them again (ge_layernorm_fold), so no zero-fill launch per layer.  Launches are stream-ordered.
    A buffer first needed while a stream capture is under way would come from the graph's private pool: it is used for that call only and
    never cached (eager code must not see graph-pool memory)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream, C)
    buf = _LN_ACC.get(key)
    if buf is None:
        buf = torch.zeros(_LN_COPIES, 2, C, device=device, dtype=_f32)
        if not torch.cuda.is_current_stream_capturing():
            _LN_ACC[key] = buf
    return buf


def _ln_bwd_and_fold(launch_multi, dwb, C):
    """accumulate (ge_layernorm_bwd_multi) + fold-and-clear (ge_layernorm_fold) as one unit: if anything fails in between (a launch error, a
    KeyboardInterrupt, an aborted capture) the accumulators are left dirty and every later LayerNorm backward of this width would silently add
    them to d_gamma / d_beta — so the cached buffer is dropped (the next call allocates a zeroed one)."""
    try:
        launch_multi()
        return _ln_fold(dwb, C)
    except BaseException:
        for k in [k for k, v in _LN_ACC.items() if v is dwb]:
            del _LN_ACC[k]
        raise


def _ln_fold(dwb, C):
    out = torch.empty(2, C, device=dwb.device, dtype=_f32)
    hip.check(hip.lib().ge_layernorm_fold(hip.ptr(dwb), hip.ptr(out), _LN_COPIES, C, hip.stream()), 'ge_layernorm_fold')
    return out


class _LayerNorm(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        x = _c(x)
        C = x.shape[-1]
        rows = x.numel() // C
        w, b = _c(weight.detach().to(_f32)), _c(bias.detach().to(_f32))
        y = torch.empty(x.shape, device=x.device, dtype=out_dtype)
        mean = torch.empty(rows, device=x.device, dtype=_f32)
        rstd = torch.empty(rows, device=x.device, dtype=_f32)
        PROFILER.run(f'layernorm_fwd[{rows}x{C} {_tag(x)}->{_tag(y)}]', x.numel() * _es(x) + y.numel() * _es(y), lambda: hip.check(
            hip.lib().ge_layernorm_fwd(hip.ptr(x, name='x'), hip.dtype_code(x), hip.ptr(w), hip.ptr(b), hip.ptr(y),
                                       hip.dtype_code(y), hip.ptr(mean), hip.ptr(rstd), rows, C, eps, hip.stream()),
            'ge_layernorm_fwd'))
        ctx.save_for_backward(x, w, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        C = x.shape[-1]
        rows = x.numel() // C
        dy = _c(dy)
        if dy.dtype not in (_f32, torch.bfloat16):
            dy = dy.to(_f32)
        dx = torch.empty_like(x)
        dwb = _ln_accumulators(x.device, C)                                                    # see ge_layernorm_bwd_multi
        dwb = _ln_bwd_and_fold(lambda: PROFILER.run(f'layernorm_bwd[{rows}x{C} {_tag(x)}<-{_tag(dy)}]', 2 * x.numel() * _es(x) + dy.numel() * _es(dy), lambda: hip.check(
            hip.lib().ge_layernorm_bwd_multi(hip.ptr(dy), hip.dtype_code(dy), hip.ptr(x), hip.dtype_code(x), hip.ptr(w), hip.ptr(mean),
                                             hip.ptr(rstd), None, hip.ptr(dx), hip.ptr(dwb), _LN_COPIES, rows, C, hip.stream()),
            'ge_layernorm_bwd_multi')), dwb, C)
        return dx, dwb[0], dwb[1], None, None


class _LayerNormRes(torch.autograd.Function):
    """``(LN(x), x)`` for a pre-norm residual block: the second output is x itself, routed through this node so that the gradient arriving
    over the skip connection is added to LN'(dy) INSIDE the LayerNorm backward kernel (ge_layernorm_bwd_res) instead of by a separate
    autograd add over the token tensor (24 per Swin-T step)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        x = _c(x)
        C = x.shape[-1]
        rows = x.numel() // C
        w, b = _c(weight.detach().to(_f32)), _c(bias.detach().to(_f32))
        y = torch.empty(x.shape, device=x.device, dtype=out_dtype)
        mean = torch.empty(rows, device=x.device, dtype=_f32)
        rstd = torch.empty(rows, device=x.device, dtype=_f32)
        PROFILER.run(f'layernorm_fwd[{rows}x{C} {_tag(x)}->{_tag(y)}]', x.numel() * _es(x) + y.numel() * _es(y), lambda: hip.check(
            hip.lib().ge_layernorm_fwd(hip.ptr(x, name='x'), hip.dtype_code(x), hip.ptr(w), hip.ptr(b), hip.ptr(y),
                                       hip.dtype_code(y), hip.ptr(mean), hip.ptr(rstd), rows, C, eps, hip.stream()),
            'ge_layernorm_fwd'))
        ctx.save_for_backward(x, w, mean, rstd)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        x, w, mean, rstd = ctx.saved_tensors
        C = x.shape[-1]
        rows = x.numel() // C
        if dy is None:                                                          # LN output unused: only the skip gradient
            return dres, None, None, None, None
        dy = _c(dy)
        if dy.dtype not in (_f32, torch.bfloat16):
            dy = dy.to(_f32)
        if dres is not None:
            dres = _c(dres.to(x.dtype))
        dx = torch.empty_like(x)
        dwb = _ln_accumulators(x.device, C)
        dwb = _ln_bwd_and_fold(lambda: PROFILER.run(f'layernorm_bwd[{rows}x{C} {_tag(x)}<-{_tag(dy)}{" +res" if dres is not None else ""}]',
                     (2 + (dres is not None)) * x.numel() * _es(x) + dy.numel() * _es(dy), lambda: hip.check(
            hip.lib().ge_layernorm_bwd_multi(hip.ptr(dy), hip.dtype_code(dy), hip.ptr(x), hip.dtype_code(x), hip.ptr(w), hip.ptr(mean),
                                             hip.ptr(rstd), hip.ptr(dres), hip.ptr(dx), hip.ptr(dwb), _LN_COPIES, rows, C, hip.stream()),
            'ge_layernorm_bwd_multi')), dwb, C)
        return dx, dwb[0], dwb[1], None, None


def layer_norm_res(x, weight, bias, eps=1e-5, out_dtype=None):
    """-> (LayerNorm(x), x') with x' == x numerically: use x' for the skip connection of a pre-norm block (see _LayerNormRes)."""
    return _LayerNormRes.apply(x, weight, bias, float(eps), out_dtype or x.dtype)


_COLSUM_WS = {}


def colsum(x):
    """(R, C) f32 / bf16 -> (C,) f32 column sums (bias gradient of a token Linear) in one streaming pass."""
    R, C = x.shape
    vn = 4 if x.dtype == _f32 else 8
    if x.dtype not in (_f32, torch.bfloat16) or C % vn or not x.is_contiguous() or x.data_ptr() % 16:
        note_fallback('colsum', f'C={C} dtype={x.dtype}')
        return x.sum(0, dtype=_f32)
    key = (x.device, C, torch.cuda.current_stream(x.device).cuda_stream)
    ws = _COLSUM_WS.get(key)
    if ws is None:                                      # launches are stream-ordered: one workspace per (width, stream) is enough
        ws = _COLSUM_WS[key] = torch.empty(int(hip.lib().ge_nhwc_workspace(C, 1)), device=x.device, dtype=torch.uint8)
    out = torch.empty(C, device=x.device, dtype=_f32)
    PROFILER.run(f'colsum[{R}x{C} {_tag(x)}]', x.numel() * _es(x), lambda: hip.check(
        hip.lib().ge_colsum(hip.ptr(x, name='x'), R, C, hip.ptr(out), hip.ptr(ws), 0, hip.dtype_code(x), hip.stream()), 'ge_colsum'))
    return out


# ---- token GEMM (csrc/gemm.hip).  (K, N) -> (M_min, M_max) for which ge_gemm_nt beat the tuned hipBLASLt / rocBLAS solution on MI355X
# (tools/ubench/gemm_time.py, profiles/r4_gemm_time.txt: ratio >= 1.1 there; everything else keeps the library GEMM).  GE_GEMM=all routes
# every supported shape through the kernel (tests), GE_DISABLE=gemm none.
GEMM_OWN = {
    (96, 288): (60000, 400000),      # Swin stage-0 qkv: 61 vs 85 us at 197120 tokens
    (96, 384): (60000, 400000),      # stage-0 FFN fc1 and d(fc2): 63 vs 76
    (288, 96): (60000, 400000),      # stage-0 d(qkv): 62 vs 101
    (192, 192): (20000, 100000),     # stage-1 proj and d(proj): 15 vs 20 at 49280
    (384, 384): (6000, 25000),       # stage-2 proj and d(proj): 17 vs 20 at 12320
    (512, 512): (150000, 400000),    # HAHI value / output projections of the 261800-token self-attention: 160 vs 178
}
_GEMM_ALL = os.environ.get('GE_GEMM', '') == 'all'


def gemm_own(M, K, N):
    """Should F.linear of an (M, K) bf16 token matrix with an (N, K) weight run on ge_gemm_nt?"""
    if 'gemm' in DISABLED or K % 8 or N % 8 or M <= 0:
        return False
    if _GEMM_ALL:
        return (M * K + K) * 2 < 2 ** 32 and (N * K + K) * 2 < 2 ** 32
    r = GEMM_OWN.get((K, N))
    return r is not None and r[0] <= M <= r[1]


def gemm_nt(x2, w, bias=None):
    """x2 (M, K) bf16 contiguous, w (N, K) bf16 contiguous, bias (N,) f32 or None -> (M, N) bf16 = x2 w^T + bias: fp32 accumulation,
    bias added in fp32 before the single rounding (ge_gemm_nt).  Raises if the kernel refuses the problem: ask gemm_own first."""
    M, K = x2.shape
    N = w.shape[0]
    assert x2.dtype == w.dtype == torch.bfloat16 and w.shape[1] == K and x2.is_contiguous() and w.is_contiguous()
    if bias is not None:
        assert bias.dtype == _f32 and bias.numel() == N and bias.is_contiguous()
    out = torch.empty(M, N, device=x2.device, dtype=torch.bfloat16)
    PROFILER.run(f'gemm_nt[{M}x{K}->{N}]', 2 * (M * K + N * K + M * N), lambda: hip.check(hip.lib().ge_gemm_nt(
        hip.ptr(x2, name='x'), K, hip.ptr(w, name='weight'), K, hip.ptr(bias), hip.ptr(out), N, M, N, K, hip.GE_BF16, hip.stream()), 'ge_gemm_nt'),
        flops=2.0 * M * N * K)
    return out


def bias_gelu_fwd(y0, bias):
    """gelu(y0 + bias) over a (rows, C) matrix (the bias-free output of the FFN's first GEMM); bias fp32 (C) or None."""
    C = y0.shape[-1]
    R = y0.numel() // C
    out = torch.empty_like(y0)
    PROFILER.run(f'bias_gelu_fwd[{R}x{C} {_tag(y0)}]', 2 * y0.numel() * _es(y0), lambda: hip.check(hip.lib().ge_bias_gelu_fwd(
        hip.ptr(y0, name='y0'), hip.ptr(bias, _f32), hip.ptr(out), R, C, hip.dtype_code(y0), hip.stream()), 'ge_bias_gelu_fwd'))
    return out


def bias_gelu_bwd(dg, y0, bias):
    """-> (dy = dg * gelu'(y0 + bias) in the storage type, d_bias fp32 (C) = column sums of dy), one sweep."""
    C = y0.shape[-1]
    R = y0.numel() // C
    dy = torch.empty_like(y0)
    db = torch.empty(C, device=y0.device, dtype=_f32)
    key = (y0.device, C, torch.cuda.current_stream(y0.device).cuda_stream)
    ws = _COLSUM_WS.get(key)
    if ws is None:
        ws = _COLSUM_WS[key] = torch.empty(int(hip.lib().ge_nhwc_workspace(C, 1)), device=y0.device, dtype=torch.uint8)
    PROFILER.run(f'bias_gelu_bwd[{R}x{C} {_tag(y0)}]', 3 * y0.numel() * _es(y0), lambda: hip.check(hip.lib().ge_bias_gelu_bwd(
        hip.ptr(dg, name='dg'), hip.ptr(y0), hip.ptr(bias, _f32), hip.ptr(dy), hip.ptr(db), hip.ptr(ws), R, C, hip.dtype_code(y0),
        hip.stream()), 'ge_bias_gelu_bwd'))
    return dy, db


def layer_norm(x, weight, bias, eps=1e-5, out_dtype=None):
    """LayerNorm over the last dim; x f32/bf16, statistics f32, output ``out_dtype`` (default: x.dtype)."""
    return _LayerNorm.apply(x, weight, bias, float(eps), out_dtype or x.dtype)


# ---------------------------------------------------------------- residual + stochastic depth
class _ResidualDropPath(torch.autograd.Function):

    @staticmethod
    def forward(ctx, identity, branch, scale):
        identity, branch = _c(identity), _c(branch)
        B = identity.shape[0]
        n = identity.numel() // max(B, 1)
        out = torch.empty_like(identity)
        PROFILER.run(f'residual_drop_path[{B}x{n} {_tag(identity)}+{_tag(branch)}]',
                     2 * identity.numel() * _es(identity) + branch.numel() * _es(branch), lambda: hip.check(
            hip.lib().ge_residual_scale_add(hip.ptr(identity, name='identity'), hip.dtype_code(identity), hip.ptr(branch),
                                            hip.dtype_code(branch), hip.ptr(scale, _f32), hip.ptr(out), B, n, hip.stream()),
            'ge_residual_scale_add'))
        ctx.save_for_backward(scale)
        ctx.branch_dtype = branch.dtype
        return out

    @staticmethod
    def backward(ctx, dy):
        scale, = ctx.saved_tensors
        dy = _c(dy)
        B = dy.shape[0]
        n = dy.numel() // max(B, 1)
        d_branch = torch.empty(dy.shape, device=dy.device, dtype=ctx.branch_dtype)
        PROFILER.run(f'scale_rows[{B}x{n} {_tag(dy)}->{_tag(d_branch)}]', dy.numel() * _es(dy) + d_branch.numel() * _es(d_branch),
                     lambda: hip.check(hip.lib().ge_scale_rows(hip.ptr(dy), hip.dtype_code(dy), hip.ptr(scale), hip.ptr(d_branch),
                                                               hip.dtype_code(d_branch), B, n, hip.stream()), 'ge_scale_rows'))
        return dy, d_branch, None


def residual_drop_path(identity, branch, scale):
    """identity + branch * scale[b] (per-sample stochastic depth), in the identity's dtype."""
    return _ResidualDropPath.apply(identity, branch, scale)


# ------------------------------------------------------------------ batch norm (training) + ReLU
class _BNAct(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, slope):
        ctx.cl = _cl_ok(x)
        if not ctx.cl:
            x = _c(x)
        N, C, H, W = x.shape
        w, b = _c(weight.detach().to(_f32)), _c(bias.detach().to(_f32))
        y = torch.empty_like(x)                                  # preserves the channels-last strides
        stats = torch.empty(2, C, device=x.device, dtype=_f32)
        ws = torch.empty(int(hip.lib().ge_nhwc_workspace(C, 2) if ctx.cl else hip.lib().ge_bn_workspace(C)), device=x.device, dtype=torch.uint8)
        if ctx.cl:
            PROFILER.run(f'bn_act_nhwc_fwd[{N}x{C}x{H}x{W} {_tag(x)}]', 3 * x.numel() * _es(x), lambda: hip.check(
                hip.lib().ge_bn_act_nhwc_fwd(_raw_ptr(x, 'x'), hip.ptr(w), hip.ptr(b), _raw_ptr(y, 'y'), hip.ptr(stats[0]), hip.ptr(stats[1]),
                                             hip.ptr(running_mean, _f32), hip.ptr(running_var, _f32), hip.ptr(ws), N * H * W, C, eps,
                                             momentum, slope, hip.dtype_code(x), hip.stream()), 'ge_bn_act_nhwc_fwd'))
            ctx.save_for_backward(x, b, w, stats)            # not y: the backward recomputes the activation decision from x
            ctx.slope = slope
            return y
        PROFILER.run(f'bn_act_fwd[{N}x{C}x{H}x{W} {_tag(x)}]', 3 * x.numel() * _es(x), lambda: hip.check(
            hip.lib().ge_bn_act_fwd(hip.ptr(x, name='x'), hip.ptr(w), hip.ptr(b), hip.ptr(y), hip.ptr(stats[0]), hip.ptr(stats[1]),
                                    hip.ptr(running_mean, _f32), hip.ptr(running_var, _f32), hip.ptr(ws), N, C, H * W, eps,
                                    momentum, slope, hip.dtype_code(x), hip.stream()), 'ge_bn_act_fwd'))
        ctx.save_for_backward(x, y, w, stats)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, w, stats = ctx.saved_tensors                   # channels-last: the second entry is beta, not y
        N, C, H, W = x.shape
        dx = torch.empty_like(x)
        dwb = torch.empty(2, C, device=x.device, dtype=_f32)
        ws = torch.empty(int(hip.lib().ge_nhwc_workspace(C, 2) if ctx.cl else hip.lib().ge_bn_workspace(C)), device=x.device, dtype=torch.uint8)
        if ctx.cl:
            dy = _cl(dy.to(x.dtype))
            PROFILER.run(f'bn_act_nhwc_bwd[{N}x{C}x{H}x{W} {_tag(x)}]', 5 * x.numel() * _es(x), lambda: hip.check(
                hip.lib().ge_bn_act_nhwc_bwd(_raw_ptr(dy, 'dy'), None, _raw_ptr(x, 'x'), hip.ptr(w), hip.ptr(y, _f32), hip.ptr(stats[0]), hip.ptr(stats[1]),
                                             _raw_ptr(dx, 'dx'), hip.ptr(dwb[0]), hip.ptr(dwb[1]), hip.ptr(ws), N * H * W, C, ctx.slope,
                                             hip.dtype_code(x), hip.stream()), 'ge_bn_act_nhwc_bwd'))
            return dx, dwb[0], dwb[1], None, None, None, None, None
        dy = _c(dy.to(x.dtype))
        PROFILER.run(f'bn_act_bwd[{N}x{C}x{H}x{W} {_tag(x)}]', 7 * x.numel() * _es(x), lambda: hip.check(
            hip.lib().ge_bn_act_bwd(hip.ptr(dy), hip.ptr(y), hip.ptr(x), hip.ptr(w), hip.ptr(stats[0]), hip.ptr(stats[1]), hip.ptr(dx),
                                    hip.ptr(dwb[0]), hip.ptr(dwb[1]), hip.ptr(ws), N, C, H * W, ctx.slope, hip.dtype_code(x),
                                    hip.stream()), 'ge_bn_act_bwd'))
        return dx, dwb[0], dwb[1], None, None, None, None, None


def bn_act(x, bn, slope=0.0):
    """Training-mode ``nn.BatchNorm2d`` followed by leaky-ReLU(slope) (0 = ReLU, 1 = none) on an NCHW map; updates the
    module's running statistics and ``num_batches_tracked`` like ``F.batch_norm``."""
    y = _BNAct.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.eps), float(bn.momentum), float(slope))
    bn.num_batches_tracked.add_(1)
    return y


# ------------------------------------------------- 1x1 conv + batch norm (training) + ReLU + position add, one pass (csrc/conv1x1_bn.hip)
class _Conv1x1BNActPos(torch.autograd.Function):
    """``y = act(bn(conv1x1(x)))`` and ``q = tokens(y) + pos`` for a 64-channel channels-last bf16 map (HAHI ``conv_proj`` and the cross-attention
    query, reference necks/hahi.py:151-157,294-306).  BatchNorm statistics come from the input's Gram matrix, the pre-BN tensor is never stored;
    the backward takes the two gradients (through y — possibly a channel slice of a wider map, read in place — and through q) in one masking
    pass and reduces the BatchNorm backward to rank-64 corrections of the convolution's two gradient GEMMs."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, pos_rows, eps, momentum, slope):
        from .mmrt.optim import lowp
        lib = hip.lib()
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        HW, rows = H * W, B * H * W
        wc = lowp(weight, torch.bfloat16).detach().reshape(Cout, Cin)
        g32, b32 = _c(gamma.detach().to(_f32)), _c(beta.detach().to(_f32))
        dev = x.device
        ws = torch.empty(int(lib.ge_conv1x1_bn_workspace(Cin, Cout)), device=dev, dtype=torch.uint8)
        gram = torch.empty(65 * Cin, device=dev, dtype=torch.float64)
        stats = torch.empty(2, Cout, device=dev, dtype=_f32)
        coef = torch.empty(2, Cout, device=dev, dtype=_f32)
        y = torch.empty((B, Cout, H, W), device=dev, dtype=x.dtype, memory_format=_CL)
        q = torch.empty(B, HW, Cout, device=dev, dtype=x.dtype) if pos_rows is not None else None
        PROFILER.run(f'conv1x1_bn_stats[{B}x{Cin}->{Cout} {H}x{W}]', x.numel() * 2, lambda: hip.check(lib.ge_conv1x1_bn_stats(
            _raw_ptr(x, 'x'), rows, Cin, hip.ptr(wc), Cout, hip.ptr(g32), hip.ptr(b32), hip.ptr(running_mean, _f32), hip.ptr(running_var, _f32),
            eps, momentum, hip.ptr(gram), hip.ptr(stats[0]), hip.ptr(stats[1]), hip.ptr(coef), hip.ptr(ws), hip.stream()), 'ge_conv1x1_bn_stats'))
        PROFILER.run(f'conv1x1_bn_act_fwd[{B}x{Cin}->{Cout} {H}x{W}{" +pos" if q is not None else ""}]',
                     x.numel() * 2 + y.numel() * 2 * (1 + (q is not None)) + (HW * Cout * 4 if q is not None else 0), lambda: hip.check(
            lib.ge_conv1x1_bn_act_fwd(_raw_ptr(x, 'x'), hip.ptr(wc), hip.ptr(coef), hip.ptr(pos_rows, _f32), _raw_ptr(y, 'y'), hip.ptr(q), B, HW, Cin, Cout,
                                      slope, hip.stream()), 'ge_conv1x1_bn_act_fwd'))
        ctx.save_for_backward(x, wc, g32, stats, gram, y)
        ctx.meta = (slope, weight.dtype, gamma.dtype, beta.dtype)
        ctx.weight_ref = weight if isinstance(weight, torch.nn.Parameter) else None
        if q is None:
            return y
        return y, q

    @staticmethod
    def backward(ctx, dy, dq=None):
        x, wc, g32, stats, gram, y = ctx.saved_tensors
        slope, w_dtype, g_dtype, b_dtype = ctx.meta
        lib = hip.lib()
        B, Cin, H, W = x.shape
        Cout = wc.shape[0]
        rows = B * H * W
        dev = x.device
        ld_y = 0
        if dy is not None:
            dy = dy.to(x.dtype)
            # a channel slice of a wider channels-last map (the concat's gradient) is read where it lies: rows of Cout channels, row stride ld
            ld_y = dy.stride(3)
            if not (dy.stride(1) == 1 and ld_y >= Cout and ld_y % 8 == 0 and dy.stride(2) == W * ld_y and dy.stride(0) == H * W * ld_y and dy.data_ptr() % 16 == 0):
                dy, ld_y = _cl(dy), Cout
        if dq is not None:
            dq = _c(dq.to(x.dtype))
        if dy is None and dq is None:
            return (None,) * 10
        ws = torch.empty(int(lib.ge_conv1x1_bn_workspace(Cin, Cout)), device=dev, dtype=torch.uint8)
        g = torch.empty(rows, Cout, device=dev, dtype=x.dtype)
        m1 = torch.empty(Cout, device=dev, dtype=_f32)
        PROFILER.run(f'conv1x1_bn_bwd_mask[{B}x{Cout} {H}x{W}]', g.numel() * 2 * (2 + (dy is not None) + (dq is not None)), lambda: hip.check(
            lib.ge_conv1x1_bn_bwd_mask(hip.ptr(dq), Cout, None if dy is None else _raw_ptr(dy, 'dy'), ld_y, _raw_ptr(y, 'y'), hip.ptr(g), hip.ptr(m1),
                                       hip.ptr(ws), rows, Cout, slope, hip.stream()), 'ge_conv1x1_bn_bwd_mask'))
        GT = torch.zeros(Cout, Cin, device=dev, dtype=_f32)
        PROFILER.run(f'conv1x1_wgrad[{B}x{Cin}->{Cout} {H}x{W}]', (x.numel() + g.numel()) * 2 + GT.numel() * 4, lambda: hip.check(
            lib.ge_conv1x1_nhwc_wgrad(_raw_ptr(x, 'x'), hip.ptr(g), hip.ptr(GT), rows, Cin, Cout, hip.GE_BF16, hip.stream()), 'ge_conv1x1_nhwc_wgrad'))
        small = torch.empty(2 * Cout + Cout * Cin + 2 * Cout, device=dev, dtype=_f32)
        dgamma, dbeta, dW, scratch = small[:Cout], small[Cout:2 * Cout], small[2 * Cout:2 * Cout + Cout * Cin].view(Cout, Cin), small[2 * Cout + Cout * Cin:]
        Wd = torch.empty(Cin, Cout + Cin, device=dev, dtype=x.dtype)
        c0 = torch.empty(Cin, device=dev, dtype=_f32)
        hip.check(lib.ge_conv1x1_bn_bwd_finalize(hip.ptr(GT), hip.ptr(m1), hip.ptr(gram), hip.ptr(wc), hip.ptr(g32), hip.ptr(stats[0]), hip.ptr(stats[1]),
                                                 rows, Cin, Cout, hip.ptr(dgamma), hip.ptr(dbeta), hip.ptr(dW), hip.ptr(Wd), hip.ptr(c0),
                                                 hip.ptr(scratch), hip.stream()), 'ge_conv1x1_bn_bwd_finalize')
        dx = None
        if ctx.needs_input_grad[0]:
            # the convolution's data gradient with the BatchNorm scale folded into the weights + the rank-64 corrections of the BatchNorm backward
            dx = torch.empty_like(x)
            PROFILER.run(f'conv1x1_bn_dgrad[{B}x{Cout}->{Cin} {H}x{W}]', (g.numel() + 2 * x.numel()) * 2, lambda: hip.check(
                lib.ge_conv1x1_bn_dgrad(hip.ptr(g), _raw_ptr(x, 'x'), hip.ptr(Wd), hip.ptr(c0), _raw_ptr(dx, 'dx'), rows, Cin, Cout, hip.stream()),
                'ge_conv1x1_bn_dgrad'))
        dw = dW.view(Cout, Cin, 1, 1)
        if ctx.weight_ref is not None:
            from .mmrt.optim import grad_into_arena
            dw = grad_into_arena(ctx.weight_ref, dw, w_dtype)
        else:
            dw = dw.to(w_dtype)
        return dx, dw, dgamma.to(g_dtype), dbeta.to(b_dtype), None, None, None, None, None, None


def conv1x1_bn_act_pos_ok(block, x):
    """The fused 1x1-conv + BatchNorm + ReLU (+ position add) kernels apply to ConvModule ``block`` on input ``x``: training-mode BatchNorm2d,
    bf16 autocast, a 64-channel channels-last bf16 map, output width a multiple of 128, no conv bias (csrc/conv1x1_bn.hip).  GE_DISABLE=conv1x1_bn
    keeps the two-pass kernels (which the fp32 parity mode always uses)."""
    conv, bn = block.conv, block.norm
    return ('conv1x1_bn' not in DISABLED and x.is_cuda and x.dtype == torch.bfloat16 and type(conv) is torch.nn.Conv2d and conv.kernel_size == (1, 1)
            and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1 and conv.bias is None and conv.in_channels == 64
            and conv.out_channels % 128 == 0 and conv.out_channels <= 1024 and conv.weight.dtype == _f32
            and torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16
            and block._fused_bn_slope() is not None and _is_cl(x) and x.data_ptr() % 16 == 0)


def conv1x1_bn_act_pos(block, x, pos=None):
    """-> (act(bn(conv(x))) as a channels-last map, tokens of it + ``pos`` as (B, H*W, C) or None) in one pass over the output; ``pos``: the cached
    (1, C, H, W) fp32 positional map.  Caller checks ``conv1x1_bn_act_pos_ok`` first."""
    bn = block.norm
    slope = float(block._fused_bn_slope())
    pos_rows = _pos_rows(pos) if pos is not None else None
    out = _Conv1x1BNActPos.apply(x, block.conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, pos_rows, float(bn.eps), float(bn.momentum), slope)
    bn.num_batches_tracked.add_(1)
    return out if pos is not None else (out, None)


# ------------------------------------------------------------------------- bias + activation
class _BiasAct(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, bias, slope):
        ctx.cl = _cl_ok(x)
        assert x.dim() == 4 and (x.is_contiguous() or ctx.cl), 'bias_act works in place on a dense NCHW / channels-last conv output'
        N, C, H, W = x.shape
        b = _c(bias.detach().to(_f32))
        if ctx.cl:
            PROFILER.run(f'bias_act_nhwc_fwd[{N}x{C}x{H}x{W} {_tag(x)}]', 2 * x.numel() * _es(x), lambda: hip.check(
                hip.lib().ge_bias_act_nhwc_fwd(_raw_ptr(x, 'x'), hip.ptr(b), N * H * W, C, slope, hip.dtype_code(x), hip.stream()),
                'ge_bias_act_nhwc_fwd'))
            ctx.mark_dirty(x)
            ctx.save_for_backward(x)
            ctx.slope = slope
            return x
        PROFILER.run(f'bias_act_fwd[{N}x{C}x{H}x{W} {_tag(x)}]', 2 * x.numel() * _es(x), lambda: hip.check(
            hip.lib().ge_bias_act_fwd(hip.ptr(x, name='x'), hip.ptr(b), N, C, H * W, slope, hip.dtype_code(x), hip.stream()),
            'ge_bias_act_fwd'))
        ctx.mark_dirty(x)
        ctx.save_for_backward(x)
        ctx.slope = slope
        return x

    @staticmethod
    def backward(ctx, dy):
        y, = ctx.saved_tensors
        N, C, H, W = y.shape
        dx = torch.empty_like(y)
        if ctx.cl:
            dy = _cl(dy.to(y.dtype))
            db = torch.empty(C, device=y.device, dtype=_f32)
            ws = torch.empty(int(hip.lib().ge_nhwc_workspace(C, 1)), device=y.device, dtype=torch.uint8)
            PROFILER.run(f'bias_act_nhwc_bwd[{N}x{C}x{H}x{W} {_tag(y)}]', 3 * y.numel() * _es(y), lambda: hip.check(
                hip.lib().ge_bias_act_nhwc_bwd(_raw_ptr(dy, 'dy'), _raw_ptr(y, 'y'), _raw_ptr(dx, 'dx'), hip.ptr(db), hip.ptr(ws), N * H * W, C,
                                               ctx.slope, hip.dtype_code(y), hip.stream()), 'ge_bias_act_nhwc_bwd'))
            return dx, db, None
        dy = _c(dy.to(y.dtype))
        db = torch.zeros(C, device=y.device, dtype=_f32)
        PROFILER.run(f'bias_act_bwd[{N}x{C}x{H}x{W} {_tag(y)}]', 3 * y.numel() * _es(y), lambda: hip.check(
            hip.lib().ge_bias_act_bwd(hip.ptr(dy), hip.ptr(y), hip.ptr(dx), hip.ptr(db), N, C, H * W, ctx.slope,
                                      hip.dtype_code(y), hip.stream()), 'ge_bias_act_bwd'))
        return dx, db, None


def bias_act_(conv_out, bias, slope=1.0):
    """In place on a fresh (bias-free) convolution output: leaky_relu(conv_out + bias[c], slope); slope 1 = no activation."""
    return _BiasAct.apply(conv_out, bias, float(slope))


# ------------------------------------------------------------------------ ground embedding
def _plane_view(img, channel):
    """(base tensor for the pointer, batch stride in elements) of img[:, channel] (B,5,H,W contiguous)."""
    if not img.is_cuda:
        raise RuntimeError('gedepth_amd ops run on MI355X only; got a CPU image tensor')
    assert img.dim() == 4 and img.is_contiguous() and img.dtype == _f32
    return img[:, channel], img.stride(0)


class _GroundEmbedAdaptive(torch.autograd.Function):

    @staticmethod
    def forward(ctx, logits_lr, y_lr, img, height, depth_scale):
        logits_lr = _c(logits_lr.to(_f32))
        y_lr = _c(y_lr.to(_f32))
        B, _, h, w = logits_lr.shape
        H, W = img.shape[2], img.shape[3]
        pe, bs = _plane_view(img, 4)
        dev = img.device
        pe_mask = torch.empty(B, 1, H, W, device=dev, dtype=_f32)
        logits_hr = torch.empty(B, 11, H, W, device=dev, dtype=_f32)
        y_hr = torch.empty(B, 1, H, W, device=dev, dtype=_f32)
        valid = torch.empty(B, H, W, device=dev, dtype=torch.uint8)
        height = None if height is None else _c(height.to(_f32))
        nbytes = B * H * W * (4 + 4 + 44 + 4 + 1) + B * h * w * 12 * 4
        PROFILER.run(f'ground_embed_fwd[{B}x{H}x{W}]', nbytes, lambda: hip.check(hip.lib().ge_ground_embed_fwd(
            hip.ptr(logits_lr), hip.ptr(y_lr), pe.data_ptr(), bs, hip.ptr(height), depth_scale,
            hip.ptr(pe_mask), hip.ptr(logits_hr), hip.ptr(y_hr), hip.ptr(valid), B, h, w, H, W, hip.stream()),
            'ge_ground_embed_fwd'))
        ctx.save_for_backward(logits_lr, y_lr, img, height)
        ctx.depth_scale = depth_scale
        ctx.mark_non_differentiable(valid)
        return pe_mask, logits_hr, y_hr, valid

    @staticmethod
    def backward(ctx, d_pe_mask, d_logits_hr, d_y_hr, _d_valid):
        logits_lr, y_lr, img, height = ctx.saved_tensors
        B, _, h, w = logits_lr.shape
        H, W = img.shape[2], img.shape[3]
        pe, bs = _plane_view(img, 4)
        dev = img.device
        d_pe_mask = _c(d_pe_mask.to(_f32))
        d_logits_hr = None if d_logits_hr is None else _c(d_logits_hr.to(_f32))
        d_y_hr = None if d_y_hr is None else _c(d_y_hr.to(_f32))
        d_logits_lr = torch.empty_like(logits_lr)
        d_y_lr = torch.empty_like(y_lr)
        scratch = torch.empty(B, 12, H, W, device=dev, dtype=_f32)
        nbytes = B * H * W * (4 + 4 + 44 + 4) + 2 * B * h * w * 12 * 4
        PROFILER.run(f'ground_embed_bwd[{B}x{H}x{W}]', nbytes, lambda: hip.check(hip.lib().ge_ground_embed_bwd(
            hip.ptr(logits_lr), hip.ptr(y_lr), pe.data_ptr(), bs, hip.ptr(height), ctx.depth_scale,
            hip.ptr(d_pe_mask), hip.ptr(d_logits_hr), hip.ptr(d_y_hr), hip.ptr(d_logits_lr), hip.ptr(d_y_lr),
            hip.ptr(scratch), B, h, w, H, W, hip.stream()), 'ge_ground_embed_bwd'))
        return d_logits_lr, d_y_lr, None, None, None


def ground_embed_adaptive(logits_lr, y_lr, img, height=None, depth_scale=200.0):
    """-> pe_mask (B,1,H,W), logits_hr (B,11,H,W), y_hr (B,1,H,W), valid_mask u8 (B,H,W)."""
    return _GroundEmbedAdaptive.apply(logits_lr, y_lr, img, height, float(depth_scale))


class _GroundEmbedVanilla(torch.autograd.Function):

    @staticmethod
    def forward(ctx, y_lr, img, gain):
        y_lr = _c(y_lr.to(_f32))
        B, _, h, w = y_lr.shape
        H, W = img.shape[2], img.shape[3]
        pe, bs = _plane_view(img, 3)
        pe_mask = torch.empty(B, 1, H, W, device=img.device, dtype=_f32)
        y_hr = torch.empty(B, 1, H, W, device=img.device, dtype=_f32)
        hip.check(hip.lib().ge_ground_vanilla_fwd(hip.ptr(y_lr), pe.data_ptr(), bs, gain, hip.ptr(pe_mask), hip.ptr(y_hr),
                                                  B, h, w, H, W, hip.stream()), 'ge_ground_vanilla_fwd')
        ctx.save_for_backward(img)
        ctx.geom = (B, h, w, H, W, gain)
        return pe_mask, y_hr

    @staticmethod
    def backward(ctx, d_pe_mask, d_y_hr):
        img, = ctx.saved_tensors
        B, h, w, H, W, gain = ctx.geom
        pe, bs = _plane_view(img, 3)
        d_pe_mask = _c(d_pe_mask.to(_f32))
        d_y_hr = None if d_y_hr is None else _c(d_y_hr.to(_f32))
        d_y_lr = torch.empty(B, 1, h, w, device=img.device, dtype=_f32)
        scratch = torch.empty(B, 1, H, W, device=img.device, dtype=_f32)
        hip.check(hip.lib().ge_ground_vanilla_bwd(pe.data_ptr(), bs, gain, hip.ptr(d_pe_mask), hip.ptr(d_y_hr), hip.ptr(d_y_lr),
                                                  hip.ptr(scratch), B, h, w, H, W, hip.stream()), 'ge_ground_vanilla_bwd')
        return d_y_lr, None, None


def ground_embed_vanilla(y_lr, img, gain=200.0):
    """-> pe_mask = img[:,3:4] * up(y) * 200, y_hr."""
    return _GroundEmbedVanilla.apply(y_lr, img, float(gain))


# ----------------------------------------------------------------------------- depth fusion
class _DepthFuse(torch.autograd.Function):

    @staticmethod
    def forward(ctx, c, pe_mask, y_hr, min_depth):
        c = _c(c.to(_f32))
        pe_mask = _c(pe_mask.to(_f32))
        y_hr = _c(y_hr.to(_f32))
        B, _, h, w = c.shape
        H, W = pe_mask.shape[2], pe_mask.shape[3]
        out = torch.empty_like(c)
        y_ds = torch.empty_like(c)
        hip.check(hip.lib().ge_depth_fuse_fwd(hip.ptr(c), hip.ptr(pe_mask), hip.ptr(y_hr), min_depth, hip.ptr(out), hip.ptr(y_ds),
                                              B, h, w, H, W, hip.stream()), 'ge_depth_fuse_fwd')
        ctx.save_for_backward(c, y_ds)
        ctx.geom = (B, h, w, H, W)
        ctx.mark_non_differentiable(y_ds)
        return out, y_ds

    @staticmethod
    def backward(ctx, d_out, _d_yds):
        c, y_ds = ctx.saved_tensors
        B, h, w, H, W = ctx.geom
        d_out = _c(d_out.to(_f32))
        d_c = torch.empty_like(c)
        d_pe = torch.empty(B, 1, H, W, device=c.device, dtype=_f32)
        d_y = torch.empty(B, 1, H, W, device=c.device, dtype=_f32)
        scratch = torch.empty(B, 2, h, w, device=c.device, dtype=_f32)
        hip.check(hip.lib().ge_depth_fuse_bwd(hip.ptr(c), hip.ptr(y_ds), hip.ptr(d_out), hip.ptr(d_c), hip.ptr(d_pe), hip.ptr(d_y),
                                              hip.ptr(scratch), B, h, w, H, W, hip.stream()), 'ge_depth_fuse_bwd')
        return d_c, d_pe, d_y, None


def depth_fuse(conv_out, pe_mask, y_hr, min_depth):
    """relu(c)*(1-dn(y)) + dn(pe) + min_depth -> (out, y_ds)."""
    return _DepthFuse.apply(conv_out, pe_mask, y_hr, float(min_depth))


# ------------------------------------------------------------------------------------ SiLog
class _SiLog(torch.autograd.Function):

    @staticmethod
    def forward(ctx, pred, gt, eps, loss_weight):
        pred = _c(pred.to(_f32))
        gt = _c(gt.to(_f32))
        stats = torch.zeros(3, device=pred.device, dtype=torch.float64)
        hip.check(hip.lib().ge_silog_stats(hip.ptr(pred), hip.ptr(gt), eps, hip.ptr(stats), pred.numel(), hip.stream()),
                  'ge_silog_stats')
        n, s1, s2 = stats[0], stats[1], stats[2]
        mean = s1 / n
        var = (s2 - n * mean * mean) / (n - 1)          # torch.var: unbiased
        D = torch.sqrt(var + 0.15 * mean * mean)
        ctx.save_for_backward(pred, gt, n, mean, D)
        ctx.eps, ctx.w = eps, loss_weight
        return (loss_weight * D).to(_f32)

    @staticmethod
    def backward(ctx, g):
        pred, gt, n, mean, D = ctx.saved_tensors
        k = (g.double() * ctx.w) / (2.0 * D)
        coef_a = (k * 2.0 / (n - 1)).to(_f32).reshape(1).contiguous()
        coef_b = (k * (-2.0 * mean / (n - 1) + 0.3 * mean / n)).to(_f32).reshape(1).contiguous()
        d_pred = torch.empty_like(pred)
        hip.check(hip.lib().ge_silog_bwd(hip.ptr(pred), hip.ptr(gt), ctx.eps, hip.ptr(coef_a), hip.ptr(coef_b), hip.ptr(d_pred),
                                         pred.numel(), hip.stream()), 'ge_silog_bwd')
        return d_pred, None, None, None


def silog_loss(pred, gt, eps=1e-3, loss_weight=1.0):
    """SigLoss with valid_mask = gt > 0, without the boolean gather (no host sync)."""
    return _SiLog.apply(pred, gt, float(eps), float(loss_weight))


# --------------------------------------------------------------------- offline ground maps
def ground_plane(rinv_row2, num, H, W, device='cuda', want_f64=True):
    """pe(u,v) = num / (r20 u + r21 v + r22); returns (pe_f64 or None, pe_f32)."""
    arr = (ctypes.c_double * 3)(*[float(v) for v in rinv_row2])
    pe64 = torch.empty(H, W, device=device, dtype=torch.float64) if want_f64 else None
    pe32 = torch.empty(H, W, device=device, dtype=_f32)
    hip.check(hip.lib().ge_ground_plane(ctypes.cast(arr, ctypes.c_void_p), float(num), hip.ptr(pe64), hip.ptr(pe32), H, W,
                                        hip.stream()), 'ge_ground_plane')
    return pe64, pe32


def slope_class(gt_f64, pe_f32, cam_height=1.65, mode='round'):
    gt_f64 = _c(gt_f64.to(torch.float64))
    pe_f32 = _c(pe_f32.to(_f32))
    H, W = gt_f64.shape
    cls = torch.empty(H, W, device=gt_f64.device, dtype=torch.int16)
    hip.check(hip.lib().ge_slope_class(hip.ptr(gt_f64), hip.ptr(pe_f32), float(cam_height), 0 if mode == 'round' else 1,
                                       hip.ptr(cls), H, W, hip.stream()), 'ge_slope_class')
    return cls


def slope_class_ddad(gt_f32, pe_f64, cam_height):
    """DDAD slope classes with the reference script's dtypes (gt float32, pe float64, truncation); int16 (H,W)."""
    gt_f32 = _c(gt_f32.to(_f32))
    pe_f64 = _c(pe_f64.to(torch.float64))
    H, W = gt_f32.shape
    cls = torch.empty(H, W, device=gt_f32.device, dtype=torch.int16)
    hip.check(hip.lib().ge_slope_class_ddad(hip.ptr(gt_f32), hip.ptr(pe_f64), float(cam_height), hip.ptr(cls), H, W,
                                            hip.stream()), 'ge_slope_class_ddad')
    return cls


def pe_channels(raw, depth_scale=200.0):
    raw = _c(raw.to(_f32))
    norm = torch.empty_like(raw)
    hip.check(hip.lib().ge_pe_channels(hip.ptr(raw), hip.ptr(norm), float(depth_scale), raw.numel(), hip.stream()), 'ge_pe_channels')
    return norm
