/*
 * gedepth_hip.h — C ABI of libgedepth_hip.so: the hand-written gfx950 (MI355X / CDNA4) kernels
 * of the GEDepth training hot path.
 *
 * The reference (qcraftai/gedepth) has no FFI: these ops replace chains of PyTorch/ATen kernels
 * and one mmcv CUDA extension.  Each entry point cites the reference code it replaces
 * (paths relative to the reference root).
 *
 * Conventions
 *   - extern "C"; every function returns 0 on success, otherwise a hipError_t value
 *     (GE_ERR_* below for argument errors).  Nothing throws, allocates or synchronises.
 *   - all pointers are DEVICE pointers owned by the caller, contiguous in the documented layout;
 *   - `stream` is a hipStream_t (pass PyTorch's current stream);
 *   - `dtype` selects the storage type of the activation tensors: GE_F32 or GE_BF16.
 *     Accumulation is always fp32.  Bias tables, sampling locations, attention weights and the
 *     whole ground-embedding path are fp32 regardless;
 *   - thread-safe, and re-entrant except for three process-wide measurement / selection knobs that live in the library (mutex-guarded):
 *     the kernel-selection mode of the deformable attention (ge_msda_mode: an A/B switch for tests and timing; kernels read it at
 *     launch), the opt-in per-kernel timing of the composite MSDA backward (ge_msda_bwd_timing*: off by default, records nothing and
 *     never synchronises when off), and cached device properties (CU count, occupancy of the persistent MSDA kernels); ge_conv3x3_nhwc_fwd
 *     reads the environment variable GE_CONV3X3 (A/B switch between its two kernels, same results).  No entry point keeps state that
 *     changes its RESULTS between calls.
 */
#ifndef GEDEPTH_HIP_H
#define GEDEPTH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { GE_F32 = 0, GE_BF16 = 1 };
enum { GE_OK = 0, GE_ERR_BAD_ARG = 10001, GE_ERR_UNSUPPORTED = 10002 };

/* Library / device identification: returns the ABI version (3: round 3 added the raw-projection deformable-attention entry points,
 * the bias+GELU epilogue, the decoder glue passes and the DDAD front end of the device pipeline; 4: round 4 added the MFMA
 * decomposition of the deformable attention (ge_msda_*_mm, ge_msda_bwd_value_raw), the token GEMM ge_gemm_nt and ge_conv1x1_nhwc_wgrad;
 * 5: round 5 — ge_msda_bwd_lw_mm takes a workspace, ge_msda_bwd_value_mm / ge_msda_bwd_mm_workspace added; 6: the fused
 * 1x1-convolution + BatchNorm + ReLU + position-add entry points ge_conv1x1_bn_*; 7: round 6 — the value-stationary d_value kernel
 * ge_msda_bwd_value_vs / ge_msda_bwd_vs_workspace / ge_msda_bwd_vs_stats_offset and the query-range variants ge_msda_fwd_mm_part /
 * ge_msda_bwd_lw_mm_part). */
int ge_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * Swin shifted-window attention core.
 * Replaces ShiftWindowMSA.forward + WindowMSA.forward between the qkv and proj Linears
 * (depth/models/backbones/depthformer_swin.py:285-360 and :193-221): zero-pad to a multiple of 7
 * (after norm1, before qkv => pad tokens carry q=k=v=qkv.bias), roll(-shift), window partition,
 * q*scale, QK^T, + relative-position bias (index (i_a-i_b+6)*13+(j_a-j_b+6), :166-172), + shift
 * mask (-100 between regions, :305-326), softmax, AV, window reverse, roll(+shift), crop.
 *
 *   qkv        (B, H*W, 3*C)  dtype; last dim ordered [3][nH][32]  (reshape of :193-194)
 *   qkv_bias   (3*C)          f32   value of q/k/v for pad tokens
 *   bias_table (169, nH)      f32   relative_position_bias_table
 *   out        (B, H*W, C)    dtype; channel = head*32 + d  (the transpose(1,2).reshape of :221)
 * head_dim is fixed at 32 (true for every Swin stage of the reference), window 7, shift in {0,3}.
 * `variant`: 0 = auto, 1 = exact-fp32 VALU kernel, 2 = bf16 MFMA kernel (dtype must be GE_BF16), 3 = fp8 (OCP e4m3) MFMA
 *   forward on bf16 storage: QK^T and PV on v_mfma_f32_32x32x16_fp8_fp8 with per-(window, head) operand scales
 *   448 / amax (BASELINE.json configs[4]); its backward is variant 2's.
 */
int ge_window_attn_fwd(const void* qkv, const float* qkv_bias, const float* bias_table, void* out,
                       int B, int H, int W, int nH, int shift, float scale, int dtype, int variant,
                       void* stream);

/* Backward.  d_qkv (B,H*W,3C) dtype is fully written.  d_bias_table (169,nH) f32 and
 * d_qkv_bias (3C) f32 (pad-token contribution, to be ADDED to the Linear's own bias grad) are
 * fully written as well (deterministic two-stage reduction through `workspace`).
 * workspace: at least ge_window_attn_bwd_workspace(...) bytes. */
size_t ge_window_attn_bwd_workspace(int B, int H, int W, int nH);
int ge_window_attn_bwd(const void* qkv, const float* qkv_bias, const float* bias_table, const void* d_out,
                       void* d_qkv, float* d_qkv_bias, float* d_bias_table, void* workspace,
                       int B, int H, int W, int nH, int shift, float scale, int dtype, int variant,
                       void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-scale deformable attention sampling core.
 * Replaces mmcv.ops.MultiScaleDeformableAttention's ms_deform_attn CUDA op as called from
 * depth/models/necks/hahi.py:279-289,316-325 (mmcv-full 1.3.13, not vendored): bilinear taps with
 * zero padding, align_corners=False, weighted sum over levels x points.
 *
 *   value          (B, Nv, nH, 64)        dtype   (nH*64 == embed dims; channels per head fixed at 64)
 *   spatial_hw     host int[2*L]          (H_0,W_0, H_1,W_1, ...)   L <= 8
 *   loc            (B, Nq, nH, L, P, 2)   f32     normalised (x,y) in [0,1]
 *   attw           (B, Nq, nH, L, P)      f32
 *   out            (B, Nq, nH*64)         dtype
 *   query_hw       host int[2*n_qseg] or NULL: the query set as n_qseg (H, W) maps in raster order (sum H*W == Nq) — the
 *                  value levels themselves for the self-attention (hahi.py:279-289), the single 176x560 map for the
 *                  cross-attention (:303-325).  With it (and P == 8) the kernels tile the queries 2-D and serve the taps
 *                  from LDS-staged value windows (csrc/msda_win.hip); NULL selects the streaming kernels.  Same results.
 */
int ge_msda_fwd(const void* value, const int* spatial_hw, const int* query_hw, int n_qseg, const float* loc,
                const float* attw, void* out, int B, int Nv, int Nq, int nH, int L, int P, int dtype, void* stream);
/* Kernel selection knob (A/B timing, tests): bit 0 = window forward, bit 1 = window d_loc/d_attw, bit 2 = owner-lane tap
 * arithmetic inside the window kernels, bit 3 = head-major work order of the streaming kernels; mode < 0 only queries.
 * Returns the previous mode (default 13: window forward, streaming head-major d_loc/d_attw). */
int ge_msda_mode(int mode);

/* Backward.  d_value (B,Nv,nH,64) is ALWAYS f32 and must be zero-filled by the caller; d_loc / d_attw are fully
 * written.  `workspace` (>= ge_msda_bwd_workspace(...) bytes, caller-owned scratch) enables the binned scatter
 * (one integer atomic per tap, tiles accumulated in registers); with workspace == NULL the scatter falls back to
 * fp32 atomic bursts straight into d_value. */
size_t ge_msda_bwd_workspace(const int* spatial_hw, int B, int Nv, int Nq, int nH, int L, int P);
/* Introspection (no device work): out4 = {binned path available, query ranges per (image, head) — the counting sort's
 * work units are (image, head, query range), one 1024-thread workgroup each —, value tiles, bins}. */
int ge_msda_bwd_plan(const int* spatial_hw, int B, int Nv, int Nq, int nH, int L, int P, int* out4);
int ge_msda_bwd(const void* value, const int* spatial_hw, const int* query_hw, int n_qseg,
                const float* loc, const float* attw,
                const void* d_out, float* d_value, float* d_loc, float* d_attw,
                void* workspace, size_t workspace_bytes,
                int B, int Nv, int Nq, int nH, int L, int P, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Bilinear resize (NCHW), both align_corners conventions of F.interpolate, used for every
 * resize on the path (depth/ops/wrappers.py:7-26; necks/pemask_neck.py:54-61;
 * decode_heads/densedepth_head.py:26; decode_head.py:492-503,515-520).
 * Backward is a deterministic gather (no atomics): d_in is fully written.
 *   in (N, C, Hi, Wi) -> out (N, C, Ho, Wo), dtype storage.
 */
int ge_bilinear_fwd(const void* in, void* out, int N, int C, int Hi, int Wi, int Ho, int Wo,
                    int align_corners, int dtype, void* stream);
int ge_bilinear_bwd(const void* d_out, void* d_in, int N, int C, int Hi, int Wi, int Ho, int Wo,
                    int align_corners, int dtype, void* stream);

/* Measurement aid (bench.py): per-kernel durations of the composite ge_msda_bwd.  ge_msda_bwd_timing(1) resets the
 * counters and makes every following workspace-backed ge_msda_bwd record HIP events between its kernels on the launch
 * stream; ge_msda_bwd_timing_read(stage, ...) waits for the recorded events and returns the accumulated milliseconds,
 * the launch count and the kernel name of stage 0..4 (lw, count, scan, fill, drain).  Off by default: the entry point
 * then records nothing and never synchronises. */
int ge_msda_bwd_timing(int enable);
int ge_msda_bwd_timing_read(int stage, double* total_ms, long* launches, char* name, int name_cap);

/* ---------------------------------------------------------------------------------------------
 * HAHI neck glue around the deformable attention (necks/hahi.py:303-346 and the mmcv 1.3.13
 * MultiScaleDeformableAttention.forward it calls).
 *
 * ge_msda_prep_fwd: raw outputs of the `sampling_offsets` / `attention_weights` linears -> the fp32 loc / attw tensors
 *   of ge_msda_fwd in one pass:  loc = ref + off_raw / (W_l, H_l),  attw = softmax_{l,p}(logit_raw).
 *   off_raw  : rows of B*Nq, row stride off_ld elements, nH*L*P*2 used columns ordered (h,l,p,xy);
 *   logit_raw: row stride logit_ld, nH*L*P used columns (both may be column ranges of one fused GEMM output);
 *   ref      : f32 reference points in [0,1], element strides (ref_sb, ref_sq, ref_sl) over (b, q, l), xy adjacent
 *              (stride 0 = broadcast).  L must be 4, P 4 or 8; row starts 16-byte aligned.
 * ge_msda_prep_bwd: d_off_raw = d_loc / (W_l, H_l), d_logit_raw = softmax backward; optional d_ref (B*Nq, L, 2) f32
 *   = sum over heads and points of d_loc (NULL to skip; needs nH in {1,2,4,8,16}).
 *
 * ge_tokens_from_map: tok[b,n,c] = (map[b,c,n] + pos[c,n]) * drop(b,n,c)   ("flatten(2).transpose(1,2)" + query_pos add)
 * ge_map_from_tokens: map[b,c,n] = tok[b,n,c] * drop(b,n,c) + res[b,c,n]   (dropout + identity + "permute(0,2,1).reshape",
 *   written with batch stride map_bs so it can land inside the torch.cat buffer of hahi.py:333/346).
 *   Batch strides are in elements; pos (f32, C x N) and res may be NULL; drop() is 1 when p_drop == 0, else a
 *   counter-based Bernoulli(1 - p_drop) / (1 - p_drop) of (seed, (b*N + n)*C + c): the two entry points with equal
 *   (seed, p_drop) apply the same mask, which is how each serves as the other's backward.
 */
int ge_msda_prep_fwd(const void* off_raw, long off_ld, const void* logit_raw, long logit_ld, const float* ref,
                     long ref_sb, long ref_sq, long ref_sl, const int* spatial_hw, float* loc, float* attw, int B,
                     int Nq, int nH, int L, int P, int dtype, void* stream);
int ge_msda_prep_bwd(const float* d_loc, const float* d_attw, const float* attw, const int* spatial_hw, void* d_off_raw,
                     long off_ld, void* d_logit_raw, long logit_ld, float* d_ref, int B, int Nq, int nH, int L, int P,
                     int dtype, void* stream);

/* Fused prepare + sampling ("raw" entry points, round 3): the deformable attention taken from the RAW outputs of the
 * `sampling_offsets` / `attention_weights` linears — mmcv MultiScaleDeformableAttention.forward's view -> softmax over L*P ->
 * `reference_points + offsets / (W_l, H_l)` -> ms_deform_attn in ONE kernel (replaces ge_msda_prep_fwd + ge_msda_fwd; reference
 * call sites depth/models/necks/hahi.py:279-289,316-325), and its backward straight to the gradient of those raw outputs
 * (replaces ge_msda_bwd + ge_msda_prep_bwd).
 *   off_raw   (B*Nq rows of off_ld elements; columns (head, level, point, xy)), logit_raw (rows of logit_ld; columns (head,
 *             level, point)), storage type `dtype`; ref (B, Nq, L, 2) f32 with ELEMENT strides ref_sb / ref_sq / ref_sl (0 = broadcast)
 *   loc, attw (B,Nq,nH,L,P[,2]) f32: WRITTEN by the forward, read by the backward (what ge_msda_prep_fwd would have produced)
 *   d_off_raw / d_logit_raw: same layout and type as off_raw / logit_raw, fully written; d_ref (B*Nq, L, 2) f32 or NULL;
 *   d_value NULL in ge_msda_bwd_raw = skip the d_value scatter (the caller takes it from ge_msda_bwd_value_raw: 8-byte records).
 * ge_msda_raw_supported() = 1 when the fused kernels apply (L == 4, P == 8, a query grid, default kernel-selection mode);
 * ge_msda_fwd_raw otherwise runs the two-pass composition itself, ge_msda_bwd_raw returns GE_ERR_UNSUPPORTED. */
int ge_msda_raw_supported(const int* spatial_hw, const int* query_hw, int n_qseg, int B, int Nv, int Nq, int nH, int L, int P);
int ge_msda_fwd_raw(const void* value, const int* spatial_hw, const int* query_hw, int n_qseg, const void* off_raw, long off_ld,
                    const void* logit_raw, long logit_ld, const float* ref, long ref_sb, long ref_sq, long ref_sl, float* loc,
                    float* attw, void* out, int B, int Nv, int Nq, int nH, int L, int P, int dtype, void* stream);
int ge_msda_bwd_raw(const void* value, const int* spatial_hw, const int* query_hw, int n_qseg, const float* loc, const float* attw,
                    const void* d_out, float* d_value, void* d_off_raw, long off_ld, void* d_logit_raw, long logit_ld, float* d_ref,
                    void* workspace, size_t workspace_bytes, int B, int Nv, int Nq, int nH, int L, int P, int dtype, void* stream);
/* Deformable attention as MFMA contractions on wave-private LDS windows (round 4; csrc/msda_mm.hip): the same op as
 * ge_msda_fwd_raw for bf16 storage, L == 4, P == 8, with the queries processed in the caller's ORDER — `order` (device, Nq ints, a
 * permutation of 0..Nq-1, the same for every image; NULL = identity) lists the queries so that 32 consecutive entries sample
 * neighbouring value rows (2-D tiles of the token maps for the self-attention, the queries sorted by the cell of their reference
 * point for the cross-attention, depth/models/necks/hahi.py:294-302).  Results do not depend on the order, only the speed does.
 * loc / attw: as ge_msda_fwd_raw, or both NULL (not written). */
int ge_msda_mm_supported(const int* spatial_hw, int B, int Nv, int Nq, int nH, int L, int P, int dtype);
int ge_msda_fwd_mm(const void* value, const int* spatial_hw, const void* off_raw, long off_ld, const void* logit_raw, long logit_ld,
                   const float* ref, long ref_sb, long ref_sq, long ref_sl, const int* order, float* loc, float* attw, void* out,
                   int B, int Nv, int Nq, int nH, int L, int P, int dtype, void* stream);
/* Backward of ge_msda_fwd_mm in two halves: ge_msda_bwd_lw_mm emits d_off_raw / d_logit_raw (layout and type of off_raw /
 * logit_raw, fully written; mmcv's view / normaliser / softmax backward folded in) on the same MFMA decomposition; ge_msda_bwd_value
 * runs the binned d_value scatter of ge_msda_bwd alone (loc / attw as written by the forward, workspace of ge_msda_bwd_workspace,
 * d_value f32 zero-filled by the caller); ge_msda_dref rebuilds d_ref (rows, L, 2) f32 from the emitted d_off_raw. */
int ge_msda_bwd_lw_mm(const void* value, const int* spatial_hw, const void* off_raw, long off_ld, const void* logit_raw, long logit_ld,
                      const float* ref, long ref_sb, long ref_sq, long ref_sl, const int* order, const void* d_out, void* d_off_raw,
                      long d_off_ld, void* d_logit_raw, long d_logit_ld, void* workspace, int B, int Nv, int Nq, int nH, int L, int P,
                      int dtype, void* stream);
/* ge_msda_fwd_mm / ge_msda_bwd_lw_mm for the FIRST Nq queries of every image (round 6, ABI 7): raw / out / d_out / d_raw are (B, q_pitch, ...) row
 * matrices with q_pitch >= Nq rows per image, rows beyond Nq are left alone, `order` permutes 0 .. Nq - 1.  Used by the self-attention of
 * depth/models/necks/hahi.py:279-289: its level-0 queries (75 %) run on these MFMA kernels, the coarse-level queries — whose 32-query patches
 * span 16 - 64 level-0 cells — on ge_msda_fwd_raw / ge_msda_bwd_raw (measured: forward + d_raw 3.37 -> 2.16 ms at 8 x 32 725 queries). */
int ge_msda_fwd_mm_part(const void* value, const int* spatial_hw, const void* off_raw, long off_ld, const void* logit_raw, long logit_ld,
                        const float* ref, long ref_sb, long ref_sq, long ref_sl, const int* order, void* out, int B, int Nv, int Nq, int q_pitch,
                        int nH, int L, int P, int dtype, void* stream);
int ge_msda_bwd_lw_mm_part(const void* value, const int* spatial_hw, const void* off_raw, long off_ld, const void* logit_raw, long logit_ld,
                           const float* ref, long ref_sb, long ref_sq, long ref_sl, const int* order, const void* d_out, void* d_off_raw,
                           long d_off_ld, void* d_logit_raw, long d_logit_ld, int B, int Nv, int Nq, int q_pitch, int nH, int L, int P, int dtype,
                           void* stream);
/* d_value of ge_msda_fwd_mm as the transpose of the forward contraction (round 5, ABI 5; csrc/msda_mm.hip: dV_window = C^T dO on the
 * matrix cores, one fp32 atomic flush per RUN of consecutive query tiles that share a window).  Replaces mmcv's
 * ms_deformable_col2im (atomicAdd into grad_value) as called from depth/models/necks/hahi.py:316-325, and the count / scan / fill /
 * drain record pipeline of ge_msda_bwd_value_raw for the cross-attention.  d_value (B, Nv, nH, 64) f32 is ACCUMULATED into
 * (zero-fill it first).  `workspace` (>= ge_msda_bwd_mm_workspace(B, Nq, nH, L) bytes, 0 = unsupported) is shared with
 * ge_msda_bwd_lw_mm: that call — same inputs, same `order`, earlier on the same stream — leaves the per-tile tap boxes in it
 * (its `workspace` argument may be NULL when no ge_msda_bwd_value_mm follows).  The kernel's cost depends on how compact the windows of
 * consecutive query tiles are; d_value == NULL only cuts the runs and leaves two ints {window rows to flush, tile passes} at byte
 * ge_msda_bwd_mm_stats_offset(...) of the workspace (they are also written by a full call), from which a caller can decide between this
 * entry point and ge_msda_bwd_value_raw, whose cost is independent of the geometry. */
size_t ge_msda_bwd_mm_workspace(int B, int Nq, int nH, int L);
size_t ge_msda_bwd_mm_stats_offset(int B, int Nq, int nH, int L);
/* `level_mask` (bit l = value level l): ge_msda_bwd_value_mm adds the d_value rows of those levels only, and
 * ge_msda_bwd_value_raw_levels (the record pipeline restricted the same way) takes the complement — the statistics are per level too
 * (ints 2 + 2 l, 3 + 2 l behind the two totals): coarse levels, where consecutive query tiles share their window, go to the MFMA kernel,
 * fine levels to the records. */
int ge_msda_bwd_value_mm(const int* spatial_hw, const void* off_raw, long off_ld, const void* logit_raw, long logit_ld, const float* ref,
                         long ref_sb, long ref_sq, long ref_sl, const int* order, const void* d_out, float* d_value, void* workspace,
                         size_t workspace_bytes, int level_mask, int B, int Nv, int Nq, int nH, int L, int P, int dtype, void* stream);
/* d_value of ge_msda_fwd_mm with the OUTPUT held still (round 6, ABI 7; csrc/msda_mm.hip: msda_mm_bwd_vs_k).  Same contraction as
 * ge_msda_bwd_value_mm, but a workgroup keeps a 24 x 16 super-block of value positions x 64 channels in MFMA accumulators while it walks
 * the query tiles whose tap boxes reach it (lists built on the device from the boxes ge_msda_bwd_lw_mm leaves in the workspace), and writes the
 * block once: no records, no flush per query tile.  Replaces mmcv's ms_deformable_col2im grad_value (atomicAdd per tap) as called from
 * depth/models/necks/hahi.py:316-325.  d_value (B, Nv, nH, 64) f32 must be ZERO on entry.  `workspace`: ge_msda_bwd_vs_workspace(...) bytes
 * (0 = geometry unsupported), the same buffer that was handed to ge_msda_bwd_lw_mm for these inputs (its head is the
 * ge_msda_bwd_mm_workspace layout).  Query tiles whose taps are spread over more than 12 super-blocks fall back, per tile, to the
 * atomic kernel of ge_msda_bwd_value_mm: correct for any geometry.  Four ints {tile visits, stray tiles, work items, items of multi-chunk
 * super-blocks} of the latest call sit at byte ge_msda_bwd_vs_stats_offset(...) of the workspace. */
size_t ge_msda_bwd_vs_workspace(const int* spatial_hw, int B, int Nv, int Nq, int nH, int L, int P);
size_t ge_msda_bwd_vs_stats_offset(const int* spatial_hw, int B, int Nv, int Nq, int nH, int L, int P);
int ge_msda_bwd_value_vs(const int* spatial_hw, const void* off_raw, long off_ld, const void* logit_raw, long logit_ld, const float* ref,
                         long ref_sb, long ref_sq, long ref_sl, const int* order, const void* d_out, float* d_value, void* workspace,
                         size_t workspace_bytes, int B, int Nv, int Nq, int nH, int L, int P, int dtype, void* stream);
int ge_msda_bwd_value_raw_levels(const int* spatial_hw, const void* off_raw, long off_ld, const void* logit_raw, long logit_ld,
                                 const float* ref, long ref_sb, long ref_sq, long ref_sl, const void* d_out, float* d_value, void* workspace,
                                 size_t workspace_bytes, int level_mask, int B, int Nv, int Nq, int nH, int L, int P, int dtype, void* stream);
int ge_msda_bwd_value(const void* value, const int* spatial_hw, const float* loc, const float* attw, const void* d_out, float* d_value,
                      void* workspace, size_t workspace_bytes, int B, int Nv, int Nq, int nH, int L, int P, int dtype, void* stream);
/* ge_msda_bwd_value fed from the raw projections instead of loc / attw (bf16, L == 4, P == 8; 8-byte records): with it the forward
 * needs no loc / attw output at all. */
int ge_msda_bwd_value_raw(const int* spatial_hw, const void* off_raw, long off_ld, const void* logit_raw, long logit_ld, const float* ref,
                          long ref_sb, long ref_sq, long ref_sl, const void* d_out, float* d_value, void* workspace,
                          size_t workspace_bytes, int B, int Nv, int Nq, int nH, int L, int P, int dtype, void* stream);
int ge_msda_dref(const void* d_off_raw, long off_ld, const int* spatial_hw, float* d_ref, long rows, int nH, int L, int P, int dtype,
                 void* stream);
/* Dropout inside ge_tokens_from_map / ge_map_from_tokens / ge_concat_rows_fwd / ge_slice_rows_drop is a counter-based hash of (seed,
 * element index); the seed is a launch argument, so a launch captured in a hipGraph would replay ONE mask for ever.  ge_rng_salt
 * registers a device counter (NULL = none, the default) whose value, read when the kernel EXECUTES, is mixed into every seed: a captured
 * training step increments it inside the graph (one more process-wide knob, like ge_msda_mode; forward and backward of a step must see the
 * same value, i.e. increment it between steps only).  The ADDRESS is read on the host when a dropout kernel is LAUNCHED and becomes a kernel
 * argument: inside a capture it is baked into the graph, so the slot only has to be set while the capture runs (round 6: GraphedTrainStep sets it
 * before and clears it after the capture; a slot that outlived the capture could be cleared under a newer graph object or point at another device's
 * counter). */
int ge_rng_salt(const unsigned long long* device_counter);
int ge_tokens_from_map(const void* map, long map_bs, const float* pos, void* tok, long tok_bs, int B, int C, long N,
                       float p_drop, unsigned long long seed, int dtype, void* stream);
int ge_map_from_tokens(const void* tok, long tok_bs, const void* res, long res_bs, void* map, long map_bs, int B, int C,
                       long N, float p_drop, unsigned long long seed, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm over the last dimension of a (rows, C) token matrix with mixed precision I/O: x / dx in x_dtype, y / dy in
 * y_dtype (each GE_F32 or GE_BF16), gamma / beta / mean / rstd / dgamma / dbeta f32.  Replaces F.layer_norm on the Swin
 * token path (depthformer_swin.py:461-472 norm1 / norm2, :98-122 PatchMerging.norm, :1166-1172 stage norms;
 * utils/embed.py:282-302) together with the dtype copies autocast puts around it.  C % 4 == 0, C <= 3072.
 * Backward: dgamma / dbeta are ACCUMULATED (zero-fill them first).
 */
int ge_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                     float* mean, float* rstd, long rows, int C, float eps, void* stream);
int ge_layernorm_bwd(const void* dy, int y_dtype, const void* x, int x_dtype, const float* gamma, const float* mean,
                     const float* rstd, void* dx, float* dgamma, float* dbeta, long rows, int C, void* stream);
/* The same with the skip-connection gradient folded in: dx = LN'(dy) + dres (dres: x's dtype, NULL = plain backward).  In a pre-norm
 * block (depthformer_swin.py:396-472: x + attn(LN x)) x feeds the LayerNorm AND the residual; this removes autograd's separate add. */
int ge_layernorm_bwd_res(const void* dy, int y_dtype, const void* x, int x_dtype, const float* gamma, const float* mean,
                         const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta, long rows, int C, void* stream);
/* The same with the d_gamma / d_beta column sums spread over `copies` accumulators (same-address fp32 atomics serialise in L2): dgb =
 * [copies][2][C] floats, zero-filled by the caller, who also sums the copies: d_gamma = sum_k dgb[k][0][:], d_beta = sum_k dgb[k][1][:].
 * 1 <= copies <= 64. */
int ge_layernorm_bwd_multi(const void* dy, int y_dtype, const void* x, int x_dtype, const float* gamma, const float* mean,
                           const float* rstd, const void* dres, void* dx, float* dgb, int copies, long rows, int C, void* stream);
/* The reduction of ge_layernorm_bwd_multi's accumulator copies that leaves them zero again: out (2, C) = sum over copies of dgb (copies, 2, C),
 * dgb := 0 — a persistent accumulator buffer then needs no zero-fill per call (round 5, ABI 6). */
int ge_layernorm_fold(float* dgb, float* out, int copies, int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Residual connection with per-sample stochastic depth: out[b, :] = identity[b, :] + branch[b, :] * scale[b]
 * (`identity + DropPath(branch)` of the Swin blocks, depthformer_swin.py:461-472; scale[b] = Bernoulli(keep) / keep, f32,
 * drawn by the caller).  out has the identity's dtype; supported (identity, branch): (f32, f32), (f32, bf16), (bf16, bf16).
 * ge_scale_rows: out[b, :] = x[b, :] * scale[b] with a dtype change — the gradient of the branch.
 */
int ge_residual_scale_add(const void* identity, int id_dtype, const void* branch, int br_dtype, const float* scale,
                          void* out, int B, long per_sample, void* stream);
int ge_scale_rows(const void* x, int x_dtype, const float* scale, void* out, int out_dtype, int B, long per_sample,
                  void* stream);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm2d in training mode + (leaky-)ReLU, NCHW: the norm -> act tail of mmcv ConvModule (necks/hahi.py:150-166,
 * depthformer_swin.py:1127-1139).  y = act((x - mean_c) * rstd_c * gamma_c + beta_c) with batch statistics over (N, H, W)
 * (biased variance), running_mean / running_var updated with `momentum` (unbiased variance), save_mean / save_rstd
 * returned for the backward pass.  Backward: g = dy * act'(y); dbeta = sum g, dgamma = sum g * xhat,
 * dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)).  x / y / dy / dx in `dtype`, everything per-channel f32.
 * `workspace`: ge_bn_workspace(C) bytes of device scratch (the entry points zero what they need).
 */
size_t ge_bn_workspace(int C);
int ge_bn_act_fwd(const void* x, const float* gamma, const float* beta, void* y, float* save_mean, float* save_rstd,
                  float* running_mean, float* running_var, void* workspace, int N, int C, long HW, float eps,
                  float momentum, float slope, int dtype, void* stream);
int ge_bn_act_bwd(const void* dy, const void* y, const void* x, const float* gamma, const float* save_mean,
                  const float* save_rstd, void* dx, float* dgamma, float* dbeta, void* workspace, int N, int C, long HW,
                  float slope, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Bias + activation after a bias-free convolution, NCHW, in place: x = act(x + bias[c]) with
 * act = leaky-relu(slope) (slope 0 = ReLU, 1 = identity).  Replaces the broadcast bias add + activation kernels of
 * mmcv ConvModule without norm (decode_heads/densedepth_head.py:14-27) and of the PE-neck convs
 * (necks/pemask_neck.py:36-42).  Backward: dx = dy * (y > 0 ? 1 : slope), dbias[c] += sum dx (dbias f32,
 * zero-filled by the caller).  HW = H*W elements per (n, c) plane.
 */
int ge_bias_act_fwd(void* x, const float* bias, int N, int C, long HW, float slope, int dtype, void* stream);
int ge_bias_act_bwd(const void* dy, const void* y, void* dx, float* dbias, int N, int C, long HW, float slope,
                    int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Ground-embedding prior (all fp32).
 * Adaptive: DepthEncoderDecoder.dynamic_pe + the y up-sampling of extract_feat
 * (depth/models/depther/encoder_decoder.py:79-102,111-114).  The bilinear (align_corners=False)
 * up-sampling of the 11 slope logits and of y happens inside the kernel.
 *   logits_lr (B,11,h,w)   y_lr (B,1,h,w)   pe_raw: B planes of (H,W), plane b at pe_raw + b*pe_batch_stride
 *   height    (B) or NULL (=> 1.65)          depth_scale (200)
 * outputs: pe_mask (B,1,H,W)  logits_hr (B,11,H,W)  y_hr (B,1,H,W)  valid_mask u8 (B,H,W) in {0,1}
 * (valid_mask is the "integer pixel mask" pe_offset_mask of :97-100: bit-exact vs the reference).
 */
int ge_ground_embed_fwd(const float* logits_lr, const float* y_lr, const float* pe_raw, long pe_batch_stride,
                        const float* height, float depth_scale,
                        float* pe_mask, float* logits_hr, float* y_hr, uint8_t* valid_mask,
                        int B, int h, int w, int H, int W, void* stream);
/* Backward: given d_pe_mask (B,1,H,W), d_logits_hr (B,11,H,W) (may be NULL), d_y_hr (B,1,H,W) (may be
 * NULL) produce d_logits_lr (B,11,h,w) and d_y_lr (B,1,h,w) (both fully written).
 * scratch: (B,12,H,W) f32 of caller-owned scratch. */
int ge_ground_embed_bwd(const float* logits_lr, const float* y_lr, const float* pe_raw, long pe_batch_stride,
                        const float* height, float depth_scale,
                        const float* d_pe_mask, const float* d_logits_hr, const float* d_y_hr,
                        float* d_logits_lr, float* d_y_lr, float* scratch,
                        int B, int h, int w, int H, int W, void* stream);

/* Vanilla: pe_mask = img[:,3:4] * up(y) * 200  (encoder_decoder.py:111-114,120-123).
 *   pe_norm: B planes (H,W) at pe_norm + b*pe_batch_stride. */
int ge_ground_vanilla_fwd(const float* y_lr, const float* pe_norm, long pe_batch_stride, float gain,
                          float* pe_mask, float* y_hr, int B, int h, int w, int H, int W, void* stream);
int ge_ground_vanilla_bwd(const float* pe_norm, long pe_batch_stride, float gain,
                          const float* d_pe_mask, const float* d_y_hr, float* d_y_lr, float* scratch,
                          int B, int h, int w, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Depth fusion (fp32):  out = relu(c)*(1 - dn(y)) + dn(pe) + min_depth, where dn() is the
 * align_corners=True bilinear resize (H,W)->(h,w) of DepthBaseDecodeHead.depth_pred
 * (depth/models/decode_heads/decode_head.py:489-506).
 *   c (B,1,h,w) conv_depth output BEFORE ReLU;  pe_mask, y_hr (B,1,H,W);  out, y_ds (B,1,h,w).
 */
int ge_depth_fuse_fwd(const float* c, const float* pe_mask, const float* y_hr, float min_depth,
                      float* out, float* y_ds, int B, int h, int w, int H, int W, void* stream);
/* d_c (B,1,h,w), d_pe_mask (B,1,H,W), d_y_hr (B,1,H,W) fully written. scratch: (B,2,h,w) f32. */
int ge_depth_fuse_bwd(const float* c, const float* y_ds, const float* d_out,
                      float* d_c, float* d_pe_mask, float* d_y_hr, float* scratch,
                      int B, int h, int w, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Offline ground-plane maps (fp64 arithmetic, as numpy does in the reference).
 * ge_ground_plane: pe(u,v) = num / (r20*u + r21*v + r22)  (tools/preprocess_data_kitti.py:47-53),
 *   rinv_row2 = host double[3], num = RT[2]-cam_height.  pe_f64 (H,W) and/or pe_f32 (H,W) (either
 *   may be NULL; the f32 copy is the astype(np.float32) of datasets/pipelines/loading.py:375).
 * ge_slope_class: k = h/gt + f32(-h)/pe; round (mode 0, KITTI :59-63,83-89) or trunc (mode 1, DDAD
 *   tools/preprocess_data_ddad.py:78) of deg(atan k), clamp to [-5,5], 255 where gt==0.
 *   gt f64 (H,W) [uint16 PNG / 256], pe f32 (H,W) -> cls int16 (H,W).
 * ge_slope_class_ddad: the DDAD script's own dtypes (tools/preprocess_data_ddad.py:47-51,68-78): gt f32 (the .npz
 *   depth), pe f64 (the map of :35-41, numerator RT[2] without a height term); a = -h/pe in f64, b = h/gt in F32,
 *   k = b + a in f64, truncation (astype(int64)), clamp, 255 where gt==0.
 * ge_pe_channels: loader-side filtering + normalisation (loading.py:397-403, transforms.py:40-48):
 *   raw f32 -> norm f32 (>200 -> 0, <0 -> 0, then / depth_scale where > 0).
 */
int ge_ground_plane(const double* rinv_row2, double num, double* pe_f64, float* pe_f32, int H, int W,
                    void* stream);
int ge_slope_class(const double* gt, const float* pe, double cam_height, int mode, int16_t* cls,
                   int H, int W, void* stream);
int ge_slope_class_ddad(const float* gt, const double* pe, double cam_height, int16_t* cls, int H, int W,
                        void* stream);
int ge_pe_channels(const float* raw, float* norm, float depth_scale, long n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Channels-last (NHWC) variants of the map kernels (csrc/nhwc.hip).  A channels-last map (B, C, H, W) with strides
 * (HWC, 1, WC, C) is the row matrix (rows = B*H*W, C): MIOpen's bf16 implicit-GEMM convolutions run on it without their
 * NCHW <-> NHWC batched_transpose kernels, and tokens <-> maps become views.  C must be a multiple of the 16-byte vector
 * (8 bf16 / 4 f32) and at most 256 such vectors wide, pointers 16-byte aligned; otherwise GE_ERR_UNSUPPORTED (the caller keeps NCHW).
 * ge_bn_act_nhwc_*: as ge_bn_act_* with a workspace of ge_nhwc_workspace(C, 2) bytes; statistics are column sums (fp32 partial per
 *   workgroup, fp64 across workgroups, no atomics).  The backward takes either `y` (forward output) or `beta` (then y may be
 *   NULL and the activation decision y > 0 is recomputed from x with the forward's fma — one tensor less per pass).
 * ge_bias_act_nhwc_*: as ge_bias_act_*; the backward needs a workspace of ge_nhwc_workspace(C, 1) bytes, d_bias is fully written.
 * ge_bilinear_nhwc_*: as ge_bilinear_* on (N, H, W, C); the backward takes an optional workspace of N*Ho*Wi*C floats that
 *   enables the separable two-pass form for up-sampling factors > 3 (PE necks: 11x35 -> 176x560).
 * ge_concat_rows_fwd: out (rows, Ca+Cb) = [a * dropout + res | b] (a first) or [b | a * dropout + res]; `a` is B batches
 *   of rows_per_batch packed rows with a free batch stride (a token range of a longer sequence), res / b / out dense:
 *   torch.cat([to_map(dropout(tokens)) + identity, fmap], 1) of necks/hahi.py:326-346 on channels-last maps.
 * ge_slice_rows_drop: its backward w.r.t. a: d_a (rows, Ca) = d_out[:, off_a : off_a + Ca] * dropout.
 * ge_add_rows: out (B, N, C) = x + pos (N, C) f32 broadcast over the batch (query + query_pos, hahi.py:303-306).
 * ge_colsum: out (C) f32 (+)= column sums of x (R, C) f32 / bf16 — the bias gradient of the token Linears
 *   (depthformer_swin.py:193-221,451-459; hahi.py value / offset / attention / output projections); workspace: ge_nhwc_workspace(C, 1) bytes.
 */
int ge_bn_act_nhwc_fwd(const void* x, const float* gamma, const float* beta, void* y, float* save_mean, float* save_rstd,
                       float* running_mean, float* running_var, void* workspace, long rows, int C, float eps,
                       float momentum, float slope, int dtype, void* stream);
int ge_bn_act_nhwc_bwd(const void* dy, const void* y, const void* x, const float* gamma, const float* beta,
                       const float* save_mean, const float* save_rstd, void* dx, float* dgamma, float* dbeta, void* workspace,
                       long rows, int C, float slope, int dtype, void* stream);
int ge_bias_act_nhwc_fwd(void* x, const float* bias, long rows, int C, float slope, int dtype, void* stream);
int ge_bias_act_nhwc_bwd(const void* dy, const void* y, void* dx, float* dbias, void* workspace, long rows, int C,
                         float slope, int dtype, void* stream);
int ge_bilinear_nhwc_fwd(const void* in, void* out, int N, int C, int Hi, int Wi, int Ho, int Wo, int align_corners,
                         int dtype, void* stream);
int ge_bilinear_nhwc_bwd(const void* d_out, void* d_in, void* workspace, size_t workspace_bytes, int N, int C, int Hi, int Wi,
                         int Ho, int Wo, int align_corners, int dtype, void* stream);
int ge_concat_rows_fwd(const void* a, long rows_per_batch, long a_batch_stride, const void* res, const void* b, void* out,
                       long rows, int Ca, int Cb, int a_first, float p_drop, unsigned long long seed, int dtype,
                       void* stream);
int ge_slice_rows_drop(const void* d_out, void* d_a, long rows, int Ca, int Co, int off_a, float p_drop,
                       unsigned long long seed, int dtype, void* stream);
int ge_add_rows(const void* x, const float* pos, void* out, int B, long N, int C, int dtype, void* stream);
int ge_colsum(const void* x, long R, int C, float* out, void* workspace, int accumulate, int dtype, void* stream);

/* bias + GELU epilogue of the FFN's first Linear (mmcv FFN inside depth/models/backbones/depthformer_swin.py:451-459; GELU = exact
 * erf form).  x (R, C) is the BIAS-FREE GEMM output (f32 / bf16, C a multiple of 4 / 8), bias (C) f32 or NULL.
 *   ge_bias_gelu_fwd: out = gelu(x + bias)
 *   ge_bias_gelu_bwd: dy = dg * gelu'(x + bias) (storage type) and d_bias (C) f32 = column sums of dy, one sweep;
 *                     workspace: ge_nhwc_workspace(C, 1) bytes. */
int ge_bias_gelu_fwd(const void* x, const float* bias, void* out, long R, int C, int dtype, void* stream);
int ge_bias_gelu_bwd(const void* dg, const void* x, const float* bias, void* dy, float* d_bias, void* workspace, long R, int C,
                     int dtype, void* stream);

/* Bandwidth-bound decoder glue fused into one pass each (csrc/decoder.hip), channels-last maps (N, H, W, C), C multiples of the
 * 16-byte vector, f32 / bf16:
 *   ge_upcat_nhwc_fwd  out (N,H,W,Cu+Cs) = [bilinear(coarse (N,Hc,Wc,Cu) -> H x W) | skip (N,H,W,Cs)]: the UpSample block of the
 *                      DenseDepth head (depth/models/decode_heads/densedepth_head.py:25-27: F.interpolate -> torch.cat) without the
 *                      up-sampled intermediate and without ATen's cat;
 *   ge_upcat_nhwc_bwd  d_coarse from the d_up columns of d_out read in place (row pitch Cu + Cs); d_skip is a channel slice of d_out;
 *   ge_upsum_nhwc_fwd  out = fine + sum_i bilinear(src_i -> H x W), nsrc <= 4: the trunk of the PE necks
 *                      (depth/models/necks/pemask_neck.py:52-64, dynamicpe_neck.py:512-539); partial sums are rounded to the storage
 *                      type in the reference's order.  hw = {H_0, W_0, H_1, W_1, ...}.  Backward: ge_bilinear_nhwc_bwd per source. */
int ge_upcat_nhwc_fwd(const void* coarse, const void* skip, void* out, int N, int Cu, int Hc, int Wc, int Cs, int H, int W,
                      int align_corners, int dtype, void* stream);
int ge_upcat_nhwc_bwd(const void* d_out, void* d_coarse, int N, int Cu, int Hc, int Wc, int Cs, int H, int W, int align_corners,
                      int dtype, void* stream);
int ge_upsum_nhwc_fwd(const void* const* srcs, const int* hw, int nsrc, const void* fine, void* out, int N, int C, int H, int W,
                      int align_corners, int dtype, void* stream);

/* 3x3 / stride 1 / pad 1 convolution on channels-last bf16 maps as an implicit GEMM on v_mfma_f32_32x32x16_bf16 (csrc/conv3x3.hip), with
 * the bias + (Leaky)ReLU epilogue of mmcv's ConvModule: y (N,H,W,Cout) = act(conv(x (N,H,W,Cin), w (Cout,3,3,Cin)) + bias).  `w` is in the
 * storage order of a channels-last conv weight (O, H, W, I); act = 0: none, 1: leaky-ReLU(slope) (slope 0 = ReLU); bias f32 or NULL.
 * Cin % 32 == 0, Cout % 8 == 0, dtype GE_BF16.  The data gradient of the same layer is this call on d_y with the flipped / transposed
 * weights w'[ci, r, s, co] = w[co, 2 - r, 2 - s, ci].  Reference call sites: decode_heads/densedepth_head.py:14-27, necks/hahi.py:140-165. */
int ge_conv3x3_nhwc_fwd(const void* x, const void* w, const float* bias, void* y, int N, int H, int W, int Cin, int Cout, int act,
                        float slope, int dtype, void* stream);
/* Weight gradient of the same layer on the matrix cores (csrc/conv3x3_wgrad.hip): dw (Cout, 3, 3, Cin) fp32 [(O, H, W, I) order] +=
 * sum over pixels of dy (N,H,W,Cout) x shifted x (N,H,W,Cin); both bf16 channels-last; the CALLER zero-fills dw (partial sums of the
 * K-split workgroups meet through fp32 atomics).  Cin % 32 == 0, Cout % 8 == 0. */
int ge_conv3x3_nhwc_wgrad(const void* x, const void* dy, float* dw, int N, int H, int W, int Cin, int Cout, int dtype, void* stream);
/* Weight gradient of a 1x1 convolution on channels-last bf16 maps (csrc/conv1x1_wgrad.hip; reference layers: the ConvModule(k = 1) blocks
 * of necks/hahi.py:120-166): dw (Cout, Cin) fp32 += dy^T x over the M = N H W rows; x (M, Cin), dy (M, Cout) row-major; the CALLER zero-fills
 * dw.  Cin % 64 == 0 or Cin % 96 == 0, Cout % 32 == 0, dtype GE_BF16. */
int ge_conv1x1_nhwc_wgrad(const void* x, const void* dy, float* dw, long M, int Cin, int Cout, int dtype, void* stream);
/* 1x1 convolution + training-mode BatchNorm2d + (Leaky)ReLU [+ positional embedding] of a 64-channel channels-last bf16 map in one pass
 * over the output (csrc/conv1x1_bn.hip) — the conv_proj ConvModule of the HAHI neck and the cross-attention query built from it (reference
 * depth/models/necks/hahi.py:151-157 and :294-306: query = flatten(conv_proj(x)) + pos).  The BatchNorm batch statistics of z = x W^T are
 * derived from the moments of the INPUT (mean_c = w_c . m, E[z_c^2] = w_c^T S w_c), so z is never stored.  Cin == 64, Cout % 128 == 0.
 *   ge_conv1x1_bn_workspace  bytes of the scratch buffer the stats / mask calls need (0: unsupported widths).
 *   ge_conv1x1_bn_stats      x (rows, 64) bf16, w (Cout, 64) bf16 -> gram (65 * 64 doubles: sum x x^T | sum x, kept for the backward),
 *                            save_mean / save_rstd (Cout), coef (2 Cout: scale | shift); running statistics updated like F.batch_norm (or NULL).
 *   ge_conv1x1_bn_act_fwd    y (B * HW, Cout) bf16 = act(coef_a * (x W^T) + coef_b); q = y + pos[p] (pos (HW, Cout) fp32; q and pos NULL together).
 *   ge_conv1x1_bn_bwd_mask   g (rows, C) bf16 = (dy1 + dy2) * act'(y) and its column sums; dy1 / dy2 bf16 with row strides ld1 / ld2 (elements),
 *                            either may be NULL — a gradient that arrives as a channel slice of a wider map is read in place.
 *   ge_conv1x1_bn_bwd_finalize  from GT = g^T x ((Cout, 64) fp32: ge_conv1x1_nhwc_wgrad), the column sums and gram: d_gamma, d_beta, dW (Cout, 64)
 *                            fp32, and the operands of the data gradient: Wd (64, Cout + 64) bf16, c0 (64) fp32; scratch 2 Cout floats.
 *   ge_conv1x1_bn_dgrad      dx (rows, 64) bf16 = [g | x] Wd^T + c0 in one pass over g and x (Wd resident in LDS; Cout <= 1024). */
size_t ge_conv1x1_bn_workspace(int Cin, int Cout);
int ge_conv1x1_bn_stats(const void* x, long rows, int Cin, const void* w, int Cout, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, float eps, float momentum, double* gram, float* save_mean, float* save_rstd, float* coef,
                        void* workspace, void* stream);
int ge_conv1x1_bn_act_fwd(const void* x, const void* w, const float* coef, const float* pos, void* y, void* q, int B, long HW, int Cin, int Cout,
                          float slope, void* stream);
int ge_conv1x1_bn_bwd_mask(const void* dy1, long ld1, const void* dy2, long ld2, const void* y, void* g, float* colsum, void* workspace, long rows,
                           int C, float slope, void* stream);
int ge_conv1x1_bn_bwd_finalize(const float* GT, const float* colsum, const double* gram, const void* w, const float* gamma, const float* save_mean,
                               const float* save_rstd, long rows, int Cin, int Cout, float* dgamma, float* dbeta, float* dW, void* Wd, float* c0,
                               float* scratch, void* stream);
int ge_conv1x1_bn_dgrad(const void* g, const void* x, const void* Wd, const float* c0, void* dx, long rows, int Cin, int Cout, void* stream);
/* The same layer with ONE output channel (csrc/conv3x3_c1.hip): the depth regressor `conv_depth` (reference
 * depth/models/decode_heads/decode_head.py: nn.Conv2d(channels, 1, 3, padding=1)) and `convfinal` of the ground-attention neck
 * (necks/pemask_neck.py:36-42): a streaming reduction on the vector pipe (v_dot2c_f32_bf16), not a GEMM with N = 1.
 * fwd: y (N,H,W) [out_dtype GE_BF16 | GE_F32] = bias[0] + conv(x (N,H,W,Cin) bf16, w (1,3,3,Cin) f32 rounded to bf16 as autocast casts it).
 * bwd: ONE pass gives dx (N,H,W,Cin) bf16, dw (1,3,3,Cin) f32 and db (1) f32 (NULL: skipped) from dy (N,H,W) [dy_dtype]; dw / db are
 * zero-filled by the call.  Cin % 8 == 0, Cin <= 1024. */
int ge_conv3x3_c1_fwd(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int Cin, int out_dtype, void* stream);
int ge_conv3x3_c1_bwd(const void* x, const void* dy, const float* w, void* dx, float* dw, float* db, int N, int H, int W, int Cin,
                      int dy_dtype, void* stream);
/* ---------------------------------------------------------------------------------------------
 * Token GEMM (csrc/gemm.hip): C[M, N] = A[M, K] B[N, K]^T + bias[N] — torch.nn.functional.linear over a token matrix and,
 * with B = the transposed weight, its input gradient.  Replaces the library GEMM behind the reference's nn.Linear layers of the
 * Swin blocks (depth/models/backbones/depthformer_swin.py:193,221 qkv / proj, :451-459 FFN) and of mmcv's
 * MultiScaleDeformableAttention in the HAHI neck (depth/models/necks/hahi.py:279-289,316-325).
 *   A (M rows, leading dimension lda elements), B (N rows, ldb), C (M rows, ldc): bf16, 16-byte aligned, ld % 8 == 0, N % 8 == 0,
 *   K % 8 == 0, operands < 4 GB each; bias (N) f32 or NULL; fp32 accumulation, bias added before the single rounding to bf16.
 *   GE_ERR_UNSUPPORTED for anything else (the caller keeps the library GEMM for those). */
int ge_gemm_nt(const void* A, long lda, const void* B, long ldb, const float* bias, void* C, long ldc, long M, int N, int K, int dtype,
               void* stream);
/* bytes of `workspace` for the channels-last column-sum users: K = 2 for ge_bn_act_nhwc_*, K = 1 for ge_bias_act_nhwc_bwd / ge_colsum */
size_t ge_nhwc_workspace(int C, int K);

/* ---------------------------------------------------------------------------------------------
 * Device-side data pipeline of the training samples (csrc/aug.hip).  Planar f32 maps (C, H, W); each
 * entry point restates one host transform of the reference's KITTI train pipeline
 * (configs/depthformer/depthformer_v.py:13-28 -> depth/datasets/pipelines/transforms.py / loading.py), applied with the
 * same parameters to the 5 image channels (B, G, R, filtered ground depth, raw ground depth), the depth and the class map.
 * ge_aug_load:  HWC uint8 BGR image + (H, W) ground depth -> (5, Hc, Wc) window at (top, left): LoadImageFromFile's USEPE
 *   branch (loading.py:366-403: channel 3 = pe with > pe_max or < 0 zeroed, channel 4 = raw) + KBCrop (transforms.py:150-205).
 * ge_aug_depth: uint16 PNG window -> metres (float32(png) / depth_scale, loading.py:136-140).
 * ge_aug_resize: mode 1 bilinear (half-pixel centres, edge replication: mmcv.imresize / cv2.INTER_LINEAR), mode 0 nearest
 *   (min(floor(dst * in / out), in - 1)); transforms.py:485-733 Resize.
 * ge_aug_rotate: inverse affine warp, constant border (mmcv.imrotate, transforms.py:209-297); inv6 = host float[6]
 *   {a00, a01, off0, a10, a11, off1} of dst -> src; mode as above (nearest rounds half to even).
 * ge_aug_window: dst[c,y,x] = src[c, y+oy, x+ox] (source mirrored horizontally first when flip) or `fill` outside: Padding
 *   (:65-111), RandomFlip (:300-354) and RandomCrop (:357-418) as index arithmetic.
 * ge_aug_color_normalize: ColorAug (:421-482; gamma / brightness f32, colours3 host double[3]) then Normalize (:13-62):
 *   truncate to uint8, BGR -> RGB, (x - mean) * (1 / std) in f64 (mean3 / std3 host double[3]), channel 3 / depth_scale
 *   where positive, channel 4 unchanged.  src, dst (5, H, W).
 */
int ge_aug_load(const uint8_t* bgr_hwc, const float* pe, float* dst, int H, int W, int top, int left, int Hc, int Wc,
                float pe_max, void* stream);
int ge_aug_depth(const uint16_t* png, float* dst, int H, int W, int top, int left, int Hc, int Wc, float depth_scale,
                 void* stream);
int ge_aug_resize(const float* src, float* dst, int C, int Hs, int Ws, int Hd, int Wd, int mode, void* stream);
int ge_aug_rotate(const float* src, float* dst, int C, int H, int W, const float* inv6, float border, int mode,
                  void* stream);
int ge_aug_window(const float* src, float* dst, int C, int Hs, int Ws, int Hd, int Wd, int oy, int ox, int flip,
                  float fill, void* stream);
int ge_aug_color_normalize(const float* src, float* dst, int H, int W, int color_on, float gamma, float brightness,
                           const double* colors3, const double* mean3, const double* std3, float depth_scale,
                           int to_rgb, void* stream);
/* DDAD front end of the device pipeline (DDADResize, depth/datasets/pipelines/transforms.py:735-783):
 *   ge_aug_area_u8  (H, W, 3) uint8 -> (3, Ho, Wo) planar f32 = cv2.INTER_AREA shrink (pixel-area averaging, float64 weights, rint + clip)
 *   ge_aug_splat    sparse map (H, W) f32 -> (Ho, Wo): every pixel > 0 re-projected to int(coord * scale); the last source pixel in
 *                   row-major order wins (deterministic gather), the rest is 0
 * (the ground-depth channels take ge_aug_resize with mode 0 = nearest). */
int ge_aug_area_u8(const uint8_t* src_hwc, float* dst, int H, int W, int Ho, int Wo, void* stream);
int ge_aug_splat(const float* src, float* dst, int H, int W, int Ho, int Wo, void* stream);

/* ---------------------------------------------------------------------------------------------
 * SiLog loss statistics (fp32 in, fp64 accumulate), SigLoss.sigloss
 * (depth/models/losses/sigloss.py:36-53) without the dynamic-shape boolean gather:
 * over valid = gt > 0:  stats[0] = n, stats[1] = sum g, stats[2] = sum g^2 with
 * g = log(pred+eps) - log(gt+eps).  `stats` (3 doubles) must be zeroed by the caller.
 * Backward: d_pred = valid ? coef_a[0] * g + coef_b[0] : 0, divided by (pred+eps); coef_* are
 * device scalars computed by the host wrapper from the statistics (no host sync). */
int ge_silog_stats(const float* pred, const float* gt, float eps, double* stats, long n, void* stream);
int ge_silog_bwd(const float* pred, const float* gt, float eps, const float* coef_a, const float* coef_b,
                 float* d_pred, long n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused AdamW over a flat fp32 parameter arena (torch.optim.AdamW semantics as configured by
 * configs/depthformer/depthformer_v.py:128-140 through mmcv's DefaultOptimizerConstructor) with the
 * L2 gradient clip of OptimizerHook(grad_clip=dict(max_norm=35)) folded in.
 *   ge_sumsq: out[0] += sum(x^2) (fp64; caller zeroes `out`).
 *   ge_adamw_step: for i<n: g = grad[i]*min(1, max_norm/(sqrt(gnorm_sq[0])+1e-6));
 *       p *= 1 - lr*wd[seg(i)];  m,v updates;  p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
 *     decay is per element via `wd_mask` (u8: 1 = apply weight decay) to honour paramwise decay_mult=0.
 *     lr / step-dependent scalars are read from a small device array `hyper` =
 *     {lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, max_norm, 1-beta1, 1-beta2} (10 floats; the last two
 *     rounded from double on the host, as torch.optim does) so that the launch can
 *     be captured in a hipGraph and replayed while the host updates `hyper`.
 */
int ge_sumsq(const float* x, long n, double* out, void* stream);
int ge_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const uint8_t* wd_mask,
                  const float* hyper, const double* gnorm_sq, long n, void* stream);
/* same step, and the updated parameters rounded to bf16 (nearest-even) into `shadow_bf16` (n elements): the low-precision
 * weights the next autocast forward reads, instead of one cast kernel per tensor and step. */
int ge_adamw_step_shadow(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const uint8_t* wd_mask,
                         const float* hyper, const double* gnorm_sq, long n, void* shadow_bf16, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GEDEPTH_HIP_H */
