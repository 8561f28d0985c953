// Deformable attention as matrix products on wave-private LDS windows (gfx950, bf16 storage).
// Same op as msda_fwd_k / msda_fwd_win_k (mmcv MultiScaleDeformableAttention.forward from the raw projection outputs: view ->
// softmax over L*P -> ref + off / (W_l, H_l) -> bilinear sampling; reference call sites depth/models/necks/hahi.py:279-289,316-325),
// third decomposition.
//
// Why: the gather kernels spend ~120 VALU + 4 row gathers per sampling point and 8-lane group and sit at the L2 -> L1 row-gather rate
// (103 GB of 16-byte-per-lane gathers per cross-attention launch).  But for a TILE of 32 queries whose sampling points
// land in a compact window of a level (K value rows), the sampling IS a contraction
//
//     out[q, ch] = sum_r C[q, r] * V[r, ch]          q: 32 queries, r: K window rows, ch: 64 channels of the head
//
// with C the 32 x K image of bilinear-times-attention coefficients (<= 32 non-zeros per row and level).  One
// v_mfma_f32_32x32x16_bf16 takes 16 window rows x 32 channels for all 32 queries; the per-point work shrinks to the tap arithmetic
// (done ONCE per point by the lane that owns it, not by 8 lanes) plus four 2-byte LDS read-modify-writes that drop the
// coefficients into C.  The value rows of the window cross L2 -> LDS once per tile (K x 128 B) instead of once per tap.
//
//   wave          = (image, head, tile of 32 queries in the caller's ORDER) — no workgroup cooperation, no barriers
//   lane          = (query q = lane & 31, hv = lane >> 5): the 8 points of ONE level per pass; pass s handles levels (s, s + 2):
//                   lanes 0-31 own level s, lanes 32-63 level s + 2, and the two windows are CONCATENATED along K into one
//                   coefficient image — a lane's row q is touched by its own points only (no atomics, no cross-lane collisions:
//                   the two lanes of a query write disjoint column ranges), and no MFMA work is duplicated
//   window        = bounding box of the in-map taps of the 32 x 8 points of a (tile, head, level), found with two packed 16-bit
//                   min / max butterflies per half wave; windows wider than the image (CAP columns) are walked in chunks (the
//                   scatter is predicated on the chunk), so correctness never depends on locality — only speed does
//   coherence     = the caller passes the query ORDER (a permutation): 2-D tiles of the token maps for the self-attention, the
//                   queries sorted by the cell of their (content-independent) reference point for the cross-attention
//                   (hahi.py:294-302); the kernel reads / writes rows through it, so the sort costs no extra pass
//   B operand     = value rows parked [row][channel] (gathered 8 rows x 128 B per instruction) and read with the transposing LDS
//                   read ds_read_b64_tr_b16, the layout of msda_drain_mfma.hip
// Numerics: coefficients are accumulated in the bf16 image (each add rounds to bf16: 2^-9 relative, the rounding class of the
// window kernel's bf16 tap weights), products exact, sums fp32, locations fp32 in mmcv's own arithmetic (IEEE
// division: the kink decisions of floor() match the gather kernels bit for bit).  The exact-fp32 parity path stays on the gather kernels.
#include "msda.h"
#include <limits.h>
#include <algorithm>

typedef __bf16 mm_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 mm_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 mm_bf16x2 __attribute__((ext_vector_type(2)));
typedef float mm_f32x16 __attribute__((ext_vector_type(16)));
typedef short mm_s16x2 __attribute__((ext_vector_type(2)));

#ifndef MM_CAP
#define MM_CAP 64                        // columns of the coefficient image = value rows staged per chunk (multiple of 16)
#endif
#define MM_HALF (MM_CAP * 32 + 32)       // bf16 elements of one channel-half image of the stage: [rows][32 channels] + 64 B skew
#define MM_STAGE (2 * MM_HALF)
#define MM_IMG ((MM_CAP + 4) * 32)       // coefficient image, TRANSPOSED: [window row k][32 queries] bf16 (+ 4 dump rows), see the kernel
#ifndef MF_CAP
#define MF_CAP 96                        // the same for the FORWARD kernel.  With the model's reference points (un-normalised sine embedding: a 32-query
                                         // tile spans ~4 x 3 level-0 cells) the two levels of a pass need ~80 rows: one chunk of 96 instead of two of 64:
                                         // forward 2.27 -> 2.02 ms (80: 2.09, 128: 2.39); the d_raw kernel is fastest at 64 (2.36; 96: 2.69) — its fp32
                                         // S^T image is twice the size per row (tools/ubench/msda_mm/dv_variants.sh, round 5)
#endif
#define MF_HALF (MF_CAP * 32 + 32)
#define MF_STAGE (2 * MF_HALF)
#define MF_IMG ((MF_CAP + 4) * 32)
#ifndef MM_WAVES
#define MM_WAVES 2                       // occupancy target per SIMD (registers); LDS allows 160 KB / (image + stage) per CU.  Measured with 3
                                         // (168 VGPRs: 28 / 56 spilled registers in forward / d_raw): forward 2.19 vs 2.20 ms, d_raw 3.29 vs 2.43 ms
                                         // (tools/ubench/ab_mm_waves.sh, round 4): the third wave does not pay for its scratch traffic
#endif
#ifndef MM_DBUF
#define MM_DBUF 0                        // 1: two stage buffers (the next chunk's LDS-DMA runs under this chunk's MFMAs); measured: the
                                         // LDS it costs (7 instead of 8 waves per CU) loses more than the overlap wins on the cross-attention
#endif
#ifndef MM_ATOMIC
#define MM_ATOMIC 0                      // 1: coefficients dropped with ds_pk_add_bf16 instead of read-modify-write
#endif
#ifndef MM_DIAG
#define MM_DIAG 0                        // measurement aid: 1 no gathers, 2 no scatter, 4 no MFMA, 8 no output stores
#endif
#define MM_LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

__device__ __forceinline__ bf16_t mm_bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }   // v_cvt_pk_bf16_f32, RNE

// n / d for a bf16-valued n and an integer-valued d <= 8191, given r = RN(1 / d): see mm_taps
__device__ __forceinline__ float mm_div(float n, float d, float r) {
  const float q0 = n * r;
  return __builtin_fmaf(__builtin_fmaf(-q0, d, n), r, q0);
}

struct MmArgs {
  const bf16_t* value; MsdaLevels lv;
  const bf16_t* off; long off_ld; const bf16_t* logit; long logit_ld;       // (B*Nq, ld) rows; columns (head, level, point[, xy])
  const float* ref; long ref_sb, ref_sq, ref_sl;                              // reference points (B, Nq, L, 2), element strides
  const int* order;                                                           // query order (Nq) or NULL = identity
  bf16_t* out; float* loc_out; float* attw_out;                               // loc_out / attw_out may be NULL
  int B, Nv, Nq, nH, ntiles;
  int qpitch;                                                                 // rows per image of the (B * rows) tensors raw / out / d_out / d_raw: Nq, or more when only the first Nq queries of each image are processed (the *_part entry points)
};

// packed (x, y) 16-bit min / max over the 32 lanes of a half wave on the DPP network (row_shr 1, 2, 4, 8 inside the 16-lane rows, then
// row_bcast:15 into rows 1 and 3): lanes 31 and 63 end up with their halves' results.  No LDS round trips (ds_bpermute: ~100 cycles each).
template <bool MAXOP> __device__ __forceinline__ int mm_pk(int a, int b) {
  const mm_s16x2 x = __builtin_bit_cast(mm_s16x2, a), y = __builtin_bit_cast(mm_s16x2, b);
  return __builtin_bit_cast(int, MAXOP ? __builtin_elementwise_max(x, y) : __builtin_elementwise_min(x, y));
}
template <bool MAXOP> __device__ __forceinline__ int mm_half_reduce(int v) {
  v = mm_pk<MAXOP>(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));   // row_shr:1
  v = mm_pk<MAXOP>(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));   // row_shr:2
  v = mm_pk<MAXOP>(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));   // row_shr:4
  v = mm_pk<MAXOP>(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));   // row_shr:8
  v = mm_pk<MAXOP>(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
  return v;
}

// Per-lane state of the 8 points of one (query, head, level): window row of corner 00 with the +1 column / +1 row flags, and the four
// factor weights (x weights carry the attention weight; masked corners are zero).
typedef _Float16 mm_f16x2 __attribute__((ext_vector_type(2)));
struct MmTaps {
  int pk[8];                    // before the box is known: xa | ya << 15; after: r00 (30 bits); always dx << 30 | dy << 31
  mm_f16x2 wt[8], wb[8];        // the four corner coefficients (bilinear x attention weight; masked corners zero) as f16 pairs:
};                              // (w00, w01), (w10, w11) — 2^-11 relative, below the bf16 rounding of the image they are added into

// Softmax over the 32 logits of (query, head): the lane holds the 16 of its two levels, the partner lane (lane ^ 32) the rest.
__device__ __forceinline__ void mm_softmax(const MmArgs& a, long row, int head, int hv, bool qok, float w[2][8]) {
  float m = -INFINITY;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const uint4 t = *(const uint4*)(a.logit + row * a.logit_ld + head * 32 + (s + 2 * hv) * 8);
    const uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { w[s][2 * i] = __uint_as_float(u[i] << 16); w[s][2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u); }
#pragma unroll
    for (int i = 0; i < 8; ++i) m = fmaxf(m, w[s][i]);
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) { w[s][i] = __expf(w[s][i] - m); sum += w[s][i]; }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = qok ? 1.f / sum : 0.f;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) w[s][i] *= inv;
}

// Tap geometry of the lane's 8 points of level `lvl` (pixel sizes Wl x Hl) from the raw offsets u[8] (bf16 pairs) and the reference
// point.  Returns the packed box contributions: bmin = (min ya << 16 | min xa), bmax = (max yb << 16 | max xb) over the in-map points
// (INT16 extremes when there are none).
template <bool LOC>
__device__ __forceinline__ void mm_taps(const MmArgs& a, const uint32_t* u, float rx, float ry, long row, int head, int lvl, int Wl, int Hl,
                                        bool qok, const float* aw, MmTaps& tp, int& bmin, int& bmax) {
  const float fW = (float)Wl, fH = (float)Hl;
  const float rW = 1.f / fW, rH = 1.f / fH;                               // correctly rounded (IEEE division), once per lane and level
  float* lp = (LOC && qok) ? a.loc_out + ((row * a.nH + head) * 32 + lvl * 8) * 2 : nullptr;
  int xmn = 32767, ymn = 32767, xmx = -32768, ymx = -32768;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const float ox = __uint_as_float(u[p] << 16), oy = __uint_as_float(u[p] & 0xffff0000u);
    // mmcv's arithmetic to the bit (`off / W` as msda_prep_fwd_k computes it): at initialisation the self-attention samples EXACT
    // pixel positions, where floor() — i.e. which one-sided derivative the offsets get — is decided by the last bit of the location.
    // The quotient is formed as q0 = off * RN(1 / W), q = fma(fma(-q0, W, off), RN(1 / W), q0): correctly rounded (Markstein), and
    // bit-identical to IEEE division for EVERY bf16 offset and every W <= 8191 (exhaustive check: tools/ubench/msda_mm/divcheck.c) at 3
    // instead of ~10 instructions
    const float lx = rx + mm_div(ox, fW, rW), ly = ry + mm_div(oy, fH, rH);
    if (LOC && lp) *(float2*)(lp + 2 * p) = make_float2(lx, ly);
    const float x = lx * fW - 0.5f, y = ly * fH - 0.5f;                 // grid_sample, align_corners=False
    const bool in = qok && y > -1.f && x > -1.f && y < fH && x < fW;     // NaN-safe
    const float xc = fminf(fmaxf(x, -1.f), fW), yc = fminf(fmaxf(y, -1.f), fH);
    const float xf = floorf(xc), yf = floorf(yc);
    const int x0 = (int)xf, y0 = (int)yf;
    const float ax = xc - xf, ay = yc - yf;
    const int xa = min(max(x0, 0), Wl - 1), xb = min(max(x0 + 1, 0), Wl - 1);
    const int ya = min(max(y0, 0), Hl - 1), yb = min(max(y0 + 1, 0), Hl - 1);
    const float wgt = in ? aw[p] : 0.f;
    const float wxa = x0 >= 0 ? (1.f - ax) * wgt : 0.f, wxb = x0 + 1 < Wl ? ax * wgt : 0.f;
    const float wya = y0 >= 0 ? 1.f - ay : 0.f, wyb = y0 + 1 < Hl ? ay : 0.f;
    tp.wt[p] = __builtin_bit_cast(mm_f16x2, __builtin_amdgcn_cvt_pkrtz(wya * wxa, wya * wxb));
    tp.wb[p] = __builtin_bit_cast(mm_f16x2, __builtin_amdgcn_cvt_pkrtz(wyb * wxa, wyb * wxb));
    tp.pk[p] = xa | (ya << 15) | ((xb - xa) << 30) | ((yb - ya) << 31);      // xa, ya < 2^15 (launcher)
    if (in) { xmn = min(xmn, xa); xmx = max(xmx, xb); ymn = min(ymn, ya); ymx = max(ymx, yb); }
    if (p & 1) __builtin_amdgcn_sched_barrier(0);                        // two points in flight, not eight: ~15 live temporaries per point
  }
  bmin = (ymn << 16) | (xmn & 0xffff);
  bmax = (ymx << 16) | (xmx & 0xffff);
}

// Window geometry of the two levels of a pass (wave-uniform) and the gather of one band of its value rows.
struct MmWin {
  int xminA, yminA, bwA, KA, WA, xminB, yminB, bwB, KB, WB, Ktot;
  float ibwA, ibwB;
  int startA, startB;                       // first value row of the two levels
  const bf16_t* vb;                         // value rows of (image, head): + row * nH * 64
};
// Stage the value rows [c0, c0 + 16 * n16) of the concatenated window into LDS with the LDS-DMA (global_load_lds_dwordx4: no staging
// registers; destination = wave-uniform base + lane * 16 bytes).  One instruction moves 16 rows x 64 bytes of ONE channel half: lane ->
// (row = lane / 4, 16-byte piece = lane % 4), which is exactly a 1 KB run of the stage's [half][row][32 channels] image.  Rows past the
// window read a row that exists (their coefficient columns are zero).
template <int HALF>
__device__ __forceinline__ void mm_stage_rows(const MmWin& w, int c0, int n16, int lane, int nh64, bf16_t* stage) {
  const int piece = (lane & 3) * 8;
  for (int j = 0; j < n16; ++j) {
    const int rr = min(c0 + j * 16 + (lane >> 2), w.Ktot - 1);
    const bool inA = rr < w.KA;
    const int t = inA ? rr : rr - w.KA;
    const int bw = inA ? w.bwA : w.bwB;
    const int ry = (int)(((float)t + 0.5f) * (inA ? w.ibwA : w.ibwB));
    const int rx = t - mul24(ry, bw);
    const int pix = inA ? mul24(w.yminA + ry, w.WA) + w.xminA + rx + w.startA : mul24(w.yminB + ry, w.WB) + w.xminB + rx + w.startB;
    const bf16_t* src = w.vb + (uint32_t)(mul24(pix, nh64) + piece);      // uniform base + 32-bit element offset (launcher: < 2^31)
    if (!(MM_DIAG & 1)) {
      __builtin_amdgcn_global_load_lds(src, MM_LDS_PTR(void, stage + j * 16 * 32), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(src + 32, MM_LDS_PTR(void, stage + HALF + j * 16 * 32), 16, 0, 0);
    }
  }
}

// The coefficient image is kept k-major ([k][q], 64-byte rows): the scatter of a tile whose points land on the same window row (the
// common case: neighbouring queries, equal offsets) then touches 32 consecutive bf16 — conflict-free — where a q-major image with
// 16-byte-aligned rows puts the 32 lanes on 8 banks (measured: the 4-way conflicts made the scatter the whole kernel's bound).  The A
// operand (8 consecutive k of one query per lane) comes back through the same transposing read as the B operand.
template <bool LOC>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MM_WAVES, MM_WAVES))) msda_mm_fwd_k(MmArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t cimg[MF_IMG];
  __shared__ __attribute__((aligned(16))) bf16_t stage2[(1 + MM_DBUF) * MF_STAGE];   // MM_DBUF: the next chunk's LDS-DMA runs under this one's MFMAs
  const int nh64 = a.nH * 64;
  const long total = (long)a.B * a.nH * a.ntiles;
  // XCD x (= blockIdx % 8) walks the x-th contiguous eighth of the (image, head, tile) list: what it has in flight samples one
  // head of one image (its 4 MB L2 ~ the 4.2 MB value slab of an (image, head) at the KITTI shape)
  const int xcd = blockIdx.x % MSDA_XCDS, jx = blockIdx.x / MSDA_XCDS, nx = gridDim.x / MSDA_XCDS;
  const long lo = total * xcd / MSDA_XCDS, hi = total * (xcd + 1) / MSDA_XCDS;
  for (long item = lo + jx; item < hi; item += nx) {
    // the lane id is laundered per item: otherwise every lane-derived address / index (~50 VGPRs of them) is hoisted out of the
    // persistent loop as loop-invariant and pins the kernel at 2 waves per SIMD
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));
    const int q = lane & 31, hv = lane >> 5;
    const int sub8 = lane & 7, row8 = lane >> 3;
    const int tr_row = (lane >> 5) * 8 + ((lane & 15) >> 2), tr_col = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    const int tile = (int)(item % a.ntiles);
    const int bh = (int)(item / a.ntiles);
    const int head = bh % a.nH, b = bh / a.nH;
    const int qi = tile * 32 + q;
    const bool qok = qi < a.Nq;
    const int qsafe = qok ? qi : tile * 32;                               // a query that exists (weights are zeroed)
    const int qq = a.order ? a.order[qsafe] : qsafe;
    const long row = (long)b * a.qpitch + qq;
    const bf16_t* op = a.off + row * a.off_ld + head * 64 + (2 * hv) * 16;
    const float* rp = a.ref + (long)b * a.ref_sb + (long)qq * a.ref_sq + (long)(2 * hv) * a.ref_sl;
    uint4 o0 = *(const uint4*)op, o1 = *(const uint4*)(op + 8);
    float rx = rp[0], ry = rp[1];
    float aw[2][8];
    mm_softmax(a, row, head, hv, qok, aw);
    mm_f32x16 acc0 = 0.f, acc1 = 0.f;
    const bf16_t* vb = a.value + ((long)b * a.Nv * a.nH + head) * 64;
#pragma unroll 1
    for (int s = 0; s < 2; ++s) {
      const int lvl = s + 2 * hv;
      MmWin w;
      w.WA = a.lv.W[s]; w.WB = a.lv.W[s + 2];
      const int HA = a.lv.H[s], HB = a.lv.H[s + 2];
      const int Wl = hv ? w.WB : w.WA, Hl = hv ? HB : HA;
      MmTaps tp;
      int bmin, bmax;
      {
        const uint32_t u[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
        float awl[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) awl[p] = s ? aw[1][p] : aw[0][p];
        mm_taps<LOC>(a, u, rx, ry, row, head, lvl, Wl, Hl, qok, awl, tp, bmin, bmax);
        if (LOC && qok) {
          float* ap = a.attw_out + (row * a.nH + head) * 32 + lvl * 8;
          *(float4*)ap = make_float4(awl[0], awl[1], awl[2], awl[3]);
          *(float4*)(ap + 4) = make_float4(awl[4], awl[5], awl[6], awl[7]);
        }
        if (s == 0) {                                                     // the second pass's raw offsets: in flight during the first
          o0 = *(const uint4*)(op + 16); o1 = *(const uint4*)(op + 24);
          rx = rp[a.ref_sl]; ry = rp[a.ref_sl + 1];
        }
      }
      bmin = mm_half_reduce<false>(bmin); bmax = mm_half_reduce<true>(bmax);
      // box of each half (wave-uniform scalars): lanes 31 and 63 hold their halves' results
      const int mnA = __builtin_amdgcn_readlane(bmin, 31), mxA = __builtin_amdgcn_readlane(bmax, 31);
      const int mnB = __builtin_amdgcn_readlane(bmin, 63), mxB = __builtin_amdgcn_readlane(bmax, 63);
      w.xminA = (short)(mnA & 0xffff); w.yminA = mnA >> 16;
      w.xminB = (short)(mnB & 0xffff); w.yminB = mnB >> 16;
      const int xmaxA = (short)(mxA & 0xffff), ymaxA = mxA >> 16, xmaxB = (short)(mxB & 0xffff), ymaxB = mxB >> 16;
      const bool anyA = xmaxA >= w.xminA && ymaxA >= w.yminA, anyB = xmaxB >= w.xminB && ymaxB >= w.yminB;
      w.bwA = anyA ? xmaxA - w.xminA + 1 : 0; w.bwB = anyB ? xmaxB - w.xminB + 1 : 0;
      w.KA = anyA ? w.bwA * (ymaxA - w.yminA + 1) : 0; w.KB = anyB ? w.bwB * (ymaxB - w.yminB + 1) : 0;
      w.Ktot = w.KA + w.KB;
      if (w.Ktot == 0) continue;                                          // nothing of this tile samples these levels (uniform)
      w.ibwA = 1.f / (float)max(w.bwA, 1); w.ibwB = 1.f / (float)max(w.bwB, 1);
      w.startA = a.lv.start[s]; w.startB = a.lv.start[s + 2]; w.vb = vb;
      mm_stage_rows<MF_HALF>(w, 0, (min(MF_CAP, w.Ktot) + 15) >> 4, lane, nh64, stage2);      // first chunk in flight under the index arithmetic + scatter
      const int bw_l = hv ? w.bwB : w.bwA;
      {   // window row of corner 00 in the concatenated K axis; points without a tap inside the map point at row 0 with zero weights
        const int xmin = hv ? w.xminB : w.xminA, ymin = hv ? w.yminB : w.yminA, kofs = hv ? w.KA : 0;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const bool live = (__builtin_bit_cast(uint32_t, tp.wt[p]) | __builtin_bit_cast(uint32_t, tp.wb[p])) != 0;   // any non-zero coefficient
          const int k = tp.pk[p];
          const int r00 = mul24(((k >> 15) & 0x7fff) - ymin, bw_l) + ((k & 0x7fff) - xmin) + kofs;
          tp.pk[p] = live ? ((k & 0xc0000000) | r00) : 0;
        }
      }
      bf16_t* crow = cimg + q;
#pragma unroll 1
      for (int c0 = 0, ci = 0; c0 < w.Ktot; c0 += MF_CAP, ci ^= 1) {
        const int cols = min(MF_CAP, w.Ktot - c0);
        const int n16 = (cols + 15) >> 4;
        bf16_t* stage = stage2 + (MM_DBUF ? ci : 0) * MF_STAGE;
        if (!MM_DBUF && c0) {                                             // single buffer: this chunk's rows start now, under the scatter
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          mm_stage_rows<MF_HALF>(w, c0, n16, lane, nh64, stage);
        }
        __builtin_amdgcn_wave_barrier();
        {   // clear the n16 * 16 rows this chunk uses: 1 KB per step
          const uint4 z = make_uint4(0, 0, 0, 0);
          for (int i = 0; i < n16; ++i) *(uint4*)(cimg + i * 512 + lane * 8) = z;
        }
        __builtin_amdgcn_wave_barrier();
        // scatter: corner (p, c) of this lane's query -> crow[r - c0] += weight
        if (!(MM_DIAG & 2)) {
#pragma unroll
          for (int p = 0; p < 8; ++p) {
            // the packed state is laundered per chunk iteration: otherwise the unpacked form of all 8 points (32 weights, 32 column
            // indices, 32 predicates) is hoisted out of the chunk loop as loop-invariant and the kernel spills
            uint32_t k_ = (uint32_t)tp.pk[p], wt_ = __builtin_bit_cast(uint32_t, tp.wt[p]), wb_ = __builtin_bit_cast(uint32_t, tp.wb[p]);
            asm volatile("" : "+v"(k_), "+v"(wt_), "+v"(wb_));
            const int k = (int)k_;
            const mm_f16x2 wt2 = __builtin_bit_cast(mm_f16x2, wt_), wb2 = __builtin_bit_cast(mm_f16x2, wb_);
            const int r00 = (k & 0x3fffffff) - c0, dx = (k >> 30) & 1;
            const int r10 = r00 + ((k >> 31) & bw_l);
            const float w00 = (float)wt2[0], w01 = (float)wt2[1], w10 = (float)wb2[0], w11 = (float)wb2[1];
            // masked corners and corners of another chunk go to one of the four dump rows behind the image (a clamped corner may alias
            // its live neighbour: it must not be written)
            const bool o00 = (unsigned)r00 < (unsigned)MF_CAP && w00 != 0.f, o01 = (unsigned)(r00 + dx) < (unsigned)MF_CAP && w01 != 0.f;
            const bool o10 = (unsigned)r10 < (unsigned)MF_CAP && w10 != 0.f, o11 = (unsigned)(r10 + dx) < (unsigned)MF_CAP && w11 != 0.f;
            if (__builtin_amdgcn_ballot_w64(o00 || o01 || o10 || o11) == 0) continue;      // no lane's point p reaches this chunk (uniform)
            const int i00 = (o00 ? r00 : MF_CAP) * 32, i01 = (o01 ? r00 + dx : MF_CAP + 1) * 32;
            const int i10 = (o10 ? r10 : MF_CAP + 2) * 32, i11 = (o11 ? r10 + dx : MF_CAP + 3) * 32;
            const float v00 = bf2f(crow[i00]) + w00, v01 = bf2f(crow[i01]) + w01, v10 = bf2f(crow[i10]) + w10, v11 = bf2f(crow[i11]) + w11;
            crow[i00] = mm_bf(v00); crow[i01] = mm_bf(v01); crow[i10] = mm_bf(v10); crow[i11] = mm_bf(v11);
          }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // this chunk's LDS-DMA has landed; earlier operand reads are done
        if (MM_DBUF && c0 + MF_CAP < w.Ktot)                              // next chunk -> the other buffer, under this chunk's MFMAs
          mm_stage_rows<MF_HALF>(w, c0 + MF_CAP, (min(MF_CAP, w.Ktot - c0 - MF_CAP) + 15) >> 4, lane, nh64, stage2 + (ci ^ 1) * MF_STAGE);
        __builtin_amdgcn_wave_barrier();
        if (!(MM_DIAG & 4)) {
#pragma unroll 1
          for (int ks = 0; ks < n16; ++ks) {
            const bf16_t* pa = cimg + (ks * 16 + tr_row) * 32 + tr_col;
            const mm_bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(MM_LDS_PTR(mm_bf16x4, pa));
            const mm_bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(MM_LDS_PTR(mm_bf16x4, pa + 4 * 32));
            const mm_bf16x8 A = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              const bf16_t* p = stage + half * MF_HALF + (ks * 16 + tr_row) * 32 + tr_col;
              const mm_bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(MM_LDS_PTR(mm_bf16x4, p));
              const mm_bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(MM_LDS_PTR(mm_bf16x4, p + 4 * 32));
              const mm_bf16x8 Bv = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
              if (half == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bv, acc0, 0, 0, 0);
              else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bv, acc1, 0, 0, 0);
            }
          }
        }
      }
    }
    // C/D layout: column = lane & 31 (channel of the half), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (query of the tile).  The tile
    // goes through LDS ([32 queries][64 channels] bf16, 144-byte rows, in the stage's space) and leaves as 16-byte pieces:
    // 8 lanes per 128-byte output row
    if (!(MM_DIAG & 8)) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * hv;
        stage2[m * 72 + (lane & 31)] = mm_bf(acc0[r]);
        stage2[m * 72 + 32 + (lane & 31)] = mm_bf(acc1[r]);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = i * 8 + row8;
        const int qm = __shfl(qq, m, 64);
        const uint4 v = *(const uint4*)(stage2 + m * 72 + sub8 * 8);
        if (tile * 32 + m < a.Nq) *(uint4*)(a.out + ((long)b * a.qpitch + qm) * nh64 + head * 64 + sub8 * 8) = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ d_loc / d_attw (raw gradients)
// Backward of the sampling w.r.t. locations and attention weights, emitted as the gradient of the RAW projection outputs (mmcv's
// view / normaliser / softmax backward folded in, as msda_bwd_lw_k<.., EMIT>).  Same decomposition as the forward: for a (tile, head,
// level pair, chunk of window rows) the four <gradient row, value row> dot products every sampling point needs are entries of
//
//     S^T[r, q] = sum_ch V[r, ch] * G[q, ch]         r: window rows of the chunk, q: the 32 queries, ch: 64 channels
//
// 4 MFMAs per 32 window rows with the gradient rows as B fragments straight from global memory (a lane's 8 consecutive channels of
// its query: no LDS, loaded once per tile) and the value rows as A fragments (ds_read_b128 from the LDS-DMA stage).  S^T goes to LDS
// as [r][32 queries] fp32: the lane that owns a point reads S^T[r_corner][its q] — always its own bank — and accumulates the three
// sums (weight, d/dx, d/dy) of its 8 points over the chunks.
struct MmBwdArgs {
  MmArgs f;                                   // forward inputs (out / loc_out / attw_out unused)
  const bf16_t* gout;                         // (B, Nq, nH*64)
  bf16_t* d_off; long d_off_ld; bf16_t* d_logit; long d_logit_ld;
  int4* bbox;                                 // optional: tap boxes (two point groups) of every (image, head, level, tile) for the d_value kernel below
};
#define MM_SIMG ((MM_CAP + 1) * 32)           // fp32 elements: [window row][32 queries] + one dump row

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MM_WAVES, MM_WAVES))) msda_mm_bwd_lw_k(MmBwdArgs ba) {
  __shared__ __attribute__((aligned(16))) float simg[MM_SIMG];
  __shared__ __attribute__((aligned(16))) bf16_t stage2[(1 + MM_DBUF) * MM_STAGE];
  const MmArgs& a = ba.f;
  const int nh64 = a.nH * 64;
  const long total = (long)a.B * a.nH * a.ntiles;
  const int xcd = blockIdx.x % MSDA_XCDS, jx = blockIdx.x / MSDA_XCDS, nx = gridDim.x / MSDA_XCDS;
  const long lo = total * xcd / MSDA_XCDS, hi = total * (xcd + 1) / MSDA_XCDS;
  for (long item = lo + jx; item < hi; item += nx) {
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));                                        // see msda_mm_fwd_k
    const int q = lane & 31, hv = lane >> 5;
    const int tile = (int)(item % a.ntiles);
    const int bh = (int)(item / a.ntiles);
    const int head = bh % a.nH, b = bh / a.nH;
    const int qi = tile * 32 + q;
    const bool qok = qi < a.Nq;
    const int qsafe = qok ? qi : tile * 32;
    const int qq = a.order ? a.order[qsafe] : qsafe;
    const long row = (long)b * a.qpitch + qq;
    const bf16_t* op = a.off + row * a.off_ld + head * 64 + (2 * hv) * 16;
    const float* rp = a.ref + (long)b * a.ref_sb + (long)qq * a.ref_sq + (long)(2 * hv) * a.ref_sl;
    uint4 o0 = *(const uint4*)op, o1 = *(const uint4*)(op + 8);
    float rx = rp[0], ry = rp[1];
    // gradient row of (query, head) as the four B fragments: channels ks * 16 + hv * 8 .. + 7
    mm_bf16x8 G[4];
    {
      const bf16_t* gp = ba.gout + row * nh64 + head * 64 + hv * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) G[ks] = qok ? *(const mm_bf16x8*)(gp + ks * 16) : mm_bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    float aw[2][8];
    mm_softmax(a, row, head, hv, qok, aw);
    float dsum = 0.f;                                                     // sum over this lane's 16 points of attw * d_attw
    float dav[2][8];
    const bf16_t* vb = a.value + ((long)b * a.Nv * a.nH + head) * 64;
#pragma unroll 1
    for (int s = 0; s < 2; ++s) {
      const int lvl = s + 2 * hv;
      MmWin w;
      w.WA = a.lv.W[s]; w.WB = a.lv.W[s + 2];
      const int HA = a.lv.H[s], HB = a.lv.H[s + 2];
      const int Wl = hv ? w.WB : w.WA, Hl = hv ? HB : HA;
      int pk[8];
      float fx[8], fy[8], awl[8];
      int bmin, bmax;
      int gmin0 = 0, gmax0 = 0, gmin1 = 0, gmax1 = 0;                      // boxes of the two point groups (0-3, 4-7)
      {
        const uint32_t u[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
        const float fW = (float)Wl, fH = (float)Hl;
        const float rW = 1.f / fW, rH = 1.f / fH;
        int xmn = 32767, ymn = 32767, xmx = -32768, ymx = -32768;
        int xmn1 = 32767, ymn1 = 32767, xmx1 = -32768, ymx1 = -32768;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          awl[p] = s ? aw[1][p] : aw[0][p];
          const float ox = __uint_as_float(u[p] << 16), oy = __uint_as_float(u[p] & 0xffff0000u);
          const float lx = rx + mm_div(ox, fW, rW), ly = ry + mm_div(oy, fH, rH);     // the forward's arithmetic to the bit
          const float x = lx * fW - 0.5f, y = ly * fH - 0.5f;
          const bool in = qok && y > -1.f && x > -1.f && y < fH && x < fW;
          const float xc = fminf(fmaxf(x, -1.f), fW), yc = fminf(fmaxf(y, -1.f), fH);
          const float xf = floorf(xc), yf = floorf(yc);
          const int x0 = (int)xf, y0 = (int)yf;
          fx[p] = xc - xf; fy[p] = yc - yf;
          const int xa = min(max(x0, 0), Wl - 1), xb = min(max(x0 + 1, 0), Wl - 1);
          const int ya = min(max(y0, 0), Hl - 1), yb = min(max(y0 + 1, 0), Hl - 1);
          // corner masks 00, 01, 10, 11 in bits 26-29 of the second word (kept in the sign bits of fx / fy would cost precision)
          const int m = (in && y0 >= 0 && x0 >= 0 ? 1 : 0) | (in && y0 >= 0 && x0 + 1 < Wl ? 2 : 0) | (in && y0 + 1 < Hl && x0 >= 0 ? 4 : 0) |
                        (in && y0 + 1 < Hl && x0 + 1 < Wl ? 8 : 0);
          pk[p] = xa | (ya << 13) | (m << 26) | ((xb - xa) << 30) | ((yb - ya) << 31);     // xa, ya < 2^13 (launcher)
          if (in) { xmn = min(xmn, xa); xmx = max(xmx, xb); ymn = min(ymn, ya); ymx = max(ymx, yb); }
          if (p == 3) { gmin0 = (ymn << 16) | (xmn & 0xffff); gmax0 = (ymx << 16) | (xmx & 0xffff); }   // box of points 0-3 (d_value kernel)
          if (p >= 4 && in) { xmn1 = min(xmn1, xa); xmx1 = max(xmx1, xb); ymn1 = min(ymn1, ya); ymx1 = max(ymx1, yb); }
        }
        bmin = (ymn << 16) | (xmn & 0xffff);
        bmax = (ymx << 16) | (xmx & 0xffff);
        gmin1 = (ymn1 << 16) | (xmn1 & 0xffff); gmax1 = (ymx1 << 16) | (xmx1 & 0xffff);
        if (s == 0) {
          o0 = *(const uint4*)(op + 16); o1 = *(const uint4*)(op + 24);
          rx = rp[a.ref_sl]; ry = rp[a.ref_sl + 1];
        }
      }
      bmin = mm_half_reduce<false>(bmin); bmax = mm_half_reduce<true>(bmax);
      const int mnA = __builtin_amdgcn_readlane(bmin, 31), mxA = __builtin_amdgcn_readlane(bmax, 31);
      const int mnB = __builtin_amdgcn_readlane(bmin, 63), mxB = __builtin_amdgcn_readlane(bmax, 63);
      w.xminA = (short)(mnA & 0xffff); w.yminA = mnA >> 16;
      w.xminB = (short)(mnB & 0xffff); w.yminB = mnB >> 16;
      const int xmaxA = (short)(mxA & 0xffff), ymaxA = mxA >> 16, xmaxB = (short)(mxB & 0xffff), ymaxB = mxB >> 16;
      const bool anyA = xmaxA >= w.xminA && ymaxA >= w.yminA, anyB = xmaxB >= w.xminB && ymaxB >= w.yminB;
      w.bwA = anyA ? xmaxA - w.xminA + 1 : 0; w.bwB = anyB ? xmaxB - w.xminB + 1 : 0;
      w.KA = anyA ? w.bwA * (ymaxA - w.yminA + 1) : 0; w.KB = anyB ? w.bwB * (ymaxB - w.yminB + 1) : 0;
      w.Ktot = w.KA + w.KB;
      if (ba.bbox) {                              // uniform.  Tap boxes of the point groups 0-3 / 4-7 of every (image, head, level, tile):
        // (min y << 16 | min x, max y << 16 | max x) x 2, INT16 extremes when no tap of the group is inside the map
        gmin0 = mm_half_reduce<false>(gmin0); gmax0 = mm_half_reduce<true>(gmax0);
        gmin1 = mm_half_reduce<false>(gmin1); gmax1 = mm_half_reduce<true>(gmax1);
        if ((lane & 31) == 31)
          ba.bbox[((long)bh * 4 + s + 2 * hv) * a.ntiles + tile] = make_int4(gmin0, gmax0, gmin1, gmax1);
      }
      float sv[8], sx[8], sy[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) { sv[p] = 0.f; sx[p] = 0.f; sy[p] = 0.f; }
      if (w.Ktot > 0) {                                                   // uniform
        w.ibwA = 1.f / (float)max(w.bwA, 1); w.ibwB = 1.f / (float)max(w.bwB, 1);
        w.startA = a.lv.start[s]; w.startB = a.lv.start[s + 2]; w.vb = vb;
        mm_stage_rows<MM_HALF>(w, 0, (min(MM_CAP, w.Ktot) + 15) >> 4, lane, nh64, stage2);
        const int bw_l = hv ? w.bwB : w.bwA;
        {
          const int xmin = hv ? w.xminB : w.xminA, ymin = hv ? w.yminB : w.yminA, kofs = hv ? w.KA : 0;
#pragma unroll
          for (int p = 0; p < 8; ++p) {
            const int k = pk[p];
            const bool live = (k & (15 << 26)) != 0;
            const int r00 = mul24(((k >> 13) & 0x1fff) - ymin, bw_l) + ((k & 0x1fff) - xmin) + kofs;
            pk[p] = live ? ((k & 0xfc000000) | r00) : 0;                  // r00 < 2^26 (launcher: Nv < 2^26)
          }
        }
#pragma unroll 1
        for (int c0 = 0, ci = 0; c0 < w.Ktot; c0 += MM_CAP, ci ^= 1) {
          const int cols = min(MM_CAP, w.Ktot - c0);
          const int n16 = (cols + 15) >> 4;
          bf16_t* stage = stage2 + (MM_DBUF ? ci : 0) * MM_STAGE;
          if (!MM_DBUF && c0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mm_stage_rows<MM_HALF>(w, c0, n16, lane, nh64, stage);
          }
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // this chunk has landed; the previous chunk's reads are done
          if (MM_DBUF && c0 + MM_CAP < w.Ktot)
            mm_stage_rows<MM_HALF>(w, c0 + MM_CAP, (min(MM_CAP, w.Ktot - c0 - MM_CAP) + 15) >> 4, lane, nh64, stage2 + (ci ^ 1) * MM_STAGE);
          __builtin_amdgcn_wave_barrier();
#pragma unroll 1
          for (int mb = 0; mb < (n16 + 1) >> 1; ++mb) {
            mm_f32x16 acc = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
              const mm_bf16x8 A = *(const mm_bf16x8*)(stage + (ks >> 1) * MM_HALF + (mb * 32 + q) * 32 + (ks & 1) * 16 + hv * 8);
              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, G[ks], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) simg[(mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hv) * 32 + q] = acc[r];
          }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int p = 0; p < 8; ++p) {
            uint32_t k_ = (uint32_t)pk[p];
            asm volatile("" : "+v"(k_));                                  // keep the unpacked form out of the chunk loop's preheader
            const int k = (int)k_;
            const int r00 = (k & 0x03ffffff) - c0, dx = (k >> 30) & 1;
            const int r10 = r00 + ((k >> 31) & bw_l);
            const bool o00 = (unsigned)r00 < (unsigned)MM_CAP && (k & (1 << 26)), o01 = (unsigned)(r00 + dx) < (unsigned)MM_CAP && (k & (2 << 26));
            const bool o10 = (unsigned)r10 < (unsigned)MM_CAP && (k & (4 << 26)), o11 = (unsigned)(r10 + dx) < (unsigned)MM_CAP && (k & (8 << 26));
            if (__builtin_amdgcn_ballot_w64(o00 || o01 || o10 || o11) == 0) continue;
            float d00 = simg[(o00 ? r00 : MM_CAP) * 32 + q], d01 = simg[(o01 ? r00 + dx : MM_CAP) * 32 + q];
            float d10 = simg[(o10 ? r10 : MM_CAP) * 32 + q], d11 = simg[(o11 ? r10 + dx : MM_CAP) * 32 + q];
            d00 = o00 ? d00 : 0.f; d01 = o01 ? d01 : 0.f; d10 = o10 ? d10 : 0.f; d11 = o11 ? d11 : 0.f;
            const float ax = fx[p], ay = fy[p], bx = 1.f - ax, by = 1.f - ay;
            sv[p] += by * (bx * d00 + ax * d01) + ay * (bx * d10 + ax * d11);
            sx[p] += by * (d01 - d00) + ay * (d11 - d10);
            sy[p] += bx * (d10 - d00) + ax * (d11 - d01);
          }
        }
      }
      // d_off = d_loc / (W, H) with d_loc = s * (weight * (W, H)): the level's 8 (x, y) pairs leave as one 32-byte run
      {
        uint32_t pr[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          pr[p] = (uint32_t)mm_bf(sx[p] * awl[p]) | ((uint32_t)mm_bf(sy[p] * awl[p]) << 16);
          dsum += awl[p] * sv[p];
          if (s == 0) dav[0][p] = sv[p]; else dav[1][p] = sv[p];
        }
        if (qok) {
          bf16_t* dp = ba.d_off + row * ba.d_off_ld + head * 64 + lvl * 16;
          *(uint4*)dp = make_uint4(pr[0], pr[1], pr[2], pr[3]);
          *(uint4*)(dp + 8) = make_uint4(pr[4], pr[5], pr[6], pr[7]);
        }
      }
    }
    // softmax backward over the 32 points of (query, head): d_logit = attw * (d_attw - sum attw * d_attw)
    dsum += __shfl_xor(dsum, 32, 64);
    if (qok) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        uint32_t pr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          pr[i] = (uint32_t)mm_bf(aw[s][2 * i] * (dav[s][2 * i] - dsum)) | ((uint32_t)mm_bf(aw[s][2 * i + 1] * (dav[s][2 * i + 1] - dsum)) << 16);
        *(uint4*)(ba.d_logit + row * ba.d_logit_ld + head * 32 + (s + 2 * hv) * 8) = make_uint4(pr[0], pr[1], pr[2], pr[3]);
      }
    }
  }
}

int msda_mm_supported(int B, int Nq, int Nv, int nH, int L, int P, int dtype, const MsdaLevels& lv) {
  if (dtype != GE_BF16 || L != 4 || P != 8 || nH < 1) return 0;
  for (int l = 0; l < L; ++l) if (lv.W[l] > 8191 || lv.H[l] > 8191) return 0;      // 13-bit tap coordinates in the packed point state
  if (Nv >= (1 << 26)) return 0;
  if ((long)Nv * nH * 64 >= (1L << 31) || (long)B * Nq * nH * L * P >= (1L << 31)) return 0;
  return 1;
}

int msda_mm_fwd_launch(const MmArgs& a0, hipStream_t s) {
  MmArgs a = a0;
  a.ntiles = (a.Nq + 31) / 32;
  const long total = (long)a.B * a.nH * a.ntiles;
  if (total <= 0) return GE_OK;
  // persistent single-wave workgroups: exactly what is resident at once (a second round of workgroups would start when the first ends)
  static int per_cu = 0, n_cu = 0;
  if (!per_cu) {
    int dev = 0, v = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return GE_ERR_UNSUPPORTED;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, msda_mm_fwd_k<false>, 64, 0) != hipSuccess || v < 1) v = 8;
    n_cu = pr.multiProcessorCount; per_cu = v;
  }
  long blocks = std::min(total, (long)n_cu * per_cu);
  blocks = std::max(blocks / MSDA_XCDS * MSDA_XCDS, (long)MSDA_XCDS);
  if (a.loc_out) msda_mm_fwd_k<true><<<(unsigned)blocks, 64, 0, s>>>(a);
  else msda_mm_fwd_k<false><<<(unsigned)blocks, 64, 0, s>>>(a);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

int msda_mm_bwd_lw_launch(const MmBwdArgs& b0, hipStream_t s) {
  MmBwdArgs ba = b0;
  ba.f.ntiles = (ba.f.Nq + 31) / 32;
  const long total = (long)ba.f.B * ba.f.nH * ba.f.ntiles;
  if (total <= 0) return GE_OK;
  static int per_cu = 0, n_cu = 0;
  if (!per_cu) {
    int dev = 0, v = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return GE_ERR_UNSUPPORTED;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, msda_mm_bwd_lw_k, 64, 0) != hipSuccess || v < 1) v = 8;
    n_cu = pr.multiProcessorCount; per_cu = v;
  }
  long blocks = std::min(total, (long)n_cu * per_cu);
  blocks = std::max(blocks / MSDA_XCDS * MSDA_XCDS, (long)MSDA_XCDS);
  msda_mm_bwd_lw_k<<<(unsigned)blocks, 64, 0, s>>>(ba);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

extern "C" int ge_msda_mm_supported(const int* spatial_hw, int B, int Nv, int Nq, int nH, int L, int P, int dtype) {
  MsdaLevels lv;
  if (!spatial_hw || msda_levels(spatial_hw, L, Nv, lv)) return 0;
  return msda_mm_supported(B, Nq, Nv, nH, L, P, dtype, lv);
}

static int fwd_mm_impl(const void* value, const int* spatial_hw, const void* off_raw, long off_ld, const void* logit_raw,
                       long logit_ld, const float* ref, long ref_sb, long ref_sq, long ref_sl, const int* order, float* loc,
                       float* attw, void* out, int B, int Nv, int Nq, int q_pitch, int nH, int L, int P, int dtype, void* stream) {
  if (!value || !spatial_hw || !off_raw || !logit_raw || !ref || !out || B < 0 || Nq < 0 || q_pitch < Nq || (loc == nullptr) != (attw == nullptr)) return GE_ERR_BAD_ARG;
  MsdaLevels lv;
  int e = msda_levels(spatial_hw, L, Nv, lv);
  if (e) return e;
  if (!msda_mm_supported(B, Nq, Nv, nH, L, P, dtype, lv)) return GE_ERR_UNSUPPORTED;
  if ((((uintptr_t)off_raw | (uintptr_t)logit_raw) & 15) || off_ld % 8 || logit_ld % 8) return GE_ERR_BAD_ARG;
  MmArgs a;
  a.value = (const bf16_t*)value; a.lv = lv;
  a.off = (const bf16_t*)off_raw; a.off_ld = off_ld; a.logit = (const bf16_t*)logit_raw; a.logit_ld = logit_ld;
  a.ref = ref; a.ref_sb = ref_sb; a.ref_sq = ref_sq; a.ref_sl = ref_sl; a.order = order;
  a.out = (bf16_t*)out; a.loc_out = loc; a.attw_out = attw;
  a.B = B; a.Nv = Nv; a.Nq = Nq; a.nH = nH; a.ntiles = 0; a.qpitch = q_pitch;
  return msda_mm_fwd_launch(a, ge_stream(stream));
}
extern "C" int ge_msda_fwd_mm(const void* value, const int* spatial_hw, const void* off_raw, long off_ld, const void* logit_raw,
                              long logit_ld, const float* ref, long ref_sb, long ref_sq, long ref_sl, const int* order, float* loc,
                              float* attw, void* out, int B, int Nv, int Nq, int nH, int L, int P, int dtype, void* stream) {
  return fwd_mm_impl(value, spatial_hw, off_raw, off_ld, logit_raw, logit_ld, ref, ref_sb, ref_sq, ref_sl, order, loc, attw, out, B, Nv, Nq, Nq, nH, L, P, dtype, stream);
}
// The first Nq queries of every image only: raw / out are (B, q_pitch, ...) row matrices with q_pitch >= Nq rows per image (the rest is left
// alone; `order` permutes 0 .. Nq - 1).  Round 6: the self-attention's level-0 queries (75 % of them) take the MFMA kernels, the coarse-level
// queries — whose 32-query patches span 16 - 64 level-0 cells — stay on the LDS-window kernels (kernels.ms_deform_attn_self_split).
extern "C" int ge_msda_fwd_mm_part(const void* value, const int* spatial_hw, const void* off_raw, long off_ld, const void* logit_raw,
                                   long logit_ld, const float* ref, long ref_sb, long ref_sq, long ref_sl, const int* order, void* out, int B,
                                   int Nv, int Nq, int q_pitch, int nH, int L, int P, int dtype, void* stream) {
  return fwd_mm_impl(value, spatial_hw, off_raw, off_ld, logit_raw, logit_ld, ref, ref_sb, ref_sq, ref_sl, order, nullptr, nullptr, out, B, Nv, Nq, q_pitch, nH, L, P, dtype,
                     stream);
}

// d_off_raw / d_logit_raw (same layout and type as off_raw / logit_raw, fully written) from the gradient of the output; d_value is NOT
// produced here (ge_msda_bwd_value_* / the binned path).
static int bwd_lw_mm_impl(const void* value, const int* spatial_hw, const void* off_raw, long off_ld, const void* logit_raw,
                          long logit_ld, const float* ref, long ref_sb, long ref_sq, long ref_sl, const int* order,
                          const void* d_out, void* d_off_raw, long d_off_ld, void* d_logit_raw, long d_logit_ld, void* workspace,
                          int B, int Nv, int Nq, int q_pitch, int nH, int L, int P, int dtype, void* stream) {
  if (!value || !spatial_hw || !off_raw || !logit_raw || !ref || !d_out || !d_off_raw || !d_logit_raw || B < 0 || Nq < 0 || q_pitch < Nq) return GE_ERR_BAD_ARG;
  MsdaLevels lv;
  int e = msda_levels(spatial_hw, L, Nv, lv);
  if (e) return e;
  if (!msda_mm_supported(B, Nq, Nv, nH, L, P, dtype, lv)) return GE_ERR_UNSUPPORTED;
  if ((((uintptr_t)off_raw | (uintptr_t)logit_raw | (uintptr_t)d_off_raw | (uintptr_t)d_logit_raw | (uintptr_t)d_out) & 15) || off_ld % 8 || logit_ld % 8 ||
      d_off_ld % 8 || d_logit_ld % 8)
    return GE_ERR_BAD_ARG;
  MmBwdArgs ba;
  MmArgs& a = ba.f;
  a.value = (const bf16_t*)value; a.lv = lv;
  a.off = (const bf16_t*)off_raw; a.off_ld = off_ld; a.logit = (const bf16_t*)logit_raw; a.logit_ld = logit_ld;
  a.ref = ref; a.ref_sb = ref_sb; a.ref_sq = ref_sq; a.ref_sl = ref_sl; a.order = order;
  a.out = nullptr; a.loc_out = nullptr; a.attw_out = nullptr;
  a.B = B; a.Nv = Nv; a.Nq = Nq; a.nH = nH; a.ntiles = 0; a.qpitch = q_pitch;
  ba.gout = (const bf16_t*)d_out;
  ba.d_off = (bf16_t*)d_off_raw; ba.d_off_ld = d_off_ld; ba.d_logit = (bf16_t*)d_logit_raw; ba.d_logit_ld = d_logit_ld;
  ba.bbox = (int4*)workspace;                  // head of the ge_msda_bwd_mm_workspace layout (mv_ws_layout)
  return msda_mm_bwd_lw_launch(ba, ge_stream(stream));
}
extern "C" int ge_msda_bwd_lw_mm(const void* value, const int* spatial_hw, const void* off_raw, long off_ld, const void* logit_raw,
                                 long logit_ld, const float* ref, long ref_sb, long ref_sq, long ref_sl, const int* order,
                                 const void* d_out, void* d_off_raw, long d_off_ld, void* d_logit_raw, long d_logit_ld, void* workspace,
                                 int B, int Nv, int Nq, int nH, int L, int P, int dtype, void* stream) {
  return bwd_lw_mm_impl(value, spatial_hw, off_raw, off_ld, logit_raw, logit_ld, ref, ref_sb, ref_sq, ref_sl, order, d_out, d_off_raw, d_off_ld, d_logit_raw,
                        d_logit_ld, workspace, B, Nv, Nq, Nq, nH, L, P, dtype, stream);
}
// ge_msda_bwd_lw_mm for the first Nq queries of every image (see ge_msda_fwd_mm_part): d_out / d_off_raw / d_logit_raw rows beyond them are left alone.
extern "C" int ge_msda_bwd_lw_mm_part(const void* value, const int* spatial_hw, const void* off_raw, long off_ld, const void* logit_raw,
                                      long logit_ld, const float* ref, long ref_sb, long ref_sq, long ref_sl, const int* order,
                                      const void* d_out, void* d_off_raw, long d_off_ld, void* d_logit_raw, long d_logit_ld, int B, int Nv, int Nq,
                                      int q_pitch, int nH, int L, int P, int dtype, void* stream) {
  return bwd_lw_mm_impl(value, spatial_hw, off_raw, off_ld, logit_raw, logit_ld, ref, ref_sb, ref_sq, ref_sl, order, d_out, d_off_raw, d_off_ld, d_logit_raw,
                        d_logit_ld, nullptr, B, Nv, Nq, q_pitch, nH, L, P, dtype, stream);
}

// ------------------------------------------------------------------------------------------------ d_value (round 5)
// Gradient of the sampling w.r.t. the value rows as the TRANSPOSE of the forward contraction,
//
//     dV_win[r, ch] = sum_q C[q, r] * dO[q, ch]        r: rows of a value window, q: queries, ch: 64 channels of the head
//
// with C the same coefficient image as in msda_mm_fwd_k.  Replaces, for the cross-attention, the count -> scan -> fill -> drain record
// pipeline (ge_msda_bwd_value_raw: 2.3 GB of 8-byte records written and read, 36 GB of L2-level gradient-row gathers per launch).
//
// The catch is where the result goes.  fp32 row atomics top out at 5.1 G wave-instructions/s chip-wide whatever the address pattern
// (tools/ubench/hip/atomics_xcd.hip: XCD-private rows, L2-resident footprints, half rows — all the same; the limit is per CU, ~120
// cycles per instruction), so flushing the window of every 32-query tile would cost more than the record pipeline.  Queries are
// processed in the order of the cell of their reference point, so CONSECUTIVE tiles sample nearly the same window: a wave takes a RUN
// of consecutive tiles of one (image, head, level) whose union window has at most MV_CAP rows, keeps dV_win of the whole run in MFMA
// accumulators (MV_CAP / 32 row blocks x 64 channels), and flushes once per run, skipping rows nothing was added to.  Runs are cut
// greedily (msda_mm_runs_k) from the per-tile tap boxes that msda_mm_bwd_lw_k leaves in the workspace; a single tile whose window
// exceeds MV_CAP rows forms its own run and is walked in chunks of MV_CAP rows (correct for any geometry, fast when the order gives
// locality).
//
//   window = TWO boxes concatenated along the row axis: the taps of points 0-3 and of points 4-7 of the level.  mmcv initialises the
//            offsets of point p as (p + 1) x the head's direction, so the 8 points of a diagonal head sweep a 9 x 9 box of which they
//            touch ~20 rows; the two half sweeps are 5 x 5 each.  (Rows that appear in both boxes are simply added twice by the flush.)
//   wave   = one run; lane = (query q = lane & 31, group g = lane >> 5): the 4 points of its group, scattered into ITS box's rows of
//            column q of the image [window row][32 queries] (bf16, 80-byte rows) — the two lanes of a query never meet
//   A      = 8 consecutive queries of a window row: plain 16-byte LDS reads; B = the tile's 32 gradient rows, staged by LDS-DMA as
//            [half][query][32 channels] and read through ds_read_b64_tr_b16; 4 MFMA 32x32x16 per 32 window rows and tile
//   loads  = two tiles ahead: the order entry of tile i + 2 and the raw projections of tile i + 1 are issued together with the
//            gradient-row DMA of tile i, before the tap arithmetic of tile i
// Numerics: as the forward (f16 corner products accumulated in a bf16 image, fp32 sums); the record path rounded weights to bf16 and
// the bilinear fractions to 8 bits.
#ifndef MV_CAP
#define MV_CAP 96                          // window rows held in accumulators (multiple of 32)
#endif
#define MV_NB (MV_CAP / 32)
#define MV_RMAX 32                         // tiles per run at most (bounds the tail of the dynamic schedule)
#define MV_CROW 40                         // bf16 elements per image row: 32 queries + 16 bytes of pad (conflict-free ds_read_b128 over 16 rows)
#define MV_CIMG ((MV_CAP + 4) * MV_CROW)   // + 4 dump rows
#define MV_HALF (32 * 32 + 32)             // stage of one channel half: [32 queries][32 channels] + 64 B skew
#define MV_EMPTY_MIN ((32767 << 16) | 32767)
#define MV_EMPTY_MAX ((int)0x80008000)
#ifndef MV_DIAG
#define MV_DIAG 0                          // measurement aid: 1 no gradient-row DMA, 2 no scatter, 4 no MFMA, 8 no flush, 16 no image clear
#endif

struct MvWs { int4* bbox; int4* runs; int* ctrl; long runs_cap; };      // ctrl[x] = runs of XCD x, ctrl[8 + x] = its work cursor,
                                                                         // ctrl[16] = window rows flushed, ctrl[17] = tile passes (statistics), ctrl[18 + 2 l], ctrl[19 + 2 l] = the same of level l
static size_t mv_ws_layout(int B, int Nq, int nH, char* base, MvWs* ws, size_t* ctrl_off = nullptr) {
  const long ntiles = (Nq + 31) / 32, segs = (long)B * nH * 4;
  const long cap = (segs + MSDA_XCDS - 1) / MSDA_XCDS * ntiles;           // runs of one XCD's segments at most
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t o_bbox = take((size_t)segs * ntiles * sizeof(int4));
  const size_t o_runs = take((size_t)MSDA_XCDS * cap * 2 * sizeof(int4));
  const size_t o_ctrl = take(32 * sizeof(int));
  if (ctrl_off) *ctrl_off = o_ctrl;
  if (ws) { ws->bbox = (int4*)(base + o_bbox); ws->runs = (int4*)(base + o_runs); ws->ctrl = (int*)(base + o_ctrl); ws->runs_cap = cap; }
  return off;
}

struct MvBox { int x0, y0, x1, y1; };
__device__ __forceinline__ MvBox mv_unpack(int mn, int mx) { return MvBox{(short)(mn & 0xffff), mn >> 16, (short)(mx & 0xffff), mx >> 16}; }
__device__ __forceinline__ bool mv_some(const MvBox& b) { return b.x1 >= b.x0 && b.y1 >= b.y0; }
__device__ __forceinline__ int mv_rows(const MvBox& b) { return mv_some(b) ? (b.x1 - b.x0 + 1) * (b.y1 - b.y0 + 1) : 0; }
__device__ __forceinline__ MvBox mv_join(const MvBox& a, const MvBox& b) {
  if (!mv_some(b)) return a;
  if (!mv_some(a)) return b;
  return MvBox{min(a.x0, b.x0), min(a.y0, b.y0), max(a.x1, b.x1), max(a.y1, b.y1)};
}

// One wave per (image, head, level): walks the tiles in order and cuts them into runs whose two union boxes have at most MV_CAP rows
// together (a tile that alone exceeds it: a run of its own, chunked by the consumer).  Runs without any tap inside the map are dropped.
// Record (2 x int4): {segment = (image * nH + head) * 4 + level, first tile | tiles << 24, box 0: min y << 16 | min x, height << 16 | width},
//                    {box 1: min y << 16 | min x, height << 16 | width, -, -}; an empty box has width = height = 0.
__global__ void __launch_bounds__(64) msda_mm_runs_k(MvWs ws, int ntiles, int nsegs, int level_mask) {
  // runs are collected in LDS and appended to the XCD's list 64 at a time: a slot reservation per run is a global atomic WITH return, ~1 us
  // of round trip each — with one run per tile (the model's geometry) that was 3 080 serial round trips per wave: 1.27 ms for this kernel
  __shared__ int4 pend[2 * 64];
  const int seg = blockIdx.x, lane = threadIdx.x;
  const int xcd = (int)(((long)(seg >> 2) * MSDA_XCDS) / (nsegs >> 2));     // image-major: the mapping of the sampling kernels
  const int4* bb = ws.bbox + (long)seg * ntiles;
  int4* list = ws.runs + (long)xcd * ws.runs_cap * 2;
  const MvBox none{32767, 32767, -32768, -32768};
  const bool wanted = (level_mask >> (seg & 3)) & 1;                         // statistics for every level, work only for the requested ones
  int t0 = 0, nrows = 0, npass = 0, npend = 0;
  MvBox c0 = none, c1 = none;                                               // current run: union boxes of the two point groups (uniform)
  auto flush = [&]() {
    if (npend == 0) return;
    int slot = 0;
    if (lane == 0) slot = atomicAdd(ws.ctrl + xcd, npend);
    slot = __builtin_amdgcn_readfirstlane(slot);
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < 2 * npend; i += 64) list[2 * slot + i] = pend[i];
    __builtin_amdgcn_wave_barrier();
    npend = 0;
  };
  auto emit = [&](int t1) {
    const int k0 = mv_rows(c0), k1 = mv_rows(c1);
    if (k0 + k1 > 0 && t1 > t0) {
      nrows += k0 + k1;
      npass += (t1 - t0) * ((k0 + k1 + MV_CAP - 1) / MV_CAP);
      if (wanted) {
        if (lane == 0) {
          const bool s0 = k0 > 0, s1 = k1 > 0;
          pend[2 * npend] = make_int4(seg, t0 | ((t1 - t0) << 24), s0 ? (c0.y0 << 16) | (c0.x0 & 0xffff) : 0,
                                      s0 ? ((c0.y1 - c0.y0 + 1) << 16) | (c0.x1 - c0.x0 + 1) : 0);
          pend[2 * npend + 1] = make_int4(s1 ? (c1.y0 << 16) | (c1.x0 & 0xffff) : 0, s1 ? ((c1.y1 - c1.y0 + 1) << 16) | (c1.x1 - c1.x0 + 1) : 0, 0, 0);
        }
        if (++npend == 64) flush();
      }
    }
  };
  for (int base = 0; base < ntiles; base += 64) {
    const int4 v = base + lane < ntiles ? bb[base + lane] : make_int4(MV_EMPTY_MIN, MV_EMPTY_MAX, MV_EMPTY_MIN, MV_EMPTY_MAX);
    const int n = min(64, ntiles - base);
    for (int j = 0; j < n; ++j) {
      const MvBox b0 = mv_unpack(__builtin_amdgcn_readlane(v.x, j), __builtin_amdgcn_readlane(v.y, j));
      const MvBox b1 = mv_unpack(__builtin_amdgcn_readlane(v.z, j), __builtin_amdgcn_readlane(v.w, j));
      const MvBox n0 = mv_join(c0, b0), n1 = mv_join(c1, b1);
      const int t = base + j;
      if (t > t0 && (mv_rows(n0) + mv_rows(n1) > MV_CAP || t - t0 >= MV_RMAX)) {
        emit(t);
        t0 = t; c0 = mv_some(b0) ? b0 : none; c1 = mv_some(b1) ? b1 : none;
      } else { c0 = n0; c1 = n1; }
    }
  }
  emit(ntiles);
  flush();
  if (lane == 0 && nrows) {
    atomicAdd(ws.ctrl + 16, nrows); atomicAdd(ws.ctrl + 17, npass);
    atomicAdd(ws.ctrl + 18 + 2 * (seg & 3), nrows); atomicAdd(ws.ctrl + 19 + 2 * (seg & 3), npass);      // the same per level
  }
}

struct MvArgs {
  MmArgs f;                                   // forward inputs (value / out / loc_out / attw_out unused)
  const bf16_t* gout;                         // (B, Nq, nH*64)
  float* d_value;                             // (B, Nv, nH, 64) f32, accumulated into
  MvWs ws;
};

// raw projections of one lane's 4 points of a tile: offsets of the run's level (16 B), logits of all four levels (4 x 8 B), reference point
struct MvRaw { uint4 o; uint2 l0, l1, l2, l3; float rx, ry; };
__device__ __forceinline__ MvRaw mv_load_raw(const MmArgs& a, int b, int head, int lvl, int g, int qq) {
  const long row = (long)b * a.Nq + qq;
  MvRaw r;
  r.o = *(const uint4*)(a.off + row * a.off_ld + head * 64 + lvl * 16 + g * 8);
  const bf16_t* lp = a.logit + row * a.logit_ld + head * 32 + g * 4;
  r.l0 = *(const uint2*)lp; r.l1 = *(const uint2*)(lp + 8); r.l2 = *(const uint2*)(lp + 16); r.l3 = *(const uint2*)(lp + 24);
  const float* rp = a.ref + (long)b * a.ref_sb + (long)qq * a.ref_sq + (long)lvl * a.ref_sl;
  r.rx = rp[0]; r.ry = rp[1];
  return r;
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MM_WAVES, MM_WAVES))) msda_mm_bwd_v_k(MvArgs va) {
  __shared__ __attribute__((aligned(16))) bf16_t cimg[MV_CIMG];
  __shared__ __attribute__((aligned(16))) bf16_t stage[2 * MV_HALF];
  const MmArgs& a = va.f;
  const int nh64 = a.nH * 64;
  const int xcd = blockIdx.x % MSDA_XCDS;
  const int nrun = va.ws.ctrl[xcd];
  const int4* list = va.ws.runs + (long)xcd * va.ws.runs_cap * 2;
  int* cursor = va.ws.ctrl + MSDA_XCDS + xcd;
  for (;;) {
    int it = 0;
    if (threadIdx.x == 0) it = atomicAdd(cursor, 1);
    it = __builtin_amdgcn_readfirstlane(it);
    if (it >= nrun) break;
    const int4 rec = list[2 * it], rec1 = list[2 * it + 1];
    const int seg = __builtin_amdgcn_readfirstlane(rec.x), tt = __builtin_amdgcn_readfirstlane(rec.y);
    const int min0 = __builtin_amdgcn_readfirstlane(rec.z), dim0 = __builtin_amdgcn_readfirstlane(rec.w);
    const int min1 = __builtin_amdgcn_readfirstlane(rec1.x), dim1 = __builtin_amdgcn_readfirstlane(rec1.y);
    const int lvl = seg & 3, bh = seg >> 2, head = bh % a.nH, b = bh / a.nH;
    const int t0 = tt & 0xffffff, nt = (int)((unsigned)tt >> 24);
    const int K0 = (dim0 & 0xffff) * (dim0 >> 16), Ks = K0 + (dim1 & 0xffff) * (dim1 >> 16);
    const int Wl = a.lv.W[lvl], Hl = a.lv.H[lvl], lstart = a.lv.start[lvl];
    const float fW = (float)Wl, fH = (float)Hl, rW = 1.f / fW, rH = 1.f / fH;
    int lane0 = threadIdx.x;
    asm volatile("" : "+v"(lane0));
    // this lane's group: box and first row in the concatenated window
    const int g = lane0 >> 5;
    const int gmin = g ? min1 : min0, gdim = g ? dim1 : dim0;
    const int xmin = (short)(gmin & 0xffff), ymin = gmin >> 16, bw = gdim & 0xffff, bhh = gdim >> 16, kofs = g ? K0 : 0;
#pragma unroll 1
    for (int c0 = 0; c0 < Ks; c0 += MV_CAP) {
      const int rows = min(MV_CAP, Ks - c0);
      const int nb = (rows + 31) >> 5;
      mm_f32x16 acc[MV_NB][2];
#pragma unroll
      for (int i = 0; i < MV_NB; ++i) { acc[i][0] = 0.f; acc[i][1] = 0.f; }
      // software pipeline over the tiles of the run: order entry two tiles ahead, raw projections one tile ahead
      const int qlast = a.Nq - 1;
      auto order_of = [&](int tile, int q) { const int qi = min(tile * 32 + q, qlast); return a.order ? a.order[qi] : qi; };
      int qq_cur = order_of(t0, lane0 & 31);
      int qq_nxt = order_of(min(t0 + 1, t0 + nt - 1), lane0 & 31);
      MvRaw raw_cur = mv_load_raw(a, b, head, lvl, g, qq_cur);
#pragma unroll 1
      for (int tile = t0; tile < t0 + nt; ++tile) {
        int lane = threadIdx.x;
        asm volatile("" : "+v"(lane));                                      // see msda_mm_fwd_k
        const int q = lane & 31;
        const int tr_row = (lane >> 5) * 8 + ((lane & 15) >> 2), tr_col = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
        const bool qok = tile * 32 + q < a.Nq;
        const int qq = qq_cur;
        const MvRaw rw = raw_cur;
        // the tile's 32 gradient rows -> LDS (2 x 16 rows x 2 channel halves); the previous tile's fragment reads have been consumed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int qr = __shfl(qq, j * 16 + (lane >> 2), 64);
          const bf16_t* src = va.gout + ((long)b * a.Nq + qr) * nh64 + head * 64 + (lane & 3) * 8;
          if (!(MV_DIAG & 1)) {
            __builtin_amdgcn_global_load_lds(src, MM_LDS_PTR(void, stage + j * 16 * 32), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(src + 32, MM_LDS_PTR(void, stage + MV_HALF + j * 16 * 32), 16, 0, 0);
          }
        }
        // in flight under this tile's arithmetic: the next tile's raw projections, the order entry of the tile after it
        const bool more = tile + 1 < t0 + nt;
        raw_cur = mv_load_raw(a, b, head, lvl, g, more ? qq_nxt : qq);
        qq_cur = qq_nxt;
        qq_nxt = order_of(min(tile + 2, t0 + nt - 1), q);
        float aw[4];
        {
          const uint32_t u[8] = {rw.l0.x, rw.l0.y, rw.l1.x, rw.l1.y, rw.l2.x, rw.l2.y, rw.l3.x, rw.l3.y};
          float e[16], m = -INFINITY;
#pragma unroll
          for (int i = 0; i < 8; ++i) { e[2 * i] = __uint_as_float(u[i] << 16); e[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u); }
#pragma unroll
          for (int i = 0; i < 16; ++i) m = fmaxf(m, e[i]);
          m = fmaxf(m, __shfl_xor(m, 32, 64));
          float sum = 0.f;
#pragma unroll
          for (int i = 0; i < 16; ++i) { e[i] = __expf(e[i] - m); sum += e[i]; }
          sum += __shfl_xor(sum, 32, 64);
          const float inv = qok ? 1.f / sum : 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) aw[i] = (lvl == 0 ? e[i] : lvl == 1 ? e[4 + i] : lvl == 2 ? e[8 + i] : e[12 + i]) * inv;
        }
        // taps (the arithmetic of mm_taps / msda_mm_bwd_lw_k to the bit: the run's boxes were computed from it)
        int pk[4];
        mm_f16x2 wt[4], wb[4];
        {
          const uint32_t u[4] = {rw.o.x, rw.o.y, rw.o.z, rw.o.w};
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const float ox = __uint_as_float(u[p] << 16), oy = __uint_as_float(u[p] & 0xffff0000u);
            const float lx = rw.rx + mm_div(ox, fW, rW), ly = rw.ry + mm_div(oy, fH, rH);
            const float x = lx * fW - 0.5f, y = ly * fH - 0.5f;
            const bool in = qok && y > -1.f && x > -1.f && y < fH && x < fW;
            const float xc = fminf(fmaxf(x, -1.f), fW), yc = fminf(fmaxf(y, -1.f), fH);
            const float xf = floorf(xc), yf = floorf(yc);
            const int x0 = (int)xf, y0 = (int)yf;
            const float ax = xc - xf, ay = yc - yf;
            const int xa = min(max(x0, 0), Wl - 1), xb = min(max(x0 + 1, 0), Wl - 1);
            const int ya = min(max(y0, 0), Hl - 1), yb = min(max(y0 + 1, 0), Hl - 1);
            // a tap outside the group's box cannot happen when the box comes from the same arithmetic; a foreign workspace must not corrupt memory
            const bool live = in && xa >= xmin && xb < xmin + bw && ya >= ymin && yb < ymin + bhh;
            const float wgt = live ? aw[p] : 0.f;
            const float wxa = x0 >= 0 ? (1.f - ax) * wgt : 0.f, wxb = x0 + 1 < Wl ? ax * wgt : 0.f;
            const float wya = y0 >= 0 ? 1.f - ay : 0.f, wyb = y0 + 1 < Hl ? ay : 0.f;
            wt[p] = __builtin_bit_cast(mm_f16x2, __builtin_amdgcn_cvt_pkrtz(wya * wxa, wya * wxb));
            wb[p] = __builtin_bit_cast(mm_f16x2, __builtin_amdgcn_cvt_pkrtz(wyb * wxa, wyb * wxb));
            const int r00 = mul24(ya - ymin, bw) + (xa - xmin) + kofs;
            pk[p] = live ? (r00 | ((xb - xa) << 30) | ((yb - ya) << 31)) : 0;       // r00 < 2^24 (two boxes of msda_levels' < 2^23 maps)
          }
        }
        // clear the row blocks in use (2.5 KB each), then drop the coefficients
        if (!(MV_DIAG & 16)) {
          const uint4 z = make_uint4(0, 0, 0, 0);
          for (int i = lane * 8; i < nb * 32 * MV_CROW; i += 512) *(uint4*)(cimg + i) = z;
        }
        __builtin_amdgcn_wave_barrier();
        bf16_t* ccol = cimg + q;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          if (MV_DIAG & 2) break;
          const int k = pk[p];
          const int r00 = (k & 0x3fffffff) - c0, dx = (k >> 30) & 1;
          const int r10 = r00 + ((k >> 31) & bw);
          const float w00 = (float)wt[p][0], w01 = (float)wt[p][1], w10 = (float)wb[p][0], w11 = (float)wb[p][1];
          const bool o00 = (unsigned)r00 < (unsigned)MV_CAP && w00 != 0.f, o01 = (unsigned)(r00 + dx) < (unsigned)MV_CAP && w01 != 0.f;
          const bool o10 = (unsigned)r10 < (unsigned)MV_CAP && w10 != 0.f, o11 = (unsigned)(r10 + dx) < (unsigned)MV_CAP && w11 != 0.f;
          if (__builtin_amdgcn_ballot_w64(o00 || o01 || o10 || o11) == 0) continue;
          const int i00 = (o00 ? r00 : MV_CAP) * MV_CROW, i01 = (o01 ? r00 + dx : MV_CAP + 1) * MV_CROW;
          const int i10 = (o10 ? r10 : MV_CAP + 2) * MV_CROW, i11 = (o11 ? r10 + dx : MV_CAP + 3) * MV_CROW;
          const float v00 = bf2f(ccol[i00]) + w00, v01 = bf2f(ccol[i01]) + w01, v10 = bf2f(ccol[i10]) + w10, v11 = bf2f(ccol[i11]) + w11;
          ccol[i00] = mm_bf(v00); ccol[i01] = mm_bf(v01); ccol[i10] = mm_bf(v10); ccol[i11] = mm_bf(v11);
        }
        // the two lanes of a query write different ROWS (their groups' boxes), but a masked corner of group 0 and one of group 1 share
        // the dump rows: harmless (never read)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");         // gradient rows have landed (and the prefetches), image complete
        __builtin_amdgcn_wave_barrier();
        mm_bf16x8 Bf[2][2];                                                 // queries ks * 16 .. + 15 of both channel halves
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const bf16_t* p = stage + half * MV_HALF + (ks * 16 + tr_row) * 32 + tr_col;
            const mm_bf16x4 u0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(MM_LDS_PTR(mm_bf16x4, p));
            const mm_bf16x4 u1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(MM_LDS_PTR(mm_bf16x4, p + 4 * 32));
            Bf[ks][half] = __builtin_shufflevector(u0, u1, 0, 1, 2, 3, 4, 5, 6, 7);
          }
#pragma unroll
        for (int rb = 0; rb < MV_NB; ++rb) {
          if (rb < nb && !(MV_DIAG & 4)) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              const mm_bf16x8 A = *(const mm_bf16x8*)(cimg + (rb * 32 + q) * MV_CROW + ks * 16 + (lane >> 5) * 8);
              acc[rb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bf[ks][0], acc[rb][0], 0, 0, 0);
              acc[rb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bf[ks][1], acc[rb][1], 0, 0, 0);
            }
          }
        }
      }
      // flush: C/D layout column = lane & 31 (channel of the half), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the block.  A row no
      // point touched holds exact zeros: its lanes sit the atomic out (the atomic units are the bound: ~2 cycles per active lane and CU)
      {
        const int lane = threadIdx.x;
        float* dvb = va.d_value + ((long)b * a.Nv * a.nH + head) * 64 + (lane & 31);
        const int bw0 = dim0 & 0xffff, bw1 = dim1 & 0xffff;
        const float ibw0 = 1.f / (float)max(bw0, 1), ibw1 = 1.f / (float)max(bw1, 1);
        const int x00 = (short)(min0 & 0xffff), y00 = min0 >> 16, x01 = (short)(min1 & 0xffff), y01 = min1 >> 16;
#pragma unroll
        for (int rb = 0; rb < MV_NB; ++rb) {
          if (rb < nb && !(MV_DIAG & 8)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int m = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
              const float v0 = acc[rb][0][r], v1 = acc[rb][1][r];
              if (m < rows && (v0 != 0.f || v1 != 0.f)) {
                const int wr = c0 + m;
                const bool g1 = wr >= K0;
                const int t = g1 ? wr - K0 : wr, bwg = g1 ? bw1 : bw0;
                const int wy = (int)(((float)t + 0.5f) * (g1 ? ibw1 : ibw0));
                const int wx = t - mul24(wy, bwg);
                const int pix = mul24((g1 ? y01 : y00) + wy, Wl) + (g1 ? x01 : x00) + wx + lstart;
                float* dst = dvb + (long)pix * nh64;
                if (v0 != 0.f) atomicAdd(dst, v0);
                if (v1 != 0.f) atomicAdd(dst + 32, v1);
              }
            }
          }
        }
      }
    }
  }
}

extern "C" size_t ge_msda_bwd_mm_workspace(int B, int Nq, int nH, int L) {
  if (B <= 0 || Nq <= 0 || nH <= 0 || L != 4) return 0;
  return mv_ws_layout(B, Nq, nH, nullptr, nullptr);
}

// Byte offset, inside the workspace, of the two ints the run cutter leaves behind: {window rows flushed, tile passes} of the latest
// ge_msda_bwd_value_mm on that workspace — what its cost is made of (~0.28 ns per row + ~2.1 ns per pass on MI355X), so that a caller can
// choose between it and the record pipeline, whose cost does not depend on the geometry.
extern "C" size_t ge_msda_bwd_mm_stats_offset(int B, int Nq, int nH, int L) {
  if (B <= 0 || Nq <= 0 || nH <= 0 || L != 4) return 0;
  size_t o_ctrl = 0;
  mv_ws_layout(B, Nq, nH, nullptr, nullptr, &o_ctrl);
  return o_ctrl + 16 * sizeof(int);
}

// d_value of ge_msda_fwd_mm (accumulated into the zero-filled f32 tensor) from the gradient of the output.  `workspace` must be the one
// ge_msda_bwd_lw_mm was given for the same inputs: its head holds the per-tile tap boxes that kernel leaves behind.
extern "C" int ge_msda_bwd_value_mm(const int* spatial_hw, const void* off_raw, long off_ld, const void* logit_raw, long logit_ld,
                                    const float* ref, long ref_sb, long ref_sq, long ref_sl, const int* order, const void* d_out,
                                    float* d_value, void* workspace, size_t workspace_bytes, int level_mask, int B, int Nv, int Nq, int nH,
                                    int L, int P, int dtype, void* stream) {
  if (!spatial_hw || !off_raw || !logit_raw || !ref || !d_out || !workspace || B < 0 || Nq < 0) return GE_ERR_BAD_ARG;
  MsdaLevels lv;
  int e = msda_levels(spatial_hw, L, Nv, lv);
  if (e) return e;
  if (!msda_mm_supported(B, Nq, Nv, nH, L, P, dtype, lv)) return GE_ERR_UNSUPPORTED;
  if ((((uintptr_t)off_raw | (uintptr_t)logit_raw | (uintptr_t)d_out) & 15) || off_ld % 8 || logit_ld % 8) return GE_ERR_BAD_ARG;
  if (B == 0 || Nq == 0) return GE_OK;
  MvArgs va;
  if (workspace_bytes < mv_ws_layout(B, Nq, nH, (char*)workspace, &va.ws)) return GE_ERR_BAD_ARG;
  MmArgs& a = va.f;
  a.value = nullptr; a.lv = lv;
  a.off = (const bf16_t*)off_raw; a.off_ld = off_ld; a.logit = (const bf16_t*)logit_raw; a.logit_ld = logit_ld;
  a.ref = ref; a.ref_sb = ref_sb; a.ref_sq = ref_sq; a.ref_sl = ref_sl; a.order = order;
  a.out = nullptr; a.loc_out = nullptr; a.attw_out = nullptr;
  a.B = B; a.Nv = Nv; a.Nq = Nq; a.nH = nH; a.ntiles = (Nq + 31) / 32; a.qpitch = Nq;
  va.gout = (const bf16_t*)d_out; va.d_value = d_value;
  hipStream_t s = ge_stream(stream);
  hipError_t he = hipMemsetAsync(va.ws.ctrl, 0, 32 * sizeof(int), s);
  if (he != hipSuccess) return (int)he;
  const int nsegs = B * nH * 4;
  msda_mm_runs_k<<<(unsigned)nsegs, 64, 0, s>>>(va.ws, a.ntiles, nsegs, d_value ? level_mask : 0);
  GE_LAUNCH_CHECK();
  if (!d_value || !(level_mask & 15)) return GE_OK;     // statistics only (ge_msda_bwd_mm_stats_offset)
  static int per_cu = 0, n_cu = 0;
  if (!per_cu) {
    int dev = 0, v = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return GE_ERR_UNSUPPORTED;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, msda_mm_bwd_v_k, 64, 0) != hipSuccess || v < 1) v = 8;
    n_cu = pr.multiProcessorCount; per_cu = v;
  }
  const long blocks = std::max((long)n_cu * per_cu / MSDA_XCDS * MSDA_XCDS, (long)MSDA_XCDS);
  msda_mm_bwd_v_k<<<(unsigned)blocks, 64, 0, s>>>(va);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// ------------------------------------------------------------------------------------------------ d_value, value-stationary (round 6)
// The same contraction as msda_mm_bwd_v_k, dV[r, ch] = sum_q C[q, r] dO[q, ch], with the OUTPUT held still instead of the queries:
//
//   work item  = (image, head, level, SUPER-BLOCK of 24 x 16 value positions, a chunk of <= VS_T query tiles that can reach it)
//   workgroup  = 4 waves; wave w keeps rows [96 w, 96 w + 96) x 64 channels of the super-block in MFMA accumulators for the WHOLE chunk
//   per tile   = lane (query q = tid >> 3, point p = tid & 7) does the tap arithmetic of ONE sampling point, drops its four coefficients
//                into the shared [384 rows][32 queries] bf16 image with ds_pk_add_bf16 (corners outside the super-block are skipped: the
//                neighbouring super-block's item takes them), the tile's 32 gradient rows arrive once by LDS-DMA; after ONE barrier every wave
//                multiplies its LIVE 32-row blocks (a 12-bit mask OR-ed over the workgroup) against the gradient tile: 4 MFMA per live block
//   result     = written once per item: plain stores when the super-block has a single chunk (the common case at level 0), fp32 atomics when
//                its tile list was cut into several chunks (coarse levels: one super-block collects thousands of tiles) — ~1.7 M row flushes
//                per launch instead of the 23 M of the query-stationary kernel, and no records.
//
// Why it can win where the transposed kernel lost: that one pays an atomic flush per (tile, window) — 5.1 G row-instructions/s chip-wide
// is the limit of the atomic units — this one pays a TILE VISIT per (tile, super-block overlap): 2.2 visits per tile and
// level at level 0 of the KITTI shape, 1.0 - 2.0 at the coarse levels (tools/ubench/msda_mm/geom_stats.py), each shared by four waves.
//
// msda_vs_index_k turns the per-tile tap boxes (left in the workspace by msda_mm_bwd_lw_k) into per-super-block tile lists: one workgroup per
// (image, head, level), LDS counting sort.  Tiles that would be listed under more than VS_MAXOV super-blocks (incoherent geometry), or all tiles
// of a segment whose lists overflow their slab, are handed to the atomic kernel msda_mm_bwd_v_k as single-tile runs instead: correct for any
// geometry, fast for coherent ones.
#define VS_SW 24
#define VS_SH 16
#define VS_ROWS (VS_SW * VS_SH)            // 384 rows = 4 waves x 3 MFMA row blocks
#define VS_CROW 40                         // bf16 elements per image row (32 queries + 16 B pad: conflict-free 16-byte reads over 16 rows)
#define VS_IMG (VS_ROWS * VS_CROW)
#define VS_MAXSB 1024                      // super-blocks of one level at most (LDS counters of the index kernel)
#define VS_MAXOV 12
#define VS_OVF 4                           // list slab of a segment: VS_OVF entries per tile
#ifndef VS_T
#define VS_T 256                           // tiles per work item at most
#endif
#ifndef VS_DIAG
#define VS_DIAG 0                          // measurement aid: 1 no gradient-row DMA, 2 no scatter, 4 no MFMA, 8 no flush
#endif

struct VsWs { int* lists; int4* items; int* ctrl; long list_cap; long items_cap; };    // ctrl[x] = items of XCD x, ctrl[8 + x] = its cursor,
                                                                                         // ctrl[16] = tile visits, ctrl[17] = stray tiles, ctrl[18] = items, ctrl[19] = multi-chunk items
static int vs_max_sb(const MsdaLevels& lv) {
  int m = 0;
  for (int l = 0; l < 4; ++l) m = std::max(m, ((lv.W[l] + VS_SW - 1) / VS_SW) * ((lv.H[l] + VS_SH - 1) / VS_SH));
  return m;
}
static size_t vs_ws_layout(int B, int Nq, int nH, const MsdaLevels& lv, char* base, MvWs* mv, VsWs* vs, size_t* vs_ctrl_off = nullptr) {
  size_t off = mv_ws_layout(B, Nq, nH, base, mv);
  const long ntiles = (Nq + 31) / 32, segs = (long)B * nH * 4;
  const long list_cap = VS_OVF * ntiles;
  const long items_cap = (segs + MSDA_XCDS - 1) / MSDA_XCDS * (vs_max_sb(lv) + 2 * list_cap / VS_T + 2);
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t o_lists = take((size_t)segs * list_cap * sizeof(int));
  const size_t o_items = take((size_t)MSDA_XCDS * items_cap * sizeof(int4));
  const size_t o_ctrl = take(32 * sizeof(int));
  if (vs_ctrl_off) *vs_ctrl_off = o_ctrl;
  if (vs) { vs->lists = (int*)(base + o_lists); vs->items = (int4*)(base + o_items); vs->ctrl = (int*)(base + o_ctrl); vs->list_cap = list_cap; vs->items_cap = items_cap; }
  return off;
}

__global__ void __launch_bounds__(256) msda_vs_index_k(MvWs mv, VsWs vs, MsdaLevels lv, int ntiles, int nsegs) {
  __shared__ int cnt[VS_MAXSB], off[VS_MAXSB + 1], fillp[VS_MAXSB];
  const int seg = blockIdx.x, tid = threadIdx.x;
  const int lvl = seg & 3;
  const int nsx = (lv.W[lvl] + VS_SW - 1) / VS_SW, nsy = (lv.H[lvl] + VS_SH - 1) / VS_SH, nsb = nsx * nsy;
  const int xcd = (int)(((long)(seg >> 2) * MSDA_XCDS) / (nsegs >> 2));       // image-major, as msda_mm_runs_k
  for (int i = tid; i < nsb; i += 256) { cnt[i] = 0; fillp[i] = 0; }
  __syncthreads();
  const int4* bb = mv.bbox + (long)seg * ntiles;
  // super-block span of a tile = of the union of its two point-group boxes; false when no tap is inside the map
  auto span = [&](const int4& v, int& sx0, int& sy0, int& sx1, int& sy1) -> bool {
    const MvBox u = mv_join(mv_unpack(v.x, v.y), mv_unpack(v.z, v.w));
    if (!mv_some(u)) return false;
    sx0 = max(u.x0, 0) / VS_SW; sx1 = min(u.x1 / VS_SW, nsx - 1); sy0 = max(u.y0, 0) / VS_SH; sy1 = min(u.y1 / VS_SH, nsy - 1);
    return sx1 >= sx0 && sy1 >= sy0;
  };
  for (int t = tid; t < ntiles; t += 256) {
    int sx0, sy0, sx1, sy1;
    if (!span(bb[t], sx0, sy0, sx1, sy1)) continue;
    if ((sx1 - sx0 + 1) * (sy1 - sy0 + 1) > VS_MAXOV) continue;
    for (int y = sy0; y <= sy1; ++y) for (int x = sx0; x <= sx1; ++x) atomicAdd(&cnt[y * nsx + x], 1);
  }
  __syncthreads();
  if (tid == 0) { int s = 0; for (int i = 0; i < nsb; ++i) { off[i] = s; s += cnt[i]; } off[nsb] = s; }
  __syncthreads();
  const bool all_stray = off[nsb] > vs.list_cap;
  int* list = vs.lists + (long)seg * vs.list_cap;
  int4* runs = mv.runs + (long)xcd * mv.runs_cap * 2;
  int nstray = 0, nvisit = 0;
  for (int t = tid; t < ntiles; t += 256) {
    int sx0, sy0, sx1, sy1;
    const int4 v = bb[t];
    if (!span(v, sx0, sy0, sx1, sy1)) continue;
    const int n = (sx1 - sx0 + 1) * (sy1 - sy0 + 1);
    if (!all_stray && n <= VS_MAXOV) {
      nvisit += n;
      for (int y = sy0; y <= sy1; ++y) for (int x = sx0; x <= sx1; ++x) { const int sb = y * nsx + x; list[off[sb] + atomicAdd(&fillp[sb], 1)] = t; }
    } else {                                                                  // a run of this one tile for msda_mm_bwd_v_k (record format: msda_mm_runs_k)
      ++nstray;
      const MvBox b0 = mv_unpack(v.x, v.y), b1 = mv_unpack(v.z, v.w);
      const bool s0 = mv_some(b0), s1 = mv_some(b1);
      const int slot = atomicAdd(mv.ctrl + xcd, 1);
      runs[2 * slot] = make_int4(seg, t | (1 << 24), s0 ? (b0.y0 << 16) | (b0.x0 & 0xffff) : 0, s0 ? ((b0.y1 - b0.y0 + 1) << 16) | (b0.x1 - b0.x0 + 1) : 0);
      runs[2 * slot + 1] = make_int4(s1 ? (b1.y0 << 16) | (b1.x0 & 0xffff) : 0, s1 ? ((b1.y1 - b1.y0 + 1) << 16) | (b1.x1 - b1.x0 + 1) : 0, 0, 0);
    }
  }
  int nitems = 0, nmulti = 0;
  if (!all_stray) {
    int4* items = vs.items + (long)xcd * vs.items_cap;
    for (int sb = tid; sb < nsb; sb += 256) {
      const int n = cnt[sb];
      if (n == 0) continue;
      const int chunks = (n + VS_T - 1) / VS_T, per = (n + chunks - 1) / chunks;
      const int slot = atomicAdd(vs.ctrl + xcd, chunks);
      for (int c = 0; c < chunks; ++c)      // {segment, super-block, first list entry, entries | single-chunk flag}
        items[slot + c] = make_int4(seg, sb, (int)((long)seg * vs.list_cap) + off[sb] + c * per, min(per, n - c * per) | (chunks == 1 ? (int)0x80000000 : 0));
      nitems += chunks; nmulti += chunks > 1 ? chunks : 0;
    }
  }
  if (nvisit) atomicAdd(vs.ctrl + 16, nvisit);
  if (nstray) atomicAdd(vs.ctrl + 17, nstray);
  if (nitems) atomicAdd(vs.ctrl + 18, nitems);
  if (nmulti) atomicAdd(vs.ctrl + 19, nmulti);
}

struct VsArgs { MmArgs f; const bf16_t* gout; float* d_value; VsWs ws; };
struct VsRaw { uint32_t o; uint32_t l01, l23; uint32_t own; float rx, ry; };      // offsets (x, y) of the lane's point, four of the 32 logits, the lane's own logit, reference point

typedef short vs_s16x2 __attribute__((ext_vector_type(2)));

// OR over the 8 lanes that share a query / over the wave, on the DPP network
__device__ __forceinline__ float vs_max8(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true)));     // quad_perm [1,0,3,2]
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true)));     // quad_perm [2,3,0,1]
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true)));    // row_half_mirror
  return v;
}
__device__ __forceinline__ float vs_sum8(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
  return v;
}
__device__ __forceinline__ int vs_wave_or(int v) {
  v |= __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false);
  v |= __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false);
  v |= __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false);
  v |= __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false);
  v |= __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false);
  return __builtin_amdgcn_readlane(v, 31) | __builtin_amdgcn_readlane(v, 63);
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) msda_mm_bwd_vs_k(VsArgs va) {
  __shared__ __attribute__((aligned(16))) bf16_t cimg[2 * VS_IMG];
  __shared__ __attribute__((aligned(16))) bf16_t stage[2 * 2 * MV_HALF];
  __shared__ int live[2][4];
  __shared__ int s_item;
  const MmArgs& a = va.f;
  const int nh64 = a.nH * 64;
  const int xcd = blockIdx.x % MSDA_XCDS;
  const int nitem = va.ws.ctrl[xcd];
  const int4* items = va.ws.items + (long)xcd * va.ws.items_cap;
  int* cursor = va.ws.ctrl + MSDA_XCDS + xcd;
  {
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (int i = threadIdx.x * 8; i < 2 * VS_IMG; i += 256 * 8) *(uint4*)(cimg + i) = z;
  }
  int c1[4] = {-1, -1, -1, -1}, c2[4] = {-1, -1, -1, -1};                    // image entries this lane wrote in the last / last but one visit
  const int qlast = a.Nq - 1;
  for (;;) {
    __syncthreads();                                                        // the previous item's fragment reads are done
    if (threadIdx.x == 0) s_item = atomicAdd(cursor, 1);
    __syncthreads();
    const int it = __builtin_amdgcn_readfirstlane(s_item);
    if (it >= nitem) break;
#pragma unroll
    for (int k = 0; k < 4; ++k) {                                            // leftovers of the previous item (own column only)
      if (c1[k] >= 0) cimg[c1[k]] = 0;
      if (c2[k] >= 0) cimg[c2[k]] = 0;
      c1[k] = -1; c2[k] = -1;
    }
    const int4 rec = items[it];
    const int seg = __builtin_amdgcn_readfirstlane(rec.x), sb = __builtin_amdgcn_readfirstlane(rec.y);
    const int first = __builtin_amdgcn_readfirstlane(rec.z), nw = __builtin_amdgcn_readfirstlane(rec.w);
    const int n = nw & 0x7fffffff;
    const bool single = nw < 0;
    const int lvl = seg & 3, bh = seg >> 2, head = bh % a.nH, b = bh / a.nH;
    const int Wl = a.lv.W[lvl], Hl = a.lv.H[lvl], lstart = a.lv.start[lvl];
    const int nsx = (Wl + VS_SW - 1) / VS_SW;
    const int sy = (sb / nsx) * VS_SH, sx = (sb - (sb / nsx) * nsx) * VS_SW;
    const float fW = (float)Wl, fH = (float)Hl, rW = 1.f / fW, rH = 1.f / fH;
    const int* list = va.ws.lists + first;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int q = tid >> 3, p = tid & 7, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tr_row = (lane >> 5) * 8 + ((lane & 15) >> 2), tr_col = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    mm_f32x16 acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i) { acc[i][0] = 0.f; acc[i][1] = 0.f; }
    auto tile_at = [&](int i) { return list[min(i, n - 1)]; };
    auto order_of = [&](int tile) { const int qi = min(tile * 32 + q, qlast); return a.order ? a.order[qi] : qi; };
    // row-relative byte offsets are 32-bit (vs_supported): uniform 64-bit bases + one zero-extended VGPR per address
    const char* off_b = (const char*)(a.off + head * 64 + lvl * 16 + p * 2);
    const char* log4_b = (const char*)(a.logit + head * 32 + p * 4);
    const char* logo_b = (const char*)(a.logit + head * 32 + lvl * 8 + p);
    const char* ref_b = (const char*)(a.ref + (long)b * a.ref_sb + (long)lvl * a.ref_sl);
    const char* go_b = (const char*)(va.gout + head * 64 + p * 8);
    const uint32_t off_ld2 = (uint32_t)a.off_ld * 2u, log_ld2 = (uint32_t)a.logit_ld * 2u, ref_sq4 = (uint32_t)a.ref_sq * 4u, go_ld2 = (uint32_t)nh64 * 2u;
    const uint32_t row0 = (uint32_t)b * (uint32_t)a.Nq;
    auto load_raw = [&](int qq) {                                         // 4 vector loads per lane (the 8 lanes of a query read contiguous bytes)
      const uint32_t row = row0 + (uint32_t)qq;
      VsRaw r;
      r.o = *(const uint32_t*)(off_b + row * off_ld2);
      const uint2 l4 = *(const uint2*)(log4_b + row * log_ld2);            // logits 4 p .. 4 p + 3 of the 32: any partition serves the max / sum
      r.l01 = l4.x; r.l23 = l4.y;
      r.own = *(const bf16_t*)(logo_b + row * log_ld2);                     // the lane's own point at the item's level
      const float2 rr = *(const float2*)(ref_b + (uint32_t)qq * ref_sq4);
      r.rx = rr.x; r.ry = rr.y;
      return r;
    };
    // 16 bytes of the lane's OWN query's gradient row (piece p of 8): plain loads into registers, parked in LDS a visit later — the compiler
    // counts the waits exactly (an LDS-DMA forces vmcnt(0) before every later read of the stage: it cannot tell the buffers apart)
    auto load_go = [&](int qq) { return (VS_DIAG & 1) ? make_uint4(0, 0, 0, 0) : *(const uint4*)(go_b + (row0 + (uint32_t)qq) * go_ld2); };
    bf16_t* const st_mine = stage + (p >> 2) * MV_HALF + q * 32 + (p & 3) * 8;
    // pipeline: a ring of THREE visit slots {raw projections, gradient-row piece, order entry, list entry}; the loop is unrolled by three so
    // that every slot lives in fixed registers (a register rotation would make the compiler wait for the load it has just issued).  At visit i,
    // after the barrier, slot i % 3 is refilled for visit i + 3 (raw projections, gradient rows), its order entry for visit i + 6, its list entry
    // for visit i + 9.  Memory latency under this access pattern is 2 - 4 us (with everything one visit ahead a visit took 2.7 us, all of it waiting)
    struct Slot { VsRaw raw; uint4 go; int qq, tt; };
    // order entry of the lane's query in `tile`; bit 31 set = the query does not exist (tail of the last tile)
    auto order_v = [&](int tile) { const int qi = tile * 32 + q; const int qq = order_of(tile); return qi < a.Nq ? qq : (qq | (int)0x80000000); };
    auto fill = [&](Slot& sl) {                                            // loads of the slot's next visit from its order entry
      const int qq = sl.qq & 0x7fffffff;
      sl.go = load_go(qq);
      sl.raw = load_raw(qq);
      sl.raw.own |= (uint32_t)sl.qq & 0x80000000u;                          // validity rides in the unused upper half of the 16-bit logit
    };
    Slot S0, S1, S2;
    S0.qq = order_v(tile_at(0)); S1.qq = order_v(tile_at(1)); S2.qq = order_v(tile_at(2));
    fill(S0); fill(S1); fill(S2);
    S0.qq = order_v(tile_at(3)); S1.qq = order_v(tile_at(4)); S2.qq = order_v(tile_at(5));
    S0.tt = tile_at(6); S1.tt = tile_at(7); S2.tt = tile_at(8);
    auto visit = [&](Slot& sl, int i) __attribute__((always_inline)) {
      const int buf = i & 1;
      bf16_t* img = cimg + buf * VS_IMG;
      const VsRaw raw0 = sl.raw;
      const bool qok = (int)raw0.own >= 0;
      // attention weight of the lane's point: softmax over the 32 logits of (query, head) = 8 lanes x 4 logits
      float aw;
      {
        const float e0 = __uint_as_float(raw0.l01 << 16), e1 = __uint_as_float(raw0.l01 & 0xffff0000u);
        const float e2 = __uint_as_float(raw0.l23 << 16), e3 = __uint_as_float(raw0.l23 & 0xffff0000u);
        const float m = vs_max8(fmaxf(fmaxf(e0, e1), fmaxf(e2, e3)));
        const float sum = vs_sum8((__expf(e0 - m) + __expf(e1 - m)) + (__expf(e2 - m) + __expf(e3 - m)));
        const float inv = qok ? 1.f / sum : 0.f;
        aw = __expf(__uint_as_float(raw0.own << 16) - m) * inv;
      }
      // the tap arithmetic of mm_taps / msda_mm_bwd_lw_k to the bit (the tile lists come from boxes computed with it)
      int e[4];                                                              // image row of the four corners, -1 = not in this super-block / masked
      float w[4];
      {
        const float ox = __uint_as_float(raw0.o << 16), oy = __uint_as_float(raw0.o & 0xffff0000u);
        const float lx = raw0.rx + mm_div(ox, fW, rW), ly = raw0.ry + mm_div(oy, fH, rH);
        const float x = lx * fW - 0.5f, y = ly * fH - 0.5f;
        const bool in = qok && y > -1.f && x > -1.f && y < fH && x < fW;
        const float xc = fminf(fmaxf(x, -1.f), fW), yc = fminf(fmaxf(y, -1.f), fH);
        const float xf = floorf(xc), yf = floorf(yc);
        const int x0 = (int)xf, y0 = (int)yf;
        const float ax = xc - xf, ay = yc - yf;
        const int xa = min(max(x0, 0), Wl - 1), xb = min(max(x0 + 1, 0), Wl - 1);
        const int ya = min(max(y0, 0), Hl - 1), yb = min(max(y0 + 1, 0), Hl - 1);
        const float wgt = in ? aw : 0.f;
        const float wxa = x0 >= 0 ? (1.f - ax) * wgt : 0.f, wxb = x0 + 1 < Wl ? ax * wgt : 0.f;
        const float wya = y0 >= 0 ? 1.f - ay : 0.f, wyb = y0 + 1 < Hl ? ay : 0.f;
        const mm_f16x2 wt = __builtin_bit_cast(mm_f16x2, __builtin_amdgcn_cvt_pkrtz(wya * wxa, wya * wxb));
        const mm_f16x2 wb = __builtin_bit_cast(mm_f16x2, __builtin_amdgcn_cvt_pkrtz(wyb * wxa, wyb * wxb));
        w[0] = (float)wt[0]; w[1] = (float)wt[1]; w[2] = (float)wb[0]; w[3] = (float)wb[1];
        const int cxa = xa - sx, cxb = xb - sx, cya = ya - sy, cyb = yb - sy;
        const bool ixa = (unsigned)cxa < (unsigned)VS_SW, ixb = (unsigned)cxb < (unsigned)VS_SW, iya = (unsigned)cya < (unsigned)VS_SH, iyb = (unsigned)cyb < (unsigned)VS_SH;
        e[0] = (ixa && iya && w[0] != 0.f) ? (cya * VS_SW + cxa) : -1;
        e[1] = (ixb && iya && w[1] != 0.f) ? (cya * VS_SW + cxb) : -1;
        e[2] = (ixa && iyb && w[2] != 0.f) ? (cyb * VS_SW + cxa) : -1;
        e[3] = (ixb && iyb && w[3] != 0.f) ? (cyb * VS_SW + cxb) : -1;
      }
      int mask = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (c2[k] >= 0) cimg[c2[k]] = 0;                                    // this buffer's entries of two visits ago (read by the MFMAs of visit i - 2)
        c2[k] = c1[k];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        c1[k] = -1;
        if (e[k] >= 0 && !(VS_DIAG & 2)) {
          mask |= 1 << (e[k] >> 5);
          const int el = buf * VS_IMG + e[k] * VS_CROW + q;
          c1[k] = el;
          const short hv = (short)mm_bf(w[k]);
          vs_s16x2 pv;
          pv[0] = (q & 1) ? (short)0 : hv; pv[1] = (q & 1) ? hv : (short)0;
          __builtin_amdgcn_ds_atomic_fadd_v2bf16(MM_LDS_PTR(vs_s16x2, cimg + (el & ~1)), pv);
        }
      }
      mask = vs_wave_or(mask);
      if (lane == 0) live[buf][wv] = mask;
      *(uint4*)(st_mine + buf * 2 * MV_HALF) = sl.go;                        // this visit's gradient rows (loaded three visits ago) -> stage[buf]
      __syncthreads();
      fill(sl);                                                             // visit i + 3
      sl.qq = order_v(sl.tt);                                               // visit i + 6
      sl.tt = tile_at(i + 9);
      const int lm = __builtin_amdgcn_readfirstlane(live[buf][0] | live[buf][1] | live[buf][2] | live[buf][3]);
      const int mine = (lm >> (3 * wv)) & 7;
      if (mine && !(VS_DIAG & 4)) {
        const bf16_t* st = stage + buf * 2 * MV_HALF;
        mm_bf16x8 Bf[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const bf16_t* pp = st + half * MV_HALF + (ks * 16 + tr_row) * 32 + tr_col;
            const mm_bf16x4 u0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(MM_LDS_PTR(mm_bf16x4, pp));
            const mm_bf16x4 u1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(MM_LDS_PTR(mm_bf16x4, pp + 4 * 32));
            Bf[ks][half] = __builtin_shufflevector(u0, u1, 0, 1, 2, 3, 4, 5, 6, 7);
          }
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
          if ((mine >> rb) & 1) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              const mm_bf16x8 A = *(const mm_bf16x8*)(img + ((wv * 3 + rb) * 32 + (lane & 31)) * VS_CROW + ks * 16 + (lane >> 5) * 8);
              acc[rb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bf[ks][0], acc[rb][0], 0, 0, 0);
              acc[rb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bf[ks][1], acc[rb][1], 0, 0, 0);
            }
          }
        }
      }
    };
    // two rounds of the ring per loop iteration: at a loop header the compiler's wait-count bookkeeping treats everything in flight as of
    // unknown age, so the first use of a prefetched register after it drains the whole queue — once per six visits instead of once per three
#pragma unroll 1
    for (int i = 0; i < n; i += 6) {
      visit(S0, i);
      if (i + 1 >= n) break;
      visit(S1, i + 1);
      if (i + 2 >= n) break;
      visit(S2, i + 2);
      if (i + 3 >= n) break;
      visit(S0, i + 3);
      if (i + 4 >= n) break;
      visit(S1, i + 4);
      if (i + 5 >= n) break;
      visit(S2, i + 5);
    }
    // C/D layout: column = lane & 31 (channel of the half), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the 32-row block
    if (!(VS_DIAG & 8)) {
      float* dvb = va.d_value + ((long)b * a.Nv * a.nH + head) * 64 + (lane & 31);
#pragma unroll
      for (int rb = 0; rb < 3; ++rb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (wv * 3 + rb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int y = (m * 2731) >> 16, x = m - y * VS_SW;                  // m / 24 for m < 384
          const int px = sx + x, py = sy + y;
          if (px < Wl && py < Hl) {
            float* dst = dvb + (long)(lstart + mul24(py, Wl) + px) * nh64;
            const float v0 = acc[rb][0][r], v1 = acc[rb][1][r];
            if (single) { dst[0] = v0; dst[32] = v1; }
            else { if (v0 != 0.f) atomicAdd(dst, v0); if (v1 != 0.f) atomicAdd(dst + 32, v1); }
          }
        }
      }
    }
  }
}

static int mv_launch_runs(const MvArgs& va, hipStream_t s) {
  static int per_cu = 0, n_cu = 0;
  if (!per_cu) {
    int dev = 0, v = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return GE_ERR_UNSUPPORTED;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, msda_mm_bwd_v_k, 64, 0) != hipSuccess || v < 1) v = 8;
    n_cu = pr.multiProcessorCount; per_cu = v;
  }
  const long blocks = std::max((long)n_cu * per_cu / MSDA_XCDS * MSDA_XCDS, (long)MSDA_XCDS);
  msda_mm_bwd_v_k<<<(unsigned)blocks, 64, 0, s>>>(va);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

static int vs_supported(int B, int Nq, int Nv, int nH, int L, int P, int dtype, const MsdaLevels& lv) {
  if (!msda_mm_supported(B, Nq, Nv, nH, L, P, dtype, lv)) return 0;
  if (vs_max_sb(lv) > VS_MAXSB) return 0;
  const long ntiles = (Nq + 31) / 32;
  if ((long)B * nH * 4 * VS_OVF * ntiles >= (1L << 31) || ntiles >= (1 << 24)) return 0;      // list entries are addressed with 32-bit indices
  return 1;
}
// msda_mm_bwd_vs_k forms row addresses as a uniform 64-bit base + a 32-bit byte offset: every row-addressed tensor must stay below 4 GB
static int vs_offsets_fit(int B, int Nq, int nH, long off_ld, long logit_ld, long ref_sq) {
  const long rows = (long)B * Nq, lim = 1L << 32;
  return rows * off_ld * 2 < lim && rows * logit_ld * 2 < lim && rows * nH * 64 * 2 < lim && (long)Nq * (ref_sq < 0 ? -ref_sq : ref_sq) * 4 < lim && ref_sq >= 0;
}

extern "C" size_t ge_msda_bwd_vs_workspace(const int* spatial_hw, int B, int Nv, int Nq, int nH, int L, int P) {
  MsdaLevels lv;
  if (!spatial_hw || B <= 0 || Nq <= 0 || msda_levels(spatial_hw, L, Nv, lv) || !vs_supported(B, Nq, Nv, nH, L, P, GE_BF16, lv)) return 0;
  return vs_ws_layout(B, Nq, nH, lv, nullptr, nullptr, nullptr);
}

// Byte offset, inside the workspace, of four ints the index kernel leaves behind: {tile visits, stray tiles, work items, items of multi-chunk super-blocks}
extern "C" size_t ge_msda_bwd_vs_stats_offset(const int* spatial_hw, int B, int Nv, int Nq, int nH, int L, int P) {
  MsdaLevels lv;
  if (!spatial_hw || B <= 0 || Nq <= 0 || msda_levels(spatial_hw, L, Nv, lv) || !vs_supported(B, Nq, Nv, nH, L, P, GE_BF16, lv)) return 0;
  size_t o = 0;
  vs_ws_layout(B, Nq, nH, lv, nullptr, nullptr, nullptr, &o);
  return o + 16 * sizeof(int);
}

// d_value of ge_msda_fwd_mm on the value-stationary kernel.  `d_value` (B, Nv, nH, 64) f32 must be ZERO on entry (super-blocks nothing samples are
// not written; multi-chunk super-blocks and stray tiles are accumulated with atomics).  `workspace` (ge_msda_bwd_vs_workspace bytes) must be the one
// ge_msda_bwd_lw_mm was given for the same inputs: its head holds the per-tile tap boxes that kernel leaves behind.
extern "C" int ge_msda_bwd_value_vs(const int* spatial_hw, const void* off_raw, long off_ld, const void* logit_raw, long logit_ld,
                                    const float* ref, long ref_sb, long ref_sq, long ref_sl, const int* order, const void* d_out,
                                    float* d_value, void* workspace, size_t workspace_bytes, int B, int Nv, int Nq, int nH, int L, int P,
                                    int dtype, void* stream) {
  if (!spatial_hw || !off_raw || !logit_raw || !ref || !d_out || !d_value || !workspace || B < 0 || Nq < 0) return GE_ERR_BAD_ARG;
  MsdaLevels lv;
  int e = msda_levels(spatial_hw, L, Nv, lv);
  if (e) return e;
  if (!vs_supported(B, Nq, Nv, nH, L, P, dtype, lv) || !vs_offsets_fit(B, Nq, nH, off_ld, logit_ld, ref_sq)) return GE_ERR_UNSUPPORTED;
  if ((((uintptr_t)off_raw | (uintptr_t)logit_raw | (uintptr_t)d_out) & 15) || off_ld % 8 || logit_ld % 8) return GE_ERR_BAD_ARG;
  if (B == 0 || Nq == 0) return GE_OK;
  MvArgs mva;
  VsArgs va;
  if (workspace_bytes < vs_ws_layout(B, Nq, nH, lv, (char*)workspace, &mva.ws, &va.ws)) return GE_ERR_BAD_ARG;
  MmArgs& a = va.f;
  a.value = nullptr; a.lv = lv;
  a.off = (const bf16_t*)off_raw; a.off_ld = off_ld; a.logit = (const bf16_t*)logit_raw; a.logit_ld = logit_ld;
  a.ref = ref; a.ref_sb = ref_sb; a.ref_sq = ref_sq; a.ref_sl = ref_sl; a.order = order;
  a.out = nullptr; a.loc_out = nullptr; a.attw_out = nullptr;
  a.B = B; a.Nv = Nv; a.Nq = Nq; a.nH = nH; a.ntiles = (Nq + 31) / 32; a.qpitch = Nq;
  va.gout = (const bf16_t*)d_out; va.d_value = d_value;
  mva.f = a; mva.gout = va.gout; mva.d_value = d_value;
  hipStream_t s = ge_stream(stream);
  hipError_t he = hipMemsetAsync(mva.ws.ctrl, 0, 32 * sizeof(int), s);
  if (he != hipSuccess) return (int)he;
  he = hipMemsetAsync(va.ws.ctrl, 0, 32 * sizeof(int), s);
  if (he != hipSuccess) return (int)he;
  const int nsegs = B * nH * 4;
  msda_vs_index_k<<<(unsigned)nsegs, 256, 0, s>>>(mva.ws, va.ws, lv, a.ntiles, nsegs);
  GE_LAUNCH_CHECK();
  static int vs_blocks = 0;
  if (!vs_blocks) {
    int dev = 0, v = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return GE_ERR_UNSUPPORTED;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, msda_mm_bwd_vs_k, 256, 0) != hipSuccess || v < 1) v = 2;
    vs_blocks = std::max(pr.multiProcessorCount * v / MSDA_XCDS * MSDA_XCDS, MSDA_XCDS);
  }
  msda_mm_bwd_vs_k<<<(unsigned)vs_blocks, 256, 0, s>>>(va);
  GE_LAUNCH_CHECK();
  return mv_launch_runs(mva, s);             // stray tiles (none for coherent geometries: the kernel then finds empty run lists)
}
