// Bandwidth-bound decoder glue on channels-last maps, fused into its producer / consumer passes (gfx950; no MFMA: streaming work).
//
// tools/library_roofline.py (profiles/r3_library_roofline.json) puts the 3x3 convolutions of the decoder on the MFMA side of the
// ridge (411 - 2141 FLOP/B, MIOpen igemm at 350 - 650 TFLOP/s) and what surrounds them on the bandwidth side.  Two such surroundings:
//
//   ge_upcat_nhwc_*   UpSample block of the DenseDepth head (reference depth/models/decode_heads/densedepth_head.py:25-27):
//                       F.interpolate(x, size=skip.shape, bilinear, align_corners=True) -> torch.cat([up, skip], 1) -> convA
//                     as ONE pass that writes the concat buffer: the up-sampled tensor is never materialised (one write + one read of
//                     N*H*W*Cu saved) and ATen's cat (read + write of the whole buffer) disappears.  Backward: the transpose of the
//                     interpolation reads the d_up columns of d_cat in place (row pitch Cu + Cs), d_skip is a view.
//   ge_upsum_nhwc_fwd  trunk of the PE necks (depth/models/necks/pemask_neck.py:52-64, dynamicpe_neck.py:512-539):
//                       x = conv4(f4) + sum_{i<4} F.interpolate(conv_i(f_i), size=f4.shape, bilinear, align_corners=True)
//                     as ONE pass over the 176 x 560 output (4 x 4 cached taps of the coarse maps + one streaming read) instead of four
//                     up-sampled tensors written and three-and-a-bit adds over them (4 writes + 8 reads + 4 writes of N*H*W*64 saved).
//                     Its backward is the existing per-level ge_bilinear_nhwc_bwd on the shared d_out (d_t4 = d_out).
#include "common.h"

template <typename T> static bool dec_geom(int C, int& lanes) {
  constexpr int VN = V8<T>::N;
  if (C <= 0 || C % VN) return false;
  lanes = C / VN;
  return true;
}
static inline bool dec_aligned(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
  return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d)) & 15) == 0;
}

// ------------------------------------------------------------------------------------------------ up-sample + concat
// grid.y = output row (n * H + y), grid.x * 256 threads over (x, 16-byte channel vector of the Cu + Cs concat row)
template <typename T>
__global__ void __launch_bounds__(256) upcat_fwd_k(const T* __restrict__ coarse, const T* __restrict__ skip, T* __restrict__ out, int Hc, int Wc,
                                                   int H, int W, int Cu, int Cs, int align) {
  constexpr int VN = V8<T>::N;
  const int lu = Cu / VN, lt = (Cu + Cs) / VN, Co = Cu + Cs;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= W * lt) return;
  const int x = idx / lt, j = idx - x * lt;
  const int row = blockIdx.y, n = row / H, y = row - n * H;
  T* o = out + ((long)row * W + x) * Co + j * VN;
  float v[VN];
  if (j >= lu) {
    V8<T>::ld(skip + ((long)row * W + x) * Cs + (j - lu) * VN, v);
    V8<T>::st(o, v);
    return;
  }
  const float sy = ge_scale(Hc, H, align), sx = ge_scale(Wc, W, align);
  const Lerp ly = ge_lerp(y, Hc, sy, align), lx = ge_lerp(x, Wc, sx, align);
  const T* base = coarse + (long)n * Hc * Wc * Cu + j * VN;
  float v00[VN], v01[VN], v10[VN], v11[VN];
  V8<T>::ld(base + ((long)ly.i0 * Wc + lx.i0) * Cu, v00);
  V8<T>::ld(base + ((long)ly.i0 * Wc + lx.i1) * Cu, v01);
  V8<T>::ld(base + ((long)ly.i1 * Wc + lx.i0) * Cu, v10);
  V8<T>::ld(base + ((long)ly.i1 * Wc + lx.i1) * Cu, v11);
#pragma unroll
  for (int k = 0; k < VN; ++k) v[k] = ly.w0 * (lx.w0 * v00[k] + lx.w1 * v01[k]) + ly.w1 * (lx.w0 * v10[k] + lx.w1 * v11[k]);   // = bilinear_nhwc_fwd_k
  V8<T>::st(o, v);
}

// output indices whose taps can touch input index X: src(o) in (X - 1, X + 1); one spare candidate on each side against rounding
// (weights are recomputed exactly, a spare candidate contributes 0) — 5 instead of the 7 of the generic kernel at factor 2
__device__ __forceinline__ void dec_cand_range(int X, int out, float scale, bool align, int& lo, int& hi) {
  if (scale <= 0.f) { lo = 0; hi = out - 1; return; }
  float a, b;
  if (align) { a = ((float)X - 1.f) / scale; b = ((float)X + 1.f) / scale; }
  else { a = ((float)X - 0.5f) / scale - 0.5f; b = ((float)X + 1.5f) / scale - 0.5f; }
  lo = (int)floorf(a);
  hi = (int)ceilf(b);
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}
// d_coarse[n, Y, X, :] = sum over the outputs (oy, ox) whose taps touch (Y, X) of w * d_out[n, oy, ox, 0:Cu]; d_out rows have `pitch`
// elements (the concat row).  Deterministic gather, no atomics.  grid.y = coarse row (n * Hc + Y).
template <typename T>
__global__ void __launch_bounds__(256) upcat_bwd_k(const T* __restrict__ gout, T* __restrict__ gin, int Hc, int Wc, int H, int W, int Cu, int pitch,
                                                   int align) {
  constexpr int VN = V8<T>::N;
  const int lu = Cu / VN;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Wc * lu) return;
  const int X = idx / lu, j = idx - X * lu;
  const int row = blockIdx.y, n = row / Hc, Y = row - n * Hc;
  const float sy = ge_scale(Hc, H, align), sx = ge_scale(Wc, W, align);
  int ylo, yhi, xlo, xhi;
  dec_cand_range(Y, H, sy, align, ylo, yhi);
  dec_cand_range(X, W, sx, align, xlo, xhi);
  const T* g = gout + (long)n * H * W * pitch + j * VN;
  float acc[VN];
#pragma unroll
  for (int k = 0; k < VN; ++k) acc[k] = 0.f;
  for (int oy = ylo; oy <= yhi; ++oy) {
    const Lerp ly = ge_lerp(oy, Hc, sy, align);
    const float wy = (ly.i0 == Y ? ly.w0 : 0.f) + (ly.i1 == Y ? ly.w1 : 0.f);
    if (wy == 0.f) continue;
    float r[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) r[k] = 0.f;
    for (int ox = xlo; ox <= xhi; ++ox) {
      const Lerp lx = ge_lerp(ox, Wc, sx, align);
      const float wx = (lx.i0 == X ? lx.w0 : 0.f) + (lx.i1 == X ? lx.w1 : 0.f);
      if (wx == 0.f) continue;
      float v[VN];
      V8<T>::ld(g + ((long)oy * W + ox) * pitch, v);
#pragma unroll
      for (int k = 0; k < VN; ++k) r[k] += wx * v[k];
    }
#pragma unroll
    for (int k = 0; k < VN; ++k) acc[k] += wy * r[k];
  }
  V8<T>::st(gin + ((long)row * Wc + X) * Cu + j * VN, acc);
}

template <typename T>
static int upcat_launch(bool fwd, const void* a, const void* b, void* o, int N, int Cu, int Hc, int Wc, int Cs, int H, int W, int align, hipStream_t s) {
  int lu, ls;
  if (!dec_geom<T>(Cu, lu) || !dec_geom<T>(Cs, ls) || !dec_aligned(a, b, o)) return GE_ERR_UNSUPPORTED;
  if ((long)N * H > 65535 || (long)N * Hc > 65535 || (long)W * (lu + ls) > (1L << 30)) return GE_ERR_UNSUPPORTED;
  if (N == 0) return GE_OK;
  if (fwd) {
    const unsigned gx = (unsigned)((W * (lu + ls) + 255) / 256);
    upcat_fwd_k<T><<<dim3(gx, N * H), 256, 0, s>>>((const T*)a, (const T*)b, (T*)o, Hc, Wc, H, W, Cu, Cs, align);
  } else {
    const unsigned gx = (unsigned)((Wc * lu + 255) / 256);
    upcat_bwd_k<T><<<dim3(gx, N * Hc), 256, 0, s>>>((const T*)a, (T*)o, Hc, Wc, H, W, Cu, Cu + Cs, align);
  }
  GE_LAUNCH_CHECK();
  return GE_OK;
}
// coarse (N, Hc, Wc, Cu), skip (N, H, W, Cs) -> out (N, H, W, Cu + Cs) = [bilinear(coarse -> H x W) | skip], channels-last
extern "C" int ge_upcat_nhwc_fwd(const void* coarse, const void* skip, void* out, int N, int Cu, int Hc, int Wc, int Cs, int H, int W,
                                 int align_corners, int dtype, void* stream) {
  if (!coarse || !skip || !out || N < 0 || Hc <= 0 || Wc <= 0 || H <= 0 || W <= 0) return GE_ERR_BAD_ARG;
  if (dtype == GE_F32) return upcat_launch<float>(true, coarse, skip, out, N, Cu, Hc, Wc, Cs, H, W, align_corners, ge_stream(stream));
  if (dtype == GE_BF16) return upcat_launch<bf16_t>(true, coarse, skip, out, N, Cu, Hc, Wc, Cs, H, W, align_corners, ge_stream(stream));
  return GE_ERR_UNSUPPORTED;
}
// d_out (N, H, W, Cu + Cs) -> d_coarse (N, Hc, Wc, Cu), fully written (d_skip is the channel slice [Cu, Cu + Cs) of d_out: a view)
extern "C" int ge_upcat_nhwc_bwd(const void* d_out, void* d_coarse, int N, int Cu, int Hc, int Wc, int Cs, int H, int W, int align_corners,
                                 int dtype, void* stream) {
  if (!d_out || !d_coarse || N < 0 || Hc <= 0 || Wc <= 0 || H <= 0 || W <= 0) return GE_ERR_BAD_ARG;
  if (dtype == GE_F32) return upcat_launch<float>(false, d_out, nullptr, d_coarse, N, Cu, Hc, Wc, Cs, H, W, align_corners, ge_stream(stream));
  if (dtype == GE_BF16) return upcat_launch<bf16_t>(false, d_out, nullptr, d_coarse, N, Cu, Hc, Wc, Cs, H, W, align_corners, ge_stream(stream));
  return GE_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------ sum of up-sampled maps
struct UpSumSrc { const void* p[4]; int H[4], W[4]; };
template <typename T>
__global__ void __launch_bounds__(256) upsum_fwd_k(UpSumSrc src, int nsrc, const T* __restrict__ fine, T* __restrict__ out, int H, int W, int C,
                                                   int align) {
  constexpr int VN = V8<T>::N;
  const int lanes = C / VN;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= W * lanes) return;
  const int x = idx / lanes, j = idx - x * lanes;
  const int row = blockIdx.y, n = row / H, y = row - n * H;
  const long o = ((long)row * W + x) * C + j * VN;
  // the reference adds in the order ((((t0_up) + t1_up) + t2_up) + t3_up) + t4, each partial sum rounded to the storage type
  float acc[VN];
  for (int i = 0; i < nsrc; ++i) {
    const int Hc = src.H[i], Wc = src.W[i];
    const float sy = ge_scale(Hc, H, align), sx = ge_scale(Wc, W, align);
    const Lerp ly = ge_lerp(y, Hc, sy, align), lx = ge_lerp(x, Wc, sx, align);
    const T* base = (const T*)src.p[i] + (long)n * Hc * Wc * C + j * VN;
    float v00[VN], v01[VN], v10[VN], v11[VN];
    V8<T>::ld(base + ((long)ly.i0 * Wc + lx.i0) * C, v00);
    V8<T>::ld(base + ((long)ly.i0 * Wc + lx.i1) * C, v01);
    V8<T>::ld(base + ((long)ly.i1 * Wc + lx.i0) * C, v10);
    V8<T>::ld(base + ((long)ly.i1 * Wc + lx.i1) * C, v11);
#pragma unroll
    for (int k = 0; k < VN; ++k) {
      const float up = Io<T>::rt(ly.w0 * (lx.w0 * v00[k] + lx.w1 * v01[k]) + ly.w1 * (lx.w0 * v10[k] + lx.w1 * v11[k]));
      acc[k] = i == 0 ? up : Io<T>::rt(acc[k] + up);
    }
  }
  float f[VN];
  V8<T>::ld(fine + o, f);
#pragma unroll
  for (int k = 0; k < VN; ++k) acc[k] = nsrc ? acc[k] + f[k] : f[k];
  V8<T>::st(out + o, acc);
}
// out (N, H, W, C) = sum_i bilinear(src_i (N, H_i, W_i, C) -> H x W) + fine (N, H, W, C); nsrc <= 4; hw = {H_0, W_0, H_1, W_1, ...}
extern "C" int ge_upsum_nhwc_fwd(const void* const* srcs, const int* hw, int nsrc, const void* fine, void* out, int N, int C, int H, int W,
                                 int align_corners, int dtype, void* stream) {
  if (!srcs || !hw || !fine || !out || nsrc < 0 || nsrc > 4 || N < 0 || H <= 0 || W <= 0) return GE_ERR_BAD_ARG;
  if (dtype != GE_F32 && dtype != GE_BF16) return GE_ERR_UNSUPPORTED;
  const int vn = dtype == GE_F32 ? 4 : 8;
  if (C <= 0 || C % vn || !dec_aligned(fine, out) || (long)N * H > 65535) return GE_ERR_UNSUPPORTED;
  UpSumSrc s;
  for (int i = 0; i < 4; ++i) {
    s.p[i] = i < nsrc ? srcs[i] : nullptr; s.H[i] = i < nsrc ? hw[2 * i] : 1; s.W[i] = i < nsrc ? hw[2 * i + 1] : 1;
    if (i < nsrc && (!s.p[i] || s.H[i] <= 0 || s.W[i] <= 0 || !dec_aligned(s.p[i]))) return GE_ERR_BAD_ARG;
  }
  if (N == 0) return GE_OK;
  const unsigned gx = (unsigned)((W * (C / vn) + 255) / 256);
  if (dtype == GE_F32) upsum_fwd_k<float><<<dim3(gx, N * H), 256, 0, ge_stream(stream)>>>(s, nsrc, (const float*)fine, (float*)out, H, W, C, align_corners);
  else upsum_fwd_k<bf16_t><<<dim3(gx, N * H), 256, 0, ge_stream(stream)>>>(s, nsrc, (const bf16_t*)fine, (bf16_t*)out, H, W, C, align_corners);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
