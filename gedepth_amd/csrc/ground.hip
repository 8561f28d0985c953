// Streaming (HBM-bound) kernels of the GEDepth hot path for gfx950:
//   bilinear resize fwd / deterministic-gather bwd, ground-embedding prior (adaptive + vanilla),
//   depth fusion, offline ground-plane / slope-class maps, SiLog statistics, fused AdamW.
// None of these is a contraction, so no MFMA: one pixel (or one 16-byte vector) per lane, coalesced
// along W, low-resolution operands served from L1/L2.  See DESIGN.md for bytes/pixel per kernel.
// Keep mul/add un-fused everywhere in this translation unit (including inlined header helpers) so that resize
// weights, the validity mask and the fp64 ground-plane map follow the same IEEE operation sequence as the
// reference's ATen / numpy code.  Must precede every include: contraction flags are attached per operation.
#pragma clang fp contract(off)
#include "common.h"

// ====================================================================================== bilinear
// Forward: a lane produces VN consecutive outputs of one row (one 16-byte store, the row interpolation weights and the
// 64-bit index split paid once); the 2 x (VN + 1) input taps are neighbours and come from L1.
template <typename T, bool VEC>
__global__ void __launch_bounds__(256) bilinear_fwd_k(const T* __restrict__ in, T* __restrict__ out, int NC,
                                                      int Hi, int Wi, int Ho, int Wo, int align, int C,
                                                      long in_bs, long in_ps, long out_bs, long out_ps) {
  constexpr int VN = VEC ? V8<T>::N : 1;
  const float sy = ge_scale(Hi, Ho, align), sx = ge_scale(Wi, Wo, align);
  const int Wg = Wo / VN;                                   // VEC: Wo % VN == 0 (checked by the launcher)
  const long total = (long)NC * Ho * Wg;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % Wg) * VN;
    long t = idx / Wg;
    const int y = (int)(t % Ho);
    const long nc = t / Ho;
    const Lerp ly = ge_lerp(y, Hi, sy, align);
    const long n_ = nc / C, c_ = nc - n_ * C;
    const T* p0 = in + n_ * in_bs + c_ * in_ps + (long)ly.i0 * Wi;
    const T* p1 = in + n_ * in_bs + c_ * in_ps + (long)ly.i1 * Wi;
    float v[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) {
      const Lerp lx = ge_lerp(x + k, Wi, sx, align);
      const float v00 = Io<T>::ld(p0 + lx.i0), v01 = Io<T>::ld(p0 + lx.i1);
      const float v10 = Io<T>::ld(p1 + lx.i0), v11 = Io<T>::ld(p1 + lx.i1);
      v[k] = ly.w0 * (lx.w0 * v00 + lx.w1 * v01) + ly.w1 * (lx.w0 * v10 + lx.w1 * v11);
    }
    T* o = out + n_ * out_bs + c_ * out_ps + (long)y * Wo + x;
    if (VEC) V8<T>::st(o, v);
    else Io<T>::st(o, v[0]);
  }
}

// candidate output indices whose taps can touch input index X
__device__ __forceinline__ void cand_range(int X, int in, int out, float scale, bool align, int& lo, int& hi) {
  if (scale <= 0.f) { lo = 0; hi = out - 1; return; }
  float a, b;
  if (align) { a = ((float)X - 1.f) / scale; b = ((float)X + 1.f) / scale; }
  else { a = ((float)X - 0.5f) / scale - 0.5f; b = ((float)X + 1.5f) / scale - 0.5f; }
  lo = (int)floorf(a) - 1;
  hi = (int)ceilf(b) + 1;
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}

#define GE_MAXC 8   // candidate taps per dimension kept in registers (up-sampling factors <= 2.5; wider ranges take the loop)

template <typename T>
__global__ void __launch_bounds__(256) bilinear_bwd_k(const T* __restrict__ gout, T* __restrict__ gin, int NC,
                                                      int Hi, int Wi, int Ho, int Wo, int align, int C,
                                                      long gout_bs, long gout_ps, long gin_bs, long gin_ps) {
  const float sy = ge_scale(Hi, Ho, align), sx = ge_scale(Wi, Wo, align);
  const long total = (long)NC * Hi * Wi;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int X = (int)(idx % Wi);
    long t = idx / Wi;
    int Y = (int)(t % Hi);
    long nc = t / Hi;
    int ylo, yhi, xlo, xhi;
    cand_range(Y, Hi, Ho, sy, align, ylo, yhi);
    cand_range(X, Wi, Wo, sx, align, xlo, xhi);
    const long n_ = nc / C, c_ = nc - n_ * C;
    const T* g = gout + n_ * gout_bs + c_ * gout_ps;
    float acc = 0.f;
    if (xhi - xlo < GE_MAXC) {
      // the column weights do not depend on the row: evaluate them once (<= GE_MAXC candidates in registers)
      float wxs[GE_MAXC];
#pragma unroll
      for (int k = 0; k < GE_MAXC; ++k) {
        const int ox = min(xlo + k, Wo - 1);
        const Lerp lx = ge_lerp(ox, Wi, sx, align);
        wxs[k] = xlo + k <= xhi ? (lx.i0 == X ? lx.w0 : 0.f) + (lx.i1 == X ? lx.w1 : 0.f) : 0.f;
      }
      for (int oy = ylo; oy <= yhi; ++oy) {
        const Lerp ly = ge_lerp(oy, Hi, sy, align);
        const float wy = (ly.i0 == Y ? ly.w0 : 0.f) + (ly.i1 == Y ? ly.w1 : 0.f);
        if (wy == 0.f) continue;
        const T* gr = g + (long)oy * Wo + xlo;
        float row = 0.f;
#pragma unroll
        for (int k = 0; k < GE_MAXC; ++k)
          if (wxs[k] != 0.f) row += wxs[k] * Io<T>::ld(gr + k);
        acc += wy * row;
      }
    } else {
      for (int oy = ylo; oy <= yhi; ++oy) {
        Lerp ly = ge_lerp(oy, Hi, sy, align);
        float wy = (ly.i0 == Y ? ly.w0 : 0.f) + (ly.i1 == Y ? ly.w1 : 0.f);
        if (wy == 0.f) continue;
        float row = 0.f;
        for (int ox = xlo; ox <= xhi; ++ox) {
          Lerp lx = ge_lerp(ox, Wi, sx, align);
          float wx = (lx.i0 == X ? lx.w0 : 0.f) + (lx.i1 == X ? lx.w1 : 0.f);
          if (wx != 0.f) row += wx * Io<T>::ld(g + (long)oy * Wo + ox);
        }
        acc += wy * row;
      }
    }
    Io<T>::st(gin + n_ * gin_bs + c_ * gin_ps + (long)Y * Wi + X, acc);
  }
}

template <typename T>
static int bilinear_fwd_launch(const void* in, void* out, int N, int C, int Hi, int Wi, int Ho, int Wo, int align,
                               long ibs, long ips, long obs, long ops, hipStream_t s) {
  long total = (long)N * C * Ho * Wo;
  if (total == 0) return GE_OK;
  const bool vec = Wo % V8<T>::N == 0 && obs % V8<T>::N == 0 && ops % V8<T>::N == 0 && (((uintptr_t)out) & 15) == 0;
  if (vec)
    bilinear_fwd_k<T, true><<<ge_blocks(total / V8<T>::N, 256, 65536), 256, 0, s>>>((const T*)in, (T*)out, N * C, Hi, Wi, Ho, Wo, align, C, ibs, ips, obs, ops);
  else
    bilinear_fwd_k<T, false><<<ge_blocks(total, 256, 65536), 256, 0, s>>>((const T*)in, (T*)out, N * C, Hi, Wi, Ho, Wo, align, C, ibs, ips, obs, ops);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
template <typename T>
static int bilinear_bwd_launch(const void* gout, void* gin, int N, int C, int Hi, int Wi, int Ho, int Wo, int align,
                               long gobs, long gops, long gibs, long gips, hipStream_t s) {
  long total = (long)N * C * Hi * Wi;
  if (total == 0) return GE_OK;
  bilinear_bwd_k<T><<<ge_blocks(total, 256, 65536), 256, 0, s>>>((const T*)gout, (T*)gin, N * C, Hi, Wi, Ho, Wo, align, C, gobs, gops, gibs, gips);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

extern "C" int ge_bilinear_fwd(const void* in, void* out, int N, int C, int Hi, int Wi, int Ho, int Wo,
                               int align_corners, int dtype, void* stream) {
  if (!in || !out || N < 0 || C < 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return GE_ERR_BAD_ARG;
  long ips = (long)Hi * Wi, ops = (long)Ho * Wo;
  if (dtype == GE_F32) return bilinear_fwd_launch<float>(in, out, N, C, Hi, Wi, Ho, Wo, align_corners, C * ips, ips, C * ops, ops, ge_stream(stream));
  if (dtype == GE_BF16) return bilinear_fwd_launch<bf16_t>(in, out, N, C, Hi, Wi, Ho, Wo, align_corners, C * ips, ips, C * ops, ops, ge_stream(stream));
  return GE_ERR_UNSUPPORTED;
}
extern "C" int ge_bilinear_bwd(const void* d_out, void* d_in, int N, int C, int Hi, int Wi, int Ho, int Wo,
                               int align_corners, int dtype, void* stream) {
  if (!d_out || !d_in || N < 0 || C < 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return GE_ERR_BAD_ARG;
  long ips = (long)Hi * Wi, ops = (long)Ho * Wo;
  if (dtype == GE_F32) return bilinear_bwd_launch<float>(d_out, d_in, N, C, Hi, Wi, Ho, Wo, align_corners, C * ops, ops, C * ips, ips, ge_stream(stream));
  if (dtype == GE_BF16) return bilinear_bwd_launch<bf16_t>(d_out, d_in, N, C, Hi, Wi, Ho, Wo, align_corners, C * ops, ops, C * ips, ips, ge_stream(stream));
  return GE_ERR_UNSUPPORTED;
}

// ============================================================================== ground embedding
#define GE_NSLOPE 11
#define GE_DEG2RAD 0.017453292519943295f

struct GroundPix {  // per-pixel recomputation shared by forward and backward
  float p[GE_NSLOPE];
  float deg, k, den, off, m, y;
};

__device__ __forceinline__ float tap4(const float* __restrict__ plane, int w, const Lerp& ly, const Lerp& lx) {
  float v00 = plane[(long)ly.i0 * w + lx.i0], v01 = plane[(long)ly.i0 * w + lx.i1];
  float v10 = plane[(long)ly.i1 * w + lx.i0], v11 = plane[(long)ly.i1 * w + lx.i1];
  return ly.w0 * (lx.w0 * v00 + lx.w1 * v01) + ly.w1 * (lx.w0 * v10 + lx.w1 * v11);
}

__device__ __forceinline__ void ground_pixel(const float* __restrict__ logits_lr, const float* __restrict__ y_lr,
                                             float pe, float hcam, float depth_scale, int b, int h, int w,
                                             const Lerp& ly, const Lerp& lx, float* logit_out, GroundPix& r) {
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < GE_NSLOPE; ++c) {
    float v = tap4(logits_lr + ((long)b * GE_NSLOPE + c) * h * w, w, ly, lx);
    logit_out[c] = v;
    mx = fmaxf(mx, v);
  }
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < GE_NSLOPE; ++c) { r.p[c] = expf(logit_out[c] - mx); sum += r.p[c]; }
  float deg = 0.f;
#pragma unroll
  for (int c = 0; c < GE_NSLOPE; ++c) { r.p[c] = r.p[c] / sum; deg += r.p[c] * (float)(c - 5); }
  r.deg = deg;
  r.k = tanf(deg * GE_DEG2RAD);
  float a = -hcam / (pe + 1e-8f);
  r.den = (a - r.k) + 1e-8f;
  r.off = -hcam / r.den;
  float m = r.off;                     // encoder_decoder.py:97-100, in that order
  if (m < 0.f) m = 0.f;
  if (m > depth_scale) m = 0.f;
  if (m > 0.f) m = 1.f;
  r.m = m;
  r.y = tap4(y_lr + (long)b * h * w, w, ly, lx);
}

__global__ void __launch_bounds__(256) ground_embed_fwd_k(const float* __restrict__ logits_lr, const float* __restrict__ y_lr,
                                                          const float* __restrict__ pe_raw, long pe_bs,
                                                          const float* __restrict__ height, float depth_scale,
                                                          float* __restrict__ pe_mask, float* __restrict__ logits_hr,
                                                          float* __restrict__ y_hr, uint8_t* __restrict__ valid,
                                                          int B, int h, int w, int H, int W) {
  const float sy = ge_scale(h, H, false), sx = ge_scale(w, W, false);
  const long HW = (long)H * W, total = (long)B * HW;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int b = (int)(idx / HW);
    long pix = idx - (long)b * HW;
    int Y = (int)(pix / W), X = (int)(pix - (long)Y * W);
    Lerp ly = ge_lerp(Y, h, sy, false), lx = ge_lerp(X, w, sx, false);
    float lg[GE_NSLOPE];
    GroundPix r;
    float hcam = height ? height[b] : 1.65f;
    ground_pixel(logits_lr, y_lr, pe_raw[(long)b * pe_bs + pix], hcam, depth_scale, b, h, w, ly, lx, lg, r);
#pragma unroll
    for (int c = 0; c < GE_NSLOPE; ++c) logits_hr[((long)b * GE_NSLOPE + c) * HW + pix] = lg[c];
    y_hr[idx] = r.y;
    pe_mask[idx] = (r.off * r.m) * r.y;
    valid[idx] = (uint8_t)(r.m == 1.f ? 1 : 0);
  }
}

// scratch layout (B,12,H,W): channels 0..10 = total grad wrt up-sampled logits, 11 = total grad wrt up-sampled y
__global__ void __launch_bounds__(256) ground_embed_bwd_k(const float* __restrict__ logits_lr, const float* __restrict__ y_lr,
                                                          const float* __restrict__ pe_raw, long pe_bs,
                                                          const float* __restrict__ height, float depth_scale,
                                                          const float* __restrict__ d_pe_mask, const float* __restrict__ d_logits_hr,
                                                          const float* __restrict__ d_y_hr, float* __restrict__ scratch,
                                                          int B, int h, int w, int H, int W) {
  const float sy = ge_scale(h, H, false), sx = ge_scale(w, W, false);
  const long HW = (long)H * W, total = (long)B * HW;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int b = (int)(idx / HW);
    long pix = idx - (long)b * HW;
    int Y = (int)(pix / W), X = (int)(pix - (long)Y * W);
    Lerp ly = ge_lerp(Y, h, sy, false), lx = ge_lerp(X, w, sx, false);
    float lg[GE_NSLOPE];
    GroundPix r;
    float hcam = height ? height[b] : 1.65f;
    ground_pixel(logits_lr, y_lr, pe_raw[(long)b * pe_bs + pix], hcam, depth_scale, b, h, w, ly, lx, lg, r);
    float g = d_pe_mask[idx];
    float gy = g * (r.off * r.m) + (d_y_hr ? d_y_hr[idx] : 0.f);
    // d off / d k = -h/den^2 ; dk/ddeg = (1+k^2) pi/180 ; ddeg/dlogit_c = p_c ((c-5) - deg)
    float G = (r.m != 0.f) ? g * r.y * r.m * (-hcam / (r.den * r.den)) * (1.f + r.k * r.k) * GE_DEG2RAD : 0.f;
    float* sc = scratch + (long)b * 12 * HW + pix;
#pragma unroll
    for (int c = 0; c < GE_NSLOPE; ++c) {
      float v = G * r.p[c] * ((float)(c - 5) - r.deg);
      if (d_logits_hr) v += d_logits_hr[((long)b * GE_NSLOPE + c) * HW + pix];
      sc[(long)c * HW] = v;
    }
    sc[(long)GE_NSLOPE * HW] = gy;
  }
}

extern "C" int ge_ground_embed_fwd(const float* logits_lr, const float* y_lr, const float* pe_raw, long pe_bs,
                                   const float* height, float depth_scale, float* pe_mask, float* logits_hr,
                                   float* y_hr, uint8_t* valid_mask, int B, int h, int w, int H, int W, void* stream) {
  if (!logits_lr || !y_lr || !pe_raw || !pe_mask || !logits_hr || !y_hr || !valid_mask) return GE_ERR_BAD_ARG;
  if (B < 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return GE_ERR_BAD_ARG;
  long total = (long)B * H * W;
  if (total == 0) return GE_OK;
  ground_embed_fwd_k<<<ge_blocks(total, 256, 65536), 256, 0, ge_stream(stream)>>>(
      logits_lr, y_lr, pe_raw, pe_bs, height, depth_scale, pe_mask, logits_hr, y_hr, valid_mask, B, h, w, H, W);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

extern "C" int ge_ground_embed_bwd(const float* logits_lr, const float* y_lr, const float* pe_raw, long pe_bs,
                                   const float* height, float depth_scale, const float* d_pe_mask,
                                   const float* d_logits_hr, const float* d_y_hr, float* d_logits_lr, float* d_y_lr,
                                   float* scratch, int B, int h, int w, int H, int W, void* stream) {
  if (!logits_lr || !y_lr || !pe_raw || !d_pe_mask || !d_logits_lr || !d_y_lr || !scratch) return GE_ERR_BAD_ARG;
  if (B < 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return GE_ERR_BAD_ARG;
  long total = (long)B * H * W;
  if (total == 0) return GE_OK;
  hipStream_t s = ge_stream(stream);
  ground_embed_bwd_k<<<ge_blocks(total, 256, 65536), 256, 0, s>>>(logits_lr, y_lr, pe_raw, pe_bs, height, depth_scale,
                                                                  d_pe_mask, d_logits_hr, d_y_hr, scratch, B, h, w, H, W);
  GE_LAUNCH_CHECK();
  const long HW = (long)H * W, hw = (long)h * w;
  // (12 planes per image in scratch) -> 11 logit planes + 1 y plane
  int e = bilinear_bwd_launch<float>(scratch, d_logits_lr, B, GE_NSLOPE, h, w, H, W, 0, 12 * HW, HW, GE_NSLOPE * hw, hw, s);
  if (e) return e;
  return bilinear_bwd_launch<float>(scratch + GE_NSLOPE * HW, d_y_lr, B, 1, h, w, H, W, 0, 12 * HW, HW, hw, hw, s);
}

__global__ void __launch_bounds__(256) ground_vanilla_fwd_k(const float* __restrict__ y_lr, const float* __restrict__ pe_norm,
                                                            long pe_bs, float gain, float* __restrict__ pe_mask,
                                                            float* __restrict__ y_hr, int B, int h, int w, int H, int W) {
  const float sy = ge_scale(h, H, false), sx = ge_scale(w, W, false);
  const long HW = (long)H * W, total = (long)B * HW;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int b = (int)(idx / HW);
    long pix = idx - (long)b * HW;
    int Y = (int)(pix / W), X = (int)(pix - (long)Y * W);
    Lerp ly = ge_lerp(Y, h, sy, false), lx = ge_lerp(X, w, sx, false);
    float y = tap4(y_lr + (long)b * h * w, w, ly, lx);
    y_hr[idx] = y;
    pe_mask[idx] = pe_norm[(long)b * pe_bs + pix] * y * gain;  // x_pe*y * 200 (encoder_decoder.py:122)
  }
}
__global__ void __launch_bounds__(256) ground_vanilla_bwd_k(const float* __restrict__ pe_norm, long pe_bs, float gain,
                                                            const float* __restrict__ d_pe_mask, const float* __restrict__ d_y_hr,
                                                            float* __restrict__ scratch, int B, long HW) {
  const long total = (long)B * HW;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int b = (int)(idx / HW);
    long pix = idx - (long)b * HW;
    scratch[idx] = d_pe_mask[idx] * pe_norm[(long)b * pe_bs + pix] * gain + (d_y_hr ? d_y_hr[idx] : 0.f);
  }
}
extern "C" int ge_ground_vanilla_fwd(const float* y_lr, const float* pe_norm, long pe_bs, float gain, float* pe_mask,
                                     float* y_hr, int B, int h, int w, int H, int W, void* stream) {
  if (!y_lr || !pe_norm || !pe_mask || !y_hr || B < 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return GE_ERR_BAD_ARG;
  long total = (long)B * H * W;
  if (total == 0) return GE_OK;
  ground_vanilla_fwd_k<<<ge_blocks(total, 256, 65536), 256, 0, ge_stream(stream)>>>(y_lr, pe_norm, pe_bs, gain, pe_mask, y_hr, B, h, w, H, W);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
extern "C" int ge_ground_vanilla_bwd(const float* pe_norm, long pe_bs, float gain, const float* d_pe_mask,
                                     const float* d_y_hr, float* d_y_lr, float* scratch, int B, int h, int w, int H,
                                     int W, void* stream) {
  if (!pe_norm || !d_pe_mask || !d_y_lr || !scratch || B < 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return GE_ERR_BAD_ARG;
  long total = (long)B * H * W;
  if (total == 0) return GE_OK;
  hipStream_t s = ge_stream(stream);
  ground_vanilla_bwd_k<<<ge_blocks(total, 256, 65536), 256, 0, s>>>(pe_norm, pe_bs, gain, d_pe_mask, d_y_hr, scratch, B, (long)H * W);
  GE_LAUNCH_CHECK();
  return bilinear_bwd_launch<float>(scratch, d_y_lr, B, 1, h, w, H, W, 0, (long)H * W, (long)H * W, (long)h * w, (long)h * w, s);
}

// ================================================================================== depth fusion
__global__ void __launch_bounds__(256) depth_fuse_fwd_k(const float* __restrict__ c, const float* __restrict__ pe_mask,
                                                        const float* __restrict__ y_hr, float min_depth,
                                                        float* __restrict__ out, float* __restrict__ y_ds,
                                                        int B, int h, int w, int H, int W) {
  const float sy = ge_scale(H, h, true), sx = ge_scale(W, w, true);
  const long hw = (long)h * w, total = (long)B * hw;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int b = (int)(idx / hw);
    long pix = idx - (long)b * hw;
    int y = (int)(pix / w), x = (int)(pix - (long)y * w);
    Lerp ly = ge_lerp(y, H, sy, true), lx = ge_lerp(x, W, sx, true);
    float pe = tap4(pe_mask + (long)b * H * W, W, ly, lx);
    float yy = tap4(y_hr + (long)b * H * W, W, ly, lx);
    float d = fmaxf(c[idx], 0.f);
    y_ds[idx] = yy;
    out[idx] = ((d * (1.f - yy)) + pe) + min_depth;  // decode_head.py:506
  }
}
__global__ void __launch_bounds__(256) depth_fuse_bwd_k(const float* __restrict__ c, const float* __restrict__ y_ds,
                                                        const float* __restrict__ d_out, float* __restrict__ d_c,
                                                        float* __restrict__ d_yds, long total) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    float g = d_out[idx], cv = c[idx];
    d_c[idx] = cv > 0.f ? g * (1.f - y_ds[idx]) : 0.f;
    d_yds[idx] = -g * fmaxf(cv, 0.f);
  }
}
extern "C" int ge_depth_fuse_fwd(const float* c, const float* pe_mask, const float* y_hr, float min_depth, float* out,
                                 float* y_ds, int B, int h, int w, int H, int W, void* stream) {
  if (!c || !pe_mask || !y_hr || !out || !y_ds || B < 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return GE_ERR_BAD_ARG;
  long total = (long)B * h * w;
  if (total == 0) return GE_OK;
  depth_fuse_fwd_k<<<ge_blocks(total, 256, 65536), 256, 0, ge_stream(stream)>>>(c, pe_mask, y_hr, min_depth, out, y_ds, B, h, w, H, W);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
extern "C" int ge_depth_fuse_bwd(const float* c, const float* y_ds, const float* d_out, float* d_c, float* d_pe_mask,
                                 float* d_y_hr, float* scratch, int B, int h, int w, int H, int W, void* stream) {
  if (!c || !y_ds || !d_out || !d_c || !d_pe_mask || !d_y_hr || !scratch) return GE_ERR_BAD_ARG;
  if (B < 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return GE_ERR_BAD_ARG;
  long total = (long)B * h * w;
  if (total == 0) return GE_OK;
  hipStream_t s = ge_stream(stream);
  depth_fuse_bwd_k<<<ge_blocks(total, 256, 65536), 256, 0, s>>>(c, y_ds, d_out, d_c, scratch, total);
  GE_LAUNCH_CHECK();
  // transposes of the align_corners=True (H,W)->(h,w) resize: d_pe_ds == d_out
  const long hw = (long)h * w, HW = (long)H * W;
  int e = bilinear_bwd_launch<float>(d_out, d_pe_mask, B, 1, H, W, h, w, 1, hw, hw, HW, HW, s);
  if (e) return e;
  return bilinear_bwd_launch<float>(scratch, d_y_hr, B, 1, H, W, h, w, 1, hw, hw, HW, HW, s);
}

// ===================================================================== offline ground-plane maps
__global__ void __launch_bounds__(256) ground_plane_k(double r0, double r1, double r2, double num, double* __restrict__ pe64,
                                                      float* __restrict__ pe32, int H, int W) {
  const long total = (long)H * W;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int v = (int)(idx / W), u = (int)(idx - (long)v * W);
    // numpy evaluates (r0*u + r1*v) + r2 with separately rounded products (contraction is off in this file)
    const double t0 = r0 * (double)u, t1 = r1 * (double)v;
    const double den = (t0 + t1) + r2;
    const double pe = num / den;
    if (pe64) pe64[idx] = pe;
    if (pe32) pe32[idx] = (float)pe;
  }
}
extern "C" int ge_ground_plane(const double* rinv_row2, double num, double* pe_f64, float* pe_f32, int H, int W, void* stream) {
  if (!rinv_row2 || (!pe_f64 && !pe_f32) || H <= 0 || W <= 0) return GE_ERR_BAD_ARG;
  ground_plane_k<<<ge_blocks((long)H * W, 256, 65536), 256, 0, ge_stream(stream)>>>(rinv_row2[0], rinv_row2[1], rinv_row2[2], num, pe_f64, pe_f32, H, W);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

__global__ void __launch_bounds__(256) slope_class_k(const double* __restrict__ gt, const float* __restrict__ pe, double hcam,
                                                     int mode, int16_t* __restrict__ cls, long total) {
  const double RAD2DEG = 57.29577951308232;  // 180/pi as numpy's rad2deg uses
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    double g = gt[idx];
    float a = (float)(-hcam) / pe[idx];         // float32 scalar / float32 array (preprocess_data_kitti.py:60)
    double b = hcam / g;                        // float64
    double k = b + (double)a;
    double deg = atan(k) * RAD2DEG;
    double r = (mode == 0) ? rint(deg) : trunc(deg);   // np.around = round-half-even
    if (r > 5.0) r = 5.0;
    if (r < -5.0) r = -5.0;
    int16_t out = (int16_t)r;
    if (!(r == r)) out = 0;                      // NaN -> astype(int) is undefined; never produced for gt != 0
    if (g == 0.0) out = 255;
    cls[idx] = out;
  }
}
extern "C" int ge_slope_class(const double* gt, const float* pe, double cam_height, int mode, int16_t* cls, int H, int W, void* stream) {
  if (!gt || !pe || !cls || H <= 0 || W <= 0 || (mode != 0 && mode != 1)) return GE_ERR_BAD_ARG;
  slope_class_k<<<ge_blocks((long)H * W, 256, 65536), 256, 0, ge_stream(stream)>>>(gt, pe, cam_height, mode, cls, (long)H * W);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// DDAD variant (tools/preprocess_data_ddad.py:47-51,68-78): gt is the float32 depth of the .npz, pe stays float64,
// a = -h/pe in float64, b = h/gt in FLOAT32 (Python scalar / float32 array), k = b + a in float64, truncation.
__global__ void __launch_bounds__(256) slope_class_ddad_k(const float* __restrict__ gt, const double* __restrict__ pe, double hcam,
                                                          int16_t* __restrict__ cls, long total) {
  const double RAD2DEG = 57.29577951308232;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const float g = gt[idx];
    const double a = (-hcam) / pe[idx];
    const float b = (float)hcam / g;
    const double k = (double)b + a;
    double r = trunc(atan(k) * RAD2DEG);
    if (r > 5.0) r = 5.0;
    if (r < -5.0) r = -5.0;
    int16_t out = (int16_t)r;
    if (!(r == r)) out = -5;           // NaN: the reference's astype(np.int64) yields INT64_MIN on x86, which its clamp turns into -5
    if (g == 0.f) out = 255;
    cls[idx] = out;
  }
}
extern "C" int ge_slope_class_ddad(const float* gt, const double* pe, double cam_height, int16_t* cls, int H, int W, void* stream) {
  if (!gt || !pe || !cls || H <= 0 || W <= 0) return GE_ERR_BAD_ARG;
  slope_class_ddad_k<<<ge_blocks((long)H * W, 256, 65536), 256, 0, ge_stream(stream)>>>(gt, pe, cam_height, cls, (long)H * W);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

__global__ void __launch_bounds__(256) pe_channels_k(const float* __restrict__ raw, float* __restrict__ norm, float depth_scale, long n) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    float v = raw[idx];
    if (v > 200.f) v = 0.f;            // loading.py:400
    if (v < 0.f) v = 0.f;              // loading.py:401
    if (v > 0.f) v = v / depth_scale;  // transforms.py:44
    norm[idx] = v;
  }
}
extern "C" int ge_pe_channels(const float* raw, float* norm, float depth_scale, long n, void* stream) {
  if (!raw || !norm || n < 0) return GE_ERR_BAD_ARG;
  if (n == 0) return GE_OK;
  pe_channels_k<<<ge_blocks(n, 256, 65536), 256, 0, ge_stream(stream)>>>(raw, norm, depth_scale, n);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// ========================================================================================= SiLog
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__global__ void __launch_bounds__(256) silog_stats_k(const float* __restrict__ pred, const float* __restrict__ gt, float eps,
                                                     double* __restrict__ stats, long n) {
  double cnt = 0, s1 = 0, s2 = 0;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    float t = gt[idx];
    if (t > 0.f) {
      float g = logf(pred[idx] + eps) - logf(t + eps);
      cnt += 1.0; s1 += (double)g; s2 += (double)g * (double)g;
    }
  }
  __shared__ double sm[3][4];
  cnt = wave_sum_d(cnt); s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
  int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  if (ln == 0) { sm[0][wv] = cnt; sm[1][wv] = s1; sm[2][wv] = s2; }
  __syncthreads();
  if (threadIdx.x < 3) {
    double v = sm[threadIdx.x][0] + sm[threadIdx.x][1] + sm[threadIdx.x][2] + sm[threadIdx.x][3];
    atomicAdd(&stats[threadIdx.x], v);
  }
}
__global__ void __launch_bounds__(256) silog_bwd_k(const float* __restrict__ pred, const float* __restrict__ gt, float eps,
                                                   const float* __restrict__ ca, const float* __restrict__ cb,
                                                   float* __restrict__ d_pred, long n) {
  const float a = ca[0], b = cb[0];
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    float t = gt[idx], p = pred[idx], o = 0.f;
    if (t > 0.f) {
      float g = logf(p + eps) - logf(t + eps);
      o = (a * g + b) / (p + eps);
    }
    d_pred[idx] = o;
  }
}
extern "C" int ge_silog_stats(const float* pred, const float* gt, float eps, double* stats, long n, void* stream) {
  if (!pred || !gt || !stats || n < 0) return GE_ERR_BAD_ARG;
  if (n == 0) return GE_OK;
  silog_stats_k<<<ge_blocks(n, 256 * 8, 2048), 256, 0, ge_stream(stream)>>>(pred, gt, eps, stats, n);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
extern "C" int ge_silog_bwd(const float* pred, const float* gt, float eps, const float* coef_a, const float* coef_b,
                            float* d_pred, long n, void* stream) {
  if (!pred || !gt || !coef_a || !coef_b || !d_pred || n < 0) return GE_ERR_BAD_ARG;
  if (n == 0) return GE_OK;
  silog_bwd_k<<<ge_blocks(n, 256, 65536), 256, 0, ge_stream(stream)>>>(pred, gt, eps, coef_a, coef_b, d_pred, n);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// ================================================================================== fused AdamW
__global__ void __launch_bounds__(256) sumsq_k(const float* __restrict__ x, long n, double* __restrict__ out) {
  double s = 0;
  const long n4 = n >> 2;
  const float4* x4 = (const float4*)x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 v = x4[i];
    s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { float v = x[(n4 << 2) + threadIdx.x]; s += (double)v * v; }
  __shared__ double sm[4];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, sm[0] + sm[1] + sm[2] + sm[3]);
}
extern "C" int ge_sumsq(const float* x, long n, double* out, void* stream) {
  if (!x || !out || n < 0 || ((uintptr_t)x & 15)) return GE_ERR_BAD_ARG;
  if (n == 0) return GE_OK;
  sumsq_k<<<ge_blocks(n, 256 * 16, 2048), 256, 0, ge_stream(stream)>>>(x, n, out);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// hyper = {lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, max_norm, 1 - beta1, 1 - beta2}; the last two are rounded
// from the host's double like torch.optim's `value=1 - beta2` (1.f - 0.999f is 1.3e-5 off 0.001f)
// SHADOW: also write the updated parameter rounded to bf16 into a second arena — the low-precision copy the autocast forward reads,
// so that no per-tensor cast kernel runs in the next step (mmrt/optim.py)
template <bool SHADOW>
__global__ void __launch_bounds__(256) adamw_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, const uint8_t* __restrict__ wdm,
                                               const float* __restrict__ hyper, const double* __restrict__ gnorm_sq, long n,
                                               bf16_t* __restrict__ shadow) {
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], bc1 = hyper[5], bc2 = hyper[6],
              max_norm = hyper[7], omb1 = hyper[8], omb2 = hyper[9];
  float clip = 1.f;
  if (max_norm > 0.f && gnorm_sq) {
    float total = (float)sqrt(gnorm_sq[0]);
    clip = fminf(max_norm / (total + 1e-6f), 1.f);   // torch.nn.utils.clip_grad_norm_
  }
  const float step = lr / bc1, rs2 = 1.f / sqrtf(bc2);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] * clip, pi = p[i], mi = m[i], vi = v[i];
    if (wdm[i]) pi *= 1.f - lr * wd;
    mi = b1 * mi + omb1 * gi;
    vi = b2 * vi + omb2 * gi * gi;
    float denom = sqrtf(vi) * rs2 + eps;
    pi -= step * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (SHADOW) shadow[i] = f2bf(pi);
  }
}
extern "C" int ge_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const uint8_t* wd_mask,
                             const float* hyper, const double* gnorm_sq, long n, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !wd_mask || !hyper || n < 0) return GE_ERR_BAD_ARG;
  if (n == 0) return GE_OK;
  adamw_k<false><<<ge_blocks(n, 256 * 4, 8192), 256, 0, ge_stream(stream)>>>(param, grad, exp_avg, exp_avg_sq, wd_mask, hyper, gnorm_sq, n,
                                                                             nullptr);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
extern "C" int ge_adamw_step_shadow(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const uint8_t* wd_mask,
                                    const float* hyper, const double* gnorm_sq, long n, void* shadow_bf16, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !wd_mask || !hyper || !shadow_bf16 || n < 0) return GE_ERR_BAD_ARG;
  if (n == 0) return GE_OK;
  adamw_k<true><<<ge_blocks(n, 256 * 4, 8192), 256, 0, ge_stream(stream)>>>(param, grad, exp_avg, exp_avg_sq, wd_mask, hyper, gnorm_sq, n,
                                                                            (bf16_t*)shadow_bf16);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// ============================================================================ bias + activation (NCHW)
// y = act(x + b[c]) in place, act = leaky-relu with `slope` (0 -> ReLU, 1 -> identity).  Replaces the separate
// broadcast bias add ATen issues after a MIOpen convolution plus the activation kernel
// (mmcv ConvModule without norm: decode_heads/densedepth_head.py:14-27, necks/pemask_neck.py:36-42).
// Backward: dx = dy * (y > 0 ? 1 : slope) and db[c] += sum dx in the same pass (block reduction + one atomic per block).
// grid: (chunks of one plane, N*C planes); HW % V8<T>::N == 0 (vector path) is checked by the launcher
template <typename T, bool VEC>
__global__ void __launch_bounds__(256) bias_act_fwd_k(T* __restrict__ x, const float* __restrict__ bias, int C, long HW, float slope) {
  const long plane = blockIdx.y;
  const float b = bias[plane % C];
  T* p = x + plane * HW;
  constexpr int VN = V8<T>::N;
  if (VEC) {
    const long nv = HW / VN;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
      float v[VN];
      V8<T>::ld(p + i * VN, v);
#pragma unroll
      for (int k = 0; k < VN; ++k) { const float t = v[k] + b; v[k] = t > 0.f ? t : t * slope; }
      V8<T>::st(p + i * VN, v);
    }
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) {
      const float t = Io<T>::ld(p + i) + b;
      Io<T>::st(p + i, t > 0.f ? t : t * slope);
    }
  }
}
template <typename T, bool VEC>
__global__ void __launch_bounds__(256) bias_act_bwd_k(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx,
                                                      float* __restrict__ dbias, int C, long HW, float slope) {
  const long plane = blockIdx.y;
  const T* gp = dy + plane * HW;
  const T* yp = y + plane * HW;
  T* dp = dx + plane * HW;
  constexpr int VN = V8<T>::N;
  float acc = 0.f;
  if (VEC) {
    const long nv = HW / VN;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
      float g[VN], v[VN];
      V8<T>::ld(gp + i * VN, g);
      V8<T>::ld(yp + i * VN, v);
#pragma unroll
      for (int k = 0; k < VN; ++k) { g[k] = v[k] > 0.f ? g[k] : g[k] * slope; acc += g[k]; }
      V8<T>::st(dp + i * VN, g);
    }
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) {
      float g = Io<T>::ld(gp + i);
      g = Io<T>::ld(yp + i) > 0.f ? g : g * slope;
      acc += g;
      Io<T>::st(dp + i, g);
    }
  }
  __shared__ float sm[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&dbias[plane % C], sm[0] + sm[1] + sm[2] + sm[3]);
}

template <typename T>
static int bias_act_launch(bool fwd, void* x_or_dx, const void* dy, const void* y, const float* bias, float* dbias, int N, int C,
                           long HW, float slope, hipStream_t s) {
  const bool vec = (HW % V8<T>::N) == 0 && (((uintptr_t)x_or_dx | (uintptr_t)dy | (uintptr_t)y) & 15) == 0;
  const long per_plane = vec ? HW / V8<T>::N : HW;
  unsigned gx = (unsigned)((per_plane + 256 * 4 - 1) / (256 * 4));
  if (gx < 1) gx = 1;
  if (gx > 64) gx = 64;
  dim3 grid(gx, (unsigned)(N * C));
  if (fwd) {
    if (vec) bias_act_fwd_k<T, true><<<grid, 256, 0, s>>>((T*)x_or_dx, bias, C, HW, slope);
    else bias_act_fwd_k<T, false><<<grid, 256, 0, s>>>((T*)x_or_dx, bias, C, HW, slope);
  } else {
    if (vec) bias_act_bwd_k<T, true><<<grid, 256, 0, s>>>((const T*)dy, (const T*)y, (T*)x_or_dx, dbias, C, HW, slope);
    else bias_act_bwd_k<T, false><<<grid, 256, 0, s>>>((const T*)dy, (const T*)y, (T*)x_or_dx, dbias, C, HW, slope);
  }
  GE_LAUNCH_CHECK();
  return GE_OK;
}
extern "C" int ge_bias_act_fwd(void* x, const float* bias, int N, int C, long HW, float slope, int dtype, void* stream) {
  if (!x || !bias || N < 0 || C <= 0 || HW < 0 || (long)N * C > 2147483647L) return GE_ERR_BAD_ARG;
  if ((long)N * C * HW == 0) return GE_OK;
  if (dtype == GE_F32) return bias_act_launch<float>(true, x, nullptr, nullptr, bias, nullptr, N, C, HW, slope, ge_stream(stream));
  if (dtype == GE_BF16) return bias_act_launch<bf16_t>(true, x, nullptr, nullptr, bias, nullptr, N, C, HW, slope, ge_stream(stream));
  return GE_ERR_UNSUPPORTED;
}
extern "C" int ge_bias_act_bwd(const void* dy, const void* y, void* dx, float* dbias, int N, int C, long HW, float slope,
                               int dtype, void* stream) {
  if (!dy || !y || !dx || !dbias || N < 0 || C <= 0 || HW < 0 || (long)N * C > 2147483647L) return GE_ERR_BAD_ARG;
  if ((long)N * C * HW == 0) return GE_OK;
  if (dtype == GE_F32) return bias_act_launch<float>(false, dx, dy, y, nullptr, dbias, N, C, HW, slope, ge_stream(stream));
  if (dtype == GE_BF16) return bias_act_launch<bf16_t>(false, dx, dy, y, nullptr, dbias, N, C, HW, slope, ge_stream(stream));
  return GE_ERR_UNSUPPORTED;
}

extern "C" int ge_abi_version(void) { return 7; }
