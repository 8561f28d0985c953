// Device-side data pipeline of the ground-embedding training samples: the per-sample geometric and
// photometric transforms of the reference's KITTI train pipeline (configs/depthformer/depthformer_v.py:13-28 ->
// depth/datasets/pipelines/loading.py:366-403,490-526 and transforms.py:13-62 Normalize, :65-111 Padding, :150-205 KBCrop,
// :209-297 RandomRotate, :300-354 RandomFlip, :357-418 RandomCrop, :421-482 ColorAug, :485-733 Resize), applied on the GPU
// to planar float32 maps, consistently for the five image channels (B, G, R, filtered ground depth, raw ground depth), the
// sparse depth map and the slope-class map.  The loader ships the uint8 image, the uint16 depth and the class map; the two
// ground-depth channels come from the calibration (ge_ground_plane + the filter below), never from disk.
//
// Each kernel restates the sampling rule of the host transform it replaces, op for op in fp32 (contraction off), so the
// device pipeline reproduces the host pipeline: bit-exact for the index-only steps (crop, pad, flip, nearest), to fp32
// rounding for the bilinear ones.  Pure streaming / gather work: HBM-bound, no MFMA.
#pragma clang fp contract(off)
#include "common.h"

// ---------------------------------------------------------------------------------------------- load + KB crop
// dst (5, Hc, Wc) planar f32 from the HWC uint8 BGR image and the (H, W) ground-depth map, window (top, left):
// channels 0-2 = B, G, R as float; 3 = pe with values > 200 or < 0 zeroed (loading.py:397-401); 4 = raw pe.
__global__ void __launch_bounds__(256) aug_load_k(const uint8_t* __restrict__ bgr, const float* __restrict__ pe, float* __restrict__ dst,
                                                  int H, int W, int top, int left, int Hc, int Wc, float pe_max) {
  const long n = (long)Hc * Wc;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int y = (int)(i / Wc), x = (int)(i - (long)y * Wc);
    const long s = (long)(y + top) * W + (x + left);
    const uint8_t* p = bgr + s * 3;
    dst[i] = (float)p[0]; dst[n + i] = (float)p[1]; dst[2 * n + i] = (float)p[2];
    const float raw = pe[s];
    float f = raw;
    if (f > pe_max) f = 0.f;
    if (f < 0.f) f = 0.f;
    dst[3 * n + i] = f; dst[4 * n + i] = raw;
  }
}
extern "C" int ge_aug_load(const uint8_t* bgr_hwc, const float* pe, float* dst, int H, int W, int top, int left, int Hc, int Wc,
                           float pe_max, void* stream) {
  if (!bgr_hwc || !pe || !dst || H <= 0 || W <= 0 || Hc <= 0 || Wc <= 0 || top < 0 || left < 0 || top + Hc > H || left + Wc > W)
    return GE_ERR_BAD_ARG;
  aug_load_k<<<ge_blocks((long)Hc * Wc, 256, 65536), 256, 0, ge_stream(stream)>>>(bgr_hwc, pe, dst, H, W, top, left, Hc, Wc, pe_max);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// uint16 depth PNG -> metres (DepthLoadAnnotations: float32(png) / depth_scale), with the same window
__global__ void __launch_bounds__(256) aug_depth_k(const uint16_t* __restrict__ png, float* __restrict__ dst, int W, int top, int left,
                                                   int Hc, int Wc, float depth_scale) {
  const long n = (long)Hc * Wc;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int y = (int)(i / Wc), x = (int)(i - (long)y * Wc);
    dst[i] = (float)png[(long)(y + top) * W + (x + left)] / depth_scale;
  }
}
extern "C" int ge_aug_depth(const uint16_t* png, float* dst, int H, int W, int top, int left, int Hc, int Wc, float depth_scale,
                            void* stream) {
  if (!png || !dst || Hc <= 0 || Wc <= 0 || top < 0 || left < 0 || top + Hc > H || left + Wc > W) return GE_ERR_BAD_ARG;
  aug_depth_k<<<ge_blocks((long)Hc * Wc, 256, 65536), 256, 0, ge_stream(stream)>>>(png, dst, W, top, left, Hc, Wc, depth_scale);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// ---------------------------------------------------------------------------------------------- resize
// mode 1: bilinear, half-pixel centres, edge replication, no antialiasing = cv2.INTER_LINEAR = F.interpolate(align_corners=
// False) (imageops.imresize); mode 0: nearest, src = min(floor(dst * in / out), in - 1) evaluated in float64 like the host.
__global__ void __launch_bounds__(256) aug_resize_k(const float* __restrict__ src, float* __restrict__ dst, int C, int Hs, int Ws, int Hd,
                                                    int Wd, int mode) {
  const long n = (long)Hd * Wd;
  const float sy = (float)Hs / (float)Hd, sx = (float)Ws / (float)Wd;
  const double dy = (double)Hs / (double)Hd, dx = (double)Ws / (double)Wd;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int y = (int)(i / Wd), x = (int)(i - (long)y * Wd);
    if (mode == 0) {
      int ys = (int)floor((double)y * dy), xs = (int)floor((double)x * dx);
      ys = min(ys, Hs - 1); xs = min(xs, Ws - 1);
      for (int c = 0; c < C; ++c) dst[(long)c * n + i] = src[((long)c * Hs + ys) * Ws + xs];
    } else {
      const Lerp ly = ge_lerp(y, Hs, sy, false), lx = ge_lerp(x, Ws, sx, false);
      for (int c = 0; c < C; ++c) {
        const float* s = src + (long)c * Hs * Ws;
        const float v00 = s[(long)ly.i0 * Ws + lx.i0], v01 = s[(long)ly.i0 * Ws + lx.i1];
        const float v10 = s[(long)ly.i1 * Ws + lx.i0], v11 = s[(long)ly.i1 * Ws + lx.i1];
        // ATen's upsample_bilinear2d: w0y * (w0x * v00 + w1x * v01) + w1y * (w0x * v10 + w1x * v11)
        dst[(long)c * n + i] = ly.w0 * (lx.w0 * v00 + lx.w1 * v01) + ly.w1 * (lx.w0 * v10 + lx.w1 * v11);
      }
    }
  }
}
extern "C" int ge_aug_resize(const float* src, float* dst, int C, int Hs, int Ws, int Hd, int Wd, int mode, void* stream) {
  if (!src || !dst || C <= 0 || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0 || (mode != 0 && mode != 1)) return GE_ERR_BAD_ARG;
  aug_resize_k<<<ge_blocks((long)Hd * Wd, 256, 65536), 256, 0, ge_stream(stream)>>>(src, dst, C, Hs, Ws, Hd, Wd, mode);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// ---------------------------------------------------------------------------------------------- rotate
// imageops.imrotate: inverse-mapped affine warp, output size = input size, constant border.  The host evaluates the source
// coordinates in float32 (matrix from float64), normalises them to F.grid_sample's align_corners=True grid and lets
// grid_sample un-normalise them again; the same chain of fp32 operations is replayed here so that the fractional weights
// (and the nearest-mode rounding, nearbyint) agree.  inv = {a00, a01, off0, a10, a11, off1} in float32.
struct AugAffine { float a00, a01, o0, a10, a11, o1; };
__global__ void __launch_bounds__(256) aug_rotate_k(const float* __restrict__ src, float* __restrict__ dst, int C, int H, int W, AugAffine m,
                                                    float border, int mode) {
  const long n = (long)H * W;
  const float nx = (float)(2.0 / (double)(W - 1)), ny = (float)(2.0 / (double)(H - 1));
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int y = (int)(i / W), x = (int)(i - (long)y * W);
    const float xf = (float)x, yf = (float)y;
    const float sx = (m.a00 * xf + m.a01 * yf) + m.o0, sy = (m.a10 * xf + m.a11 * yf) + m.o1;
    const float gx = sx * nx - 1.0f, gy = sy * ny - 1.0f;
    const float ux = ((gx + 1.f) / 2.f) * (float)(W - 1), uy = ((gy + 1.f) / 2.f) * (float)(H - 1);    // grid_sampler_unnormalize
    if (mode == 0) {
      const float rx = nearbyintf(ux), ry = nearbyintf(uy);
      const bool in = rx >= 0.f && rx <= (float)(W - 1) && ry >= 0.f && ry <= (float)(H - 1);
      const long o = in ? (long)ry * W + (long)rx : 0;
      for (int c = 0; c < C; ++c) dst[(long)c * n + i] = in ? (src[(long)c * n + o] - border) + border : (0.f + border);
    } else {
      const float x0f = floorf(ux), y0f = floorf(uy);
      const int x0 = (int)x0f, y0 = (int)y0f;
      // ATen's vectorised CPU grid_sampler_2d: w = x - floor(x), e = 1 - w, n = y - floor(y), s = 1 - n
      const float w = ux - x0f, e = 1.f - w, nn = uy - y0f, ss = 1.f - nn;
      const float nw = e * ss, ne = w * ss, sw = e * nn, se = w * nn;
      const bool kx0 = x0 >= 0 && x0 < W, kx1 = x0 + 1 >= 0 && x0 + 1 < W, ky0 = y0 >= 0 && y0 < H, ky1 = y0 + 1 >= 0 && y0 + 1 < H;
      for (int c = 0; c < C; ++c) {
        const float* s = src + (long)c * n;
        float acc = 0.f;
        if (ky0 && kx0) acc += (s[(long)y0 * W + x0] - border) * nw;
        if (ky0 && kx1) acc += (s[(long)y0 * W + x0 + 1] - border) * ne;
        if (ky1 && kx0) acc += (s[(long)(y0 + 1) * W + x0] - border) * sw;
        if (ky1 && kx1) acc += (s[(long)(y0 + 1) * W + x0 + 1] - border) * se;
        dst[(long)c * n + i] = acc + border;
      }
    }
  }
}
extern "C" int ge_aug_rotate(const float* src, float* dst, int C, int H, int W, const float* inv6, float border, int mode, void* stream) {
  if (!src || !dst || !inv6 || C <= 0 || H < 2 || W < 2 || (mode != 0 && mode != 1)) return GE_ERR_BAD_ARG;
  AugAffine m{inv6[0], inv6[1], inv6[2], inv6[3], inv6[4], inv6[5]};
  aug_rotate_k<<<ge_blocks((long)H * W, 256, 65536), 256, 0, ge_stream(stream)>>>(src, dst, C, H, W, m, border, mode);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// ---------------------------------------------------------------------------------------------- window copy
// dst[c, y, x] = src[c, y + oy, flip ? Ws - 1 - (x + ox) : x + ox] where that lies inside the source, else `fill`:
// Padding (negative offsets, zero / 255 canvas), RandomFlip (the flip is applied to the SOURCE, then the window is cut) and
// RandomCrop in one index-only pass.
__global__ void __launch_bounds__(256) aug_window_k(const float* __restrict__ src, float* __restrict__ dst, int C, int Hs, int Ws, int Hd,
                                                    int Wd, int oy, int ox, int flip, float fill) {
  const long n = (long)Hd * Wd;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int y = (int)(i / Wd), x = (int)(i - (long)y * Wd);
    const int ys = y + oy;
    int xs = x + ox;
    const bool in = ys >= 0 && ys < Hs && xs >= 0 && xs < Ws;
    if (flip) xs = Ws - 1 - xs;
    for (int c = 0; c < C; ++c) dst[(long)c * n + i] = in ? src[((long)c * Hs + ys) * Ws + xs] : fill;
  }
}
extern "C" int ge_aug_window(const float* src, float* dst, int C, int Hs, int Ws, int Hd, int Wd, int oy, int ox, int flip, float fill,
                             void* stream) {
  if (!src || !dst || C <= 0 || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0) return GE_ERR_BAD_ARG;
  aug_window_k<<<ge_blocks((long)Hd * Wd, 256, 65536), 256, 0, ge_stream(stream)>>>(src, dst, C, Hs, Ws, Hd, Wd, oy, ox, flip, fill);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// ---------------------------------------------------------------------------------------------- colour + normalise
// ColorAug (transforms.py:421-482) on channels 0-2 in the 0..255 range: x ** gamma, * brightness, * colour[c], clip; then
// Normalize (:13-62): truncate to uint8, BGR -> RGB, (x - mean) * (1 / std) in float64 rounded to float32; channel 3:
// positive values / depth_scale; channel 4 untouched.  src (5, H, W) BGR order -> dst (5, H, W) RGB order.
struct AugColor { float gamma, brightness; double col[3]; int on; double mean[3], stdinv[3]; float depth_scale; int to_rgb; };
__global__ void __launch_bounds__(256) aug_color_k(const float* __restrict__ src, float* __restrict__ dst, long n, AugColor p) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v[3] = {src[i], src[n + i], src[2 * n + i]};
    if (p.on) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        // numpy: float32 array ** Python float -> float32 power (evaluated here in double and rounded once); * Python float
        // brightness in float32; * the float64 colour array in FLOAT64, clip, stored back into the float32 image
        float a = (float)pow((double)v[c], (double)p.gamma);
        a = a * p.brightness;
        const double d = fmin(fmax((double)a * p.col[c], 0.0), 255.0);
        v[c] = (float)d;
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float t = truncf(v[c]);                                   // astype(np.uint8) of a value in [0, 255]
      const int oc = p.to_rgb ? 2 - c : c;
      dst[(long)oc * n + i] = (float)(((double)t - p.mean[oc]) * p.stdinv[oc]);
    }
    float pe = src[3 * n + i];
    if (pe > 0.f) pe = pe / p.depth_scale;
    dst[3 * n + i] = pe;
    dst[4 * n + i] = src[4 * n + i];
  }
}
extern "C" int ge_aug_color_normalize(const float* src, float* dst, int H, int W, int color_on, float gamma, float brightness,
                                      const double* colors3, const double* mean3, const double* std3, float depth_scale, int to_rgb,
                                      void* stream) {
  if (!src || !dst || !mean3 || !std3 || H <= 0 || W <= 0 || (color_on && !colors3)) return GE_ERR_BAD_ARG;
  AugColor p;
  p.on = color_on; p.gamma = gamma; p.brightness = brightness;
  for (int c = 0; c < 3; ++c) { p.col[c] = color_on ? colors3[c] : 1.0; p.mean[c] = mean3[c]; p.stdinv[c] = 1.0 / std3[c]; }
  p.depth_scale = depth_scale; p.to_rgb = to_rgb;
  aug_color_k<<<ge_blocks((long)H * W, 256, 65536), 256, 0, ge_stream(stream)>>>(src, dst, (long)H * W, p);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// ---------------------------------------------------------------------------------------------- DDAD front end (DDADResize)
// depth/datasets/pipelines/transforms.py:735-783: the 1216 x 1936 frame is brought to 384 x 640 BEFORE the shared augmentation chain:
// colour by cv2.INTER_AREA (pixel-area averaging, uint8 in / uint8 out), the two ground-depth channels by cv2.INTER_NEAREST, and the
// sparse LiDAR depth / slope classes by re-projecting every valid pixel to int(coord * scale) — later pixels (row-major) overwrite
// earlier ones, nothing is interpolated.
//
// Area filter: destination cell j covers the source interval [j s, (j + 1) s), s = in / out; source pixel i weighs by its overlap,
// weights normalised per cell, float64 (imageops._area_weights); rows first, then columns; the uint8 result is rint + clip.
__device__ __forceinline__ void area_span(int j, int n_in, int n_out, int& i0, int& i1, double& lo, double& hi, double& inv) {
  const double s = (double)n_in / (double)n_out;
  lo = (double)j * s; hi = (double)(j + 1) * s;
  i0 = (int)floor(lo);
  i1 = (int)ceil(hi);
  if (i1 > n_in) i1 = n_in;
  double tot = 0.0;
  for (int i = i0; i < i1; ++i) tot += fmin(hi, (double)(i + 1)) - fmax(lo, (double)i);
  inv = 1.0 / tot;
}
// src: (H, W, 3) uint8 HWC -> dst: (3, Ho, Wo) planar f32 holding the uint8-rounded averages
__global__ void __launch_bounds__(256) aug_area_u8_k(const uint8_t* __restrict__ src, float* __restrict__ dst, int H, int W, int Ho, int Wo) {
  const long n = (long)Ho * Wo;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int y = (int)(i / Wo), x = (int)(i - (long)y * Wo);
    int y0, y1, x0, x1;
    double ylo, yhi, yinv, xlo, xhi, xinv;
    area_span(y, H, Ho, y0, y1, ylo, yhi, yinv);
    area_span(x, W, Wo, x0, x1, xlo, xhi, xinv);
    // host order: out[oh, w] = sum_h Wy[oh, h] src[h, w] (float64), then out[oh, ow] = sum_w Wx[ow, w] out[oh, w]
    double acc[3] = {0.0, 0.0, 0.0};
    for (int xx = x0; xx < x1; ++xx) {
      const double wx = (fmin(xhi, (double)(xx + 1)) - fmax(xlo, (double)xx)) * xinv;
      double col[3] = {0.0, 0.0, 0.0};
      for (int yy = y0; yy < y1; ++yy) {
        const double wy = (fmin(yhi, (double)(yy + 1)) - fmax(ylo, (double)yy)) * yinv;
        const uint8_t* p = src + ((long)yy * W + xx) * 3;
        col[0] += wy * (double)p[0]; col[1] += wy * (double)p[1]; col[2] += wy * (double)p[2];
      }
      acc[0] += wx * col[0]; acc[1] += wx * col[1]; acc[2] += wx * col[2];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[c * n + i] = (float)fmin(fmax(rint(acc[c]), 0.0), 255.0);
  }
}
extern "C" int ge_aug_area_u8(const uint8_t* src_hwc, float* dst, int H, int W, int Ho, int Wo, void* stream) {
  if (!src_hwc || !dst || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return GE_ERR_BAD_ARG;
  if (Ho > H || Wo > W) return GE_ERR_UNSUPPORTED;               // the area filter is the shrinking branch (DDAD: 1216 x 1936 -> 384 x 640)
  aug_area_u8_k<<<ge_blocks((long)Ho * Wo, 256, 65536), 256, 0, ge_stream(stream)>>>(src_hwc, dst, H, W, Ho, Wo);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
// Sparse re-projection as a deterministic gather: dst[Y, X] = the LAST source pixel in row-major order with src > 0,
// (int)(y * (Ho / H)) == Y and (int)(x * (Wo / W)) == X (float64 product, truncation: numpy's int64 * float -> astype(int32)), else 0.
__global__ void __launch_bounds__(256) aug_splat_k(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int Ho, int Wo) {
  const double sy = (double)Ho / (double)H, sx = (double)Wo / (double)W;
  const long n = (long)Ho * Wo;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int Y = (int)(i / Wo), X = (int)(i - (long)Y * Wo);
    int ylo = (int)floor((double)Y / sy) - 1, yhi = (int)ceil((double)(Y + 1) / sy) + 1;
    int xlo = (int)floor((double)X / sx) - 1, xhi = (int)ceil((double)(X + 1) / sx) + 1;
    ylo = max(ylo, 0); xlo = max(xlo, 0); yhi = min(yhi, H - 1); xhi = min(xhi, W - 1);
    float v = 0.f;
    bool found = false;
    for (int y = yhi; y >= ylo && !found; --y) {
      if ((int)((double)y * sy) != Y) continue;
      for (int x = xhi; x >= xlo; --x) {
        if ((int)((double)x * sx) != X) continue;
        const float s = src[(long)y * W + x];
        if (s > 0.f) { v = s; found = true; break; }
      }
    }
    dst[i] = v;
  }
}
extern "C" int ge_aug_splat(const float* src, float* dst, int H, int W, int Ho, int Wo, void* stream) {
  if (!src || !dst || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return GE_ERR_BAD_ARG;
  aug_splat_k<<<ge_blocks((long)Ho * Wo, 256, 65536), 256, 0, ge_stream(stream)>>>(src, dst, H, W, Ho, Wo);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
