// Channels-last (NHWC) variants of the map kernels of the HAHI neck / PE necks / DenseDepth head.
//
// MIOpen's bf16 implicit-GEMM convolutions are NHWC inside; on NCHW tensors it brackets every call with batched_transpose
// kernels (4.9 ms of the 92 ms KITTI step, profiles/r1_bench_kernel_stats.csv) and the HAHI neck pays two more transposing
// passes to get tokens (B, H*W, C) out of / into maps.  A channels-last map (B, C, H, W) with strides (HWC, 1, WC, C) IS the
// token matrix (B*H*W, C): 1x1 convs, the deformable attention and the 3x3 convs then share one layout and nothing is
// transposed.  These kernels are what the path needs on that layout:
//   ge_bn_act_nhwc_*     training-mode BatchNorm2d + (Leaky)ReLU (mmcv ConvModule, necks/hahi.py:150-166): per-channel
//                        statistics are COLUMN sums of the (rows, C) matrix
//   ge_bias_act_nhwc_*   conv bias + LeakyReLU (densedepth_head.py:14-27, pemask_neck.py:36-42), bias gradient = column sums
//   ge_bilinear_nhwc_*   F.interpolate(bilinear), both align_corners conventions, 16 bytes of channels per lane
//   ge_concat_rows_*     torch.cat([a, b], 1) of two channels-last maps / token matrices with the HAHI glue folded in:
//                        dropout(tokens) + identity written into its slot (hahi.py:326-346)
//   ge_add_rows          tokens + positional embedding (fp32 (N, C), broadcast over the batch; hahi.py:303-306)
// All HBM-bound streaming work: 16-byte accesses along the channel dimension, fp32 arithmetic, no MFMA.
#include "common.h"

#define NH_MAXLANES 256

// ---------------------------------------------------------------------------------------------- column sums
// K column sums of f_k(row element) over a (R, C) matrix: thread = (row slot, channel vector); fp32 per thread over its rows, LDS
// tree across the row slots of a workgroup, one fp32 partial per (workgroup, k, channel) into `partial`, then
// nh_partials_reduce_k sums the partials of all workgroups in fp64 into ws[k * C + c].  No atomics: same-address fp64 atomics
// serialise in L2 at ~0.1 us each (512 workgroups x C channels cost 50-100 us whatever the matrix size; the first version).
#define NH_MAXBLOCKS 1024
template <int K, int VN>
__device__ __forceinline__ void nh_col_commit(float (&acc)[K][VN], float* __restrict__ partial, int C, int lpr, int rpi, int c0) {
  __shared__ float sm[K][NH_MAXLANES][VN];
  const int t = threadIdx.x, rs = t / lpr;
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int v = 0; v < VN; ++v) sm[k][t][v] = acc[k][v];
  __syncthreads();
  int span = 1;
  while (span < rpi) span <<= 1;
  for (int stride = span >> 1; stride > 0; stride >>= 1) {          // tree over the row slots: all of them work
    if (rs < stride && rs + stride < rpi) {
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int v = 0; v < VN; ++v) sm[k][t][v] += sm[k][t + stride * lpr][v];
    }
    __syncthreads();
  }
  if (t < lpr) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float* o = partial + ((long)blockIdx.x * K + k) * C + c0;
#pragma unroll
      for (int v = 0; v < VN; v += 4) *(float4*)(o + v) = make_float4(sm[k][t][v], sm[k][t][v + 1], sm[k][t][v + 2], sm[k][t][v + 3]);
    }
  }
}
// ws[kc] = sum over nb workgroups of partial[b * KC + kc], fp64; 32 columns x 8 workgroup slices per 256 threads
template <typename O>
__global__ void __launch_bounds__(256) nh_partials_reduce_k(const float* __restrict__ partial, O* __restrict__ ws, int KC, int nb, int accumulate) {
  __shared__ double sm[8][32];
  const int j = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int kc = blockIdx.x * 32 + j;
  double a = 0.0;
  if (kc < KC) {
#pragma unroll 8
    for (int b = sl; b < nb; b += 8) a += (double)partial[(long)b * KC + kc];
  }
  sm[sl][j] = a;
  __syncthreads();
  if (sl == 0 && kc < KC) {
#pragma unroll
    for (int q = 1; q < 8; ++q) a += sm[q][j];
    ws[kc] = accumulate ? (O)((double)ws[kc] + a) : (O)a;
  }
}
static inline unsigned nh_grid_rows(long R, int rpi) {
  long b = (R + (long)rpi * 8 - 1) / ((long)rpi * 8);
  if (b < 1) b = 1;
  if (b > NH_MAXBLOCKS) b = NH_MAXBLOCKS;
  return (unsigned)b;
}
// workspace of the column-sum users: K*C doubles (sums) | 3*C floats (BatchNorm coefficients) | NH_MAXBLOCKS*K*C floats (partials)
extern "C" size_t ge_nhwc_workspace(int C, int K) {
  return (size_t)C * ((size_t)K * sizeof(double) + 3 * sizeof(float) + (size_t)NH_MAXBLOCKS * K * sizeof(float));
}
static inline float* nh_partials(void* workspace, int C, int K) { return (float*)((char*)workspace + (size_t)C * (K * sizeof(double) + 3 * sizeof(float))); }
// K = 1 users that want the sums as fp32 directly (bias gradients)
static inline void nh_reduce_launch_f32(void* workspace, int C, unsigned nb, float* out, int accumulate, hipStream_t s) {
  nh_partials_reduce_k<float><<<(C + 31) / 32, 256, 0, s>>>(nh_partials(workspace, C, 1), out, C, (int)nb, accumulate);
}
static inline void nh_reduce_launch(void* workspace, int C, int K, unsigned nb, hipStream_t s) {
  nh_partials_reduce_k<double><<<(K * C + 31) / 32, 256, 0, s>>>(nh_partials(workspace, C, K), (double*)workspace, K * C, (int)nb, 0);
}

// ============================================================================ BatchNorm2d (training) + (Leaky)ReLU
template <typename T>
__global__ void __launch_bounds__(256) bn_stats_nhwc_k(const T* __restrict__ x, float* __restrict__ ws, int C, long R, int lpr, int rpi) {
  constexpr int VN = V8<T>::N;
  const int t = threadIdx.x, lane = t % lpr, rs = t / lpr;
  const int c0 = lane * VN;
  float acc[2][VN];
#pragma unroll
  for (int v = 0; v < VN; ++v) { acc[0][v] = 0.f; acc[1][v] = 0.f; }
  if (rs < rpi)
    for (long r = (long)blockIdx.x * rpi + rs; r < R; r += (long)gridDim.x * rpi) {
      float v[VN];
      V8<T>::ld(x + r * C + c0, v);
#pragma unroll
      for (int k = 0; k < VN; ++k) { acc[0][k] += v[k]; acc[1][k] += v[k] * v[k]; }
    }
  nh_col_commit<2, VN>(acc, ws, C, lpr, rpi, c0);
}
template <typename T>
__global__ void __launch_bounds__(256) bn_apply_nhwc_k(const T* __restrict__ x, const float* __restrict__ coef, T* __restrict__ y, int C,
                                                       long R, int lpr, int rpi, float slope) {
  constexpr int VN = V8<T>::N;
  const int t = threadIdx.x, lane = t % lpr, rs = t / lpr;
  if (rs >= rpi) return;
  const int c0 = lane * VN;
  float a[VN], b[VN];
#pragma unroll
  for (int k = 0; k < VN; ++k) { a[k] = coef[c0 + k]; b[k] = coef[C + c0 + k]; }
  for (long r = (long)blockIdx.x * rpi + rs; r < R; r += (long)gridDim.x * rpi) {
    float v[VN];
    V8<T>::ld(x + r * C + c0, v);
#pragma unroll
    for (int k = 0; k < VN; ++k) { const float tv = __fmaf_rn(v[k], a[k], b[k]); v[k] = tv > 0.f ? tv : tv * slope; }
    V8<T>::st(y + r * C + c0, v);
  }
}
// RECOMP: the activation decision y > 0 is recomputed from x with the forward's coefficients (same fma, same operands) instead of
// reading y: one tensor less per pass (2 of the 7 passes of the backward)
template <typename T, bool RECOMP>
__global__ void __launch_bounds__(256) bn_bwd_stats_nhwc_k(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ x,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
                                                           float* __restrict__ ws, int C, long R, int lpr, int rpi, float slope) {
  constexpr int VN = V8<T>::N;
  const int t = threadIdx.x, lane = t % lpr, rs = t / lpr;
  const int c0 = lane * VN;
  float mu[VN], rsd[VN], acc[2][VN], fa[VN], fb[VN];
#pragma unroll
  for (int v = 0; v < VN; ++v) {
    mu[v] = save_mean[c0 + v]; rsd[v] = save_rstd[c0 + v]; acc[0][v] = 0.f; acc[1][v] = 0.f;
    if (RECOMP) { fa[v] = gamma[c0 + v] * rsd[v]; fb[v] = __fmaf_rn(-mu[v], fa[v], beta[c0 + v]); }
  }
  if (rs < rpi)
    for (long r = (long)blockIdx.x * rpi + rs; r < R; r += (long)gridDim.x * rpi) {
      float g[VN], yv[VN], xv[VN];
      V8<T>::ld(dy + r * C + c0, g);
      V8<T>::ld(x + r * C + c0, xv);
      if (RECOMP) {
#pragma unroll
        for (int k = 0; k < VN; ++k) yv[k] = __fmaf_rn(xv[k], fa[k], fb[k]);
      } else {
        V8<T>::ld(y + r * C + c0, yv);
      }
#pragma unroll
      for (int k = 0; k < VN; ++k) {
        const float gg = yv[k] > 0.f ? g[k] : g[k] * slope;
        acc[0][k] += gg;
        acc[1][k] += gg * ((xv[k] - mu[k]) * rsd[k]);
      }
    }
  nh_col_commit<2, VN>(acc, ws, C, lpr, rpi, c0);
}
template <typename T, bool RECOMP>
__global__ void __launch_bounds__(256) bn_bwd_apply_nhwc_k(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ x,
                                                           const float* __restrict__ beta, const float* __restrict__ save_mean,
                                                           const float* __restrict__ save_rstd, const float* __restrict__ coef,
                                                           T* __restrict__ dx, int C, long R, int lpr, int rpi, float slope) {
  constexpr int VN = V8<T>::N;
  const int t = threadIdx.x, lane = t % lpr, rs = t / lpr;
  if (rs >= rpi) return;
  const int c0 = lane * VN;
  float a[VN], m1[VN], m2[VN], mu[VN], rsd[VN], fb[VN];
#pragma unroll
  for (int k = 0; k < VN; ++k) {
    a[k] = coef[c0 + k]; m1[k] = coef[C + c0 + k]; m2[k] = coef[2 * C + c0 + k]; mu[k] = save_mean[c0 + k]; rsd[k] = save_rstd[c0 + k];
    if (RECOMP) fb[k] = __fmaf_rn(-mu[k], a[k], beta[c0 + k]);          // coef[c] is gamma * rstd, the forward's scale
  }
  for (long r = (long)blockIdx.x * rpi + rs; r < R; r += (long)gridDim.x * rpi) {
    float g[VN], yv[VN], xv[VN];
    V8<T>::ld(dy + r * C + c0, g);
    V8<T>::ld(x + r * C + c0, xv);
    if (RECOMP) {
#pragma unroll
      for (int k = 0; k < VN; ++k) yv[k] = __fmaf_rn(xv[k], a[k], fb[k]);
    } else {
      V8<T>::ld(y + r * C + c0, yv);
    }
#pragma unroll
    for (int k = 0; k < VN; ++k) {
      const float gg = yv[k] > 0.f ? g[k] : g[k] * slope;
      g[k] = a[k] * (gg - m1[k] - ((xv[k] - mu[k]) * rsd[k]) * m2[k]);
    }
    V8<T>::st(dx + r * C + c0, g);
  }
}

// finalize kernels: same arithmetic as the NCHW path (norm.hip), restated here because the two files are separate TUs
__global__ void __launch_bounds__(256) bn_finalize_nhwc_k(const double* __restrict__ ws, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ save_mean,
                                                          float* __restrict__ save_rstd, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, float* __restrict__ coef, int C, double n,
                                                          float eps, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mean = ws[c] / n;
  double var = ws[C + c] / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  save_mean[c] = (float)mean;
  save_rstd[c] = rstd;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(n > 1.0 ? var * n / (n - 1.0) : var);
  const float a = gamma[c] * rstd;
  coef[c] = a;
  coef[C + c] = __fmaf_rn(-(float)mean, a, beta[c]);       // explicit fma: bn_bwd_*_nhwc_k<.., true> recompute exactly this
}
__global__ void __launch_bounds__(256) bn_bwd_finalize_nhwc_k(const double* __restrict__ ws, const float* __restrict__ gamma,
                                                              const float* __restrict__ save_rstd, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ coef, int C, double n) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  dbeta[c] = (float)ws[c];
  dgamma[c] = (float)ws[C + c];
  coef[c] = gamma[c] * save_rstd[c];
  coef[C + c] = (float)(ws[c] / n);
  coef[2 * C + c] = (float)(ws[C + c] / n);
}

template <typename T> static bool nh_geom(int C, int& lpr, int& rpi) {
  constexpr int VN = V8<T>::N;
  if (C <= 0 || C % VN) return false;
  lpr = C / VN;
  if (lpr > NH_MAXLANES) return false;
  rpi = NH_MAXLANES / lpr;
  return true;
}
static inline bool nh_aligned(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
  return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d)) & 15) == 0;
}
static inline unsigned nh_grid_vec(long nvec) { return ge_blocks(nvec, 256 * 4, 65536); }
static inline unsigned nh_grid_apply(long R, int rpi) {           // ~4 rows per thread, enough workgroups to fill 256 CUs several times
  long b = (R + (long)rpi * 4 - 1) / ((long)rpi * 4);
  if (b < 1) b = 1;
  if (b > 16384) b = 16384;
  return (unsigned)b;
}

// workspace: ge_nhwc_workspace(C, 2) bytes: double[2C] sums | float[3C] coefficients | partials
template <typename T>
static int bn_nhwc_fwd_launch(const void* x, const float* gamma, const float* beta, void* y, float* save_mean, float* save_rstd,
                              float* running_mean, float* running_var, void* workspace, long R, int C, float eps, float momentum,
                              float slope, hipStream_t s) {
  int lpr, rpi;
  if (!nh_geom<T>(C, lpr, rpi) || !nh_aligned(x, y)) return GE_ERR_UNSUPPORTED;
  double* ws = (double*)workspace;
  float* coef = (float*)(ws + 2 * C);
  const unsigned nb = nh_grid_rows(R, rpi);
  bn_stats_nhwc_k<T><<<nb, 256, 0, s>>>((const T*)x, nh_partials(workspace, C, 2), C, R, lpr, rpi);
  GE_LAUNCH_CHECK();
  nh_reduce_launch(workspace, C, 2, nb, s);
  GE_LAUNCH_CHECK();
  bn_finalize_nhwc_k<<<(C + 255) / 256, 256, 0, s>>>(ws, gamma, beta, save_mean, save_rstd, running_mean, running_var, coef, C, (double)R, eps, momentum);
  GE_LAUNCH_CHECK();
  bn_apply_nhwc_k<T><<<nh_grid_apply(R, rpi), 256, 0, s>>>((const T*)x, coef, (T*)y, C, R, lpr, rpi, slope);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
template <typename T>
static int bn_nhwc_bwd_launch(const void* dy, const void* y, const void* x, const float* gamma, const float* beta, const float* save_mean,
                              const float* save_rstd, void* dx, float* dgamma, float* dbeta, void* workspace, long R, int C, float slope,
                              hipStream_t s) {
  int lpr, rpi;
  if (!nh_geom<T>(C, lpr, rpi) || !nh_aligned(dy, y, x, dx)) return GE_ERR_UNSUPPORTED;
  double* ws = (double*)workspace;
  float* coef = (float*)(ws + 2 * C);
  const unsigned nb = nh_grid_rows(R, rpi);
  float* part = nh_partials(workspace, C, 2);
  if (beta) bn_bwd_stats_nhwc_k<T, true><<<nb, 256, 0, s>>>((const T*)dy, nullptr, (const T*)x, gamma, beta, save_mean, save_rstd, part, C, R, lpr, rpi, slope);
  else bn_bwd_stats_nhwc_k<T, false><<<nb, 256, 0, s>>>((const T*)dy, (const T*)y, (const T*)x, gamma, beta, save_mean, save_rstd, part, C, R, lpr, rpi, slope);
  GE_LAUNCH_CHECK();
  nh_reduce_launch(workspace, C, 2, nb, s);
  GE_LAUNCH_CHECK();
  bn_bwd_finalize_nhwc_k<<<(C + 255) / 256, 256, 0, s>>>(ws, gamma, save_rstd, dgamma, dbeta, coef, C, (double)R);
  GE_LAUNCH_CHECK();
  if (beta) bn_bwd_apply_nhwc_k<T, true><<<nh_grid_apply(R, rpi), 256, 0, s>>>((const T*)dy, nullptr, (const T*)x, beta, save_mean, save_rstd, coef, (T*)dx, C, R, lpr, rpi, slope);
  else bn_bwd_apply_nhwc_k<T, false><<<nh_grid_apply(R, rpi), 256, 0, s>>>((const T*)dy, (const T*)y, (const T*)x, beta, save_mean, save_rstd, coef, (T*)dx, C, R, lpr, rpi, slope);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
extern "C" int ge_bn_act_nhwc_fwd(const void* x, const float* gamma, const float* beta, void* y, float* save_mean, float* save_rstd,
                                  float* running_mean, float* running_var, void* workspace, long rows, int C, float eps,
                                  float momentum, float slope, int dtype, void* stream) {
  if (!x || !gamma || !beta || !y || !save_mean || !save_rstd || !workspace || rows <= 0 || C <= 0) return GE_ERR_BAD_ARG;
  if (dtype == GE_F32) return bn_nhwc_fwd_launch<float>(x, gamma, beta, y, save_mean, save_rstd, running_mean, running_var, workspace, rows, C, eps, momentum, slope, ge_stream(stream));
  if (dtype == GE_BF16) return bn_nhwc_fwd_launch<bf16_t>(x, gamma, beta, y, save_mean, save_rstd, running_mean, running_var, workspace, rows, C, eps, momentum, slope, ge_stream(stream));
  return GE_ERR_UNSUPPORTED;
}
// exactly one of `y` (the forward output) and `beta` (the forward's shift: the activation decision is recomputed from x) is needed
extern "C" int ge_bn_act_nhwc_bwd(const void* dy, const void* y, const void* x, const float* gamma, const float* beta,
                                  const float* save_mean, const float* save_rstd, void* dx, float* dgamma, float* dbeta, void* workspace,
                                  long rows, int C, float slope, int dtype, void* stream) {
  if (!dy || (!y && !beta) || !x || !gamma || !save_mean || !save_rstd || !dx || !dgamma || !dbeta || !workspace || rows <= 0 || C <= 0) return GE_ERR_BAD_ARG;
  if (dtype == GE_F32) return bn_nhwc_bwd_launch<float>(dy, y, x, gamma, beta, save_mean, save_rstd, dx, dgamma, dbeta, workspace, rows, C, slope, ge_stream(stream));
  if (dtype == GE_BF16) return bn_nhwc_bwd_launch<bf16_t>(dy, y, x, gamma, beta, save_mean, save_rstd, dx, dgamma, dbeta, workspace, rows, C, slope, ge_stream(stream));
  return GE_ERR_UNSUPPORTED;
}

// ============================================================================================ bias + LeakyReLU
template <typename T>
__global__ void __launch_bounds__(256) bias_act_nhwc_fwd_k(T* __restrict__ x, const float* __restrict__ bias, int C, long R, int lpr, int rpi,
                                                           float slope) {
  constexpr int VN = V8<T>::N;
  const int t = threadIdx.x, lane = t % lpr, rs = t / lpr;
  if (rs >= rpi) return;
  const int c0 = lane * VN;
  float b[VN];
#pragma unroll
  for (int k = 0; k < VN; ++k) b[k] = bias[c0 + k];
  for (long r = (long)blockIdx.x * rpi + rs; r < R; r += (long)gridDim.x * rpi) {
    float v[VN];
    V8<T>::ld(x + r * C + c0, v);
#pragma unroll
    for (int k = 0; k < VN; ++k) { const float tv = v[k] + b[k]; v[k] = tv > 0.f ? tv : tv * slope; }
    V8<T>::st(x + r * C + c0, v);
  }
}
// dx = dy * act'(y); d_bias (fp64 workspace, C doubles, zeroed by the launcher) = column sums of dx
template <typename T>
__global__ void __launch_bounds__(256) bias_act_nhwc_bwd_k(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx,
                                                           float* __restrict__ ws, int C, long R, int lpr, int rpi, float slope) {
  constexpr int VN = V8<T>::N;
  const int t = threadIdx.x, lane = t % lpr, rs = t / lpr;
  const int c0 = lane * VN;
  float acc[1][VN];
#pragma unroll
  for (int v = 0; v < VN; ++v) acc[0][v] = 0.f;
  if (rs < rpi)
    for (long r = (long)blockIdx.x * rpi + rs; r < R; r += (long)gridDim.x * rpi) {
      float g[VN], yv[VN];
      V8<T>::ld(dy + r * C + c0, g);
      V8<T>::ld(y + r * C + c0, yv);
#pragma unroll
      for (int k = 0; k < VN; ++k) { g[k] = yv[k] > 0.f ? g[k] : g[k] * slope; acc[0][k] += g[k]; }
      V8<T>::st(dx + r * C + c0, g);
    }
  nh_col_commit<1, VN>(acc, ws, C, lpr, rpi, c0);
}
extern "C" int ge_bias_act_nhwc_fwd(void* x, const float* bias, long rows, int C, float slope, int dtype, void* stream) {
  if (!x || !bias || rows <= 0 || C <= 0) return GE_ERR_BAD_ARG;
  int lpr, rpi;
  hipStream_t s = ge_stream(stream);
  if (dtype == GE_F32) {
    if (!nh_geom<float>(C, lpr, rpi) || !nh_aligned(x)) return GE_ERR_UNSUPPORTED;
    bias_act_nhwc_fwd_k<float><<<nh_grid_apply(rows, rpi), 256, 0, s>>>((float*)x, bias, C, rows, lpr, rpi, slope);
  } else if (dtype == GE_BF16) {
    if (!nh_geom<bf16_t>(C, lpr, rpi) || !nh_aligned(x)) return GE_ERR_UNSUPPORTED;
    bias_act_nhwc_fwd_k<bf16_t><<<nh_grid_apply(rows, rpi), 256, 0, s>>>((bf16_t*)x, bias, C, rows, lpr, rpi, slope);
  } else {
    return GE_ERR_UNSUPPORTED;
  }
  GE_LAUNCH_CHECK();
  return GE_OK;
}
// workspace: ge_nhwc_workspace(C, 1) bytes
extern "C" int ge_bias_act_nhwc_bwd(const void* dy, const void* y, void* dx, float* dbias, void* workspace, long rows, int C, float slope,
                                    int dtype, void* stream) {
  if (!dy || !y || !dx || !dbias || !workspace || rows <= 0 || C <= 0) return GE_ERR_BAD_ARG;
  int lpr, rpi;
  hipStream_t s = ge_stream(stream);
  double* ws = (double*)workspace;
  float* part = nh_partials(workspace, C, 1);
  unsigned nb = 1;
  if (dtype == GE_F32) {
    if (!nh_geom<float>(C, lpr, rpi) || !nh_aligned(dy, y, dx)) return GE_ERR_UNSUPPORTED;
    nb = nh_grid_rows(rows, rpi);
    bias_act_nhwc_bwd_k<float><<<nb, 256, 0, s>>>((const float*)dy, (const float*)y, (float*)dx, part, C, rows, lpr, rpi, slope);
  } else if (dtype == GE_BF16) {
    if (!nh_geom<bf16_t>(C, lpr, rpi) || !nh_aligned(dy, y, dx)) return GE_ERR_UNSUPPORTED;
    nb = nh_grid_rows(rows, rpi);
    bias_act_nhwc_bwd_k<bf16_t><<<nb, 256, 0, s>>>((const bf16_t*)dy, (const bf16_t*)y, (bf16_t*)dx, part, C, rows, lpr, rpi, slope);
  } else {
    return GE_ERR_UNSUPPORTED;
  }
  GE_LAUNCH_CHECK();
  nh_reduce_launch_f32(workspace, C, nb, dbias, 0, s);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// ======================================================================================================= bilinear
template <typename T>
__global__ void __launch_bounds__(256) bilinear_nhwc_fwd_k(const T* __restrict__ in, T* __restrict__ out, int N, int Hi, int Wi, int Ho, int Wo,
                                                           int C, int lpr, int align) {
  constexpr int VN = V8<T>::N;
  const float sy = ge_scale(Hi, Ho, align), sx = ge_scale(Wi, Wo, align);
  const long total = (long)N * Ho * Wo * lpr;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c0 = (int)(idx % lpr) * VN;
    long t = idx / lpr;
    const int x = (int)(t % Wo); t /= Wo;
    const int y = (int)(t % Ho);
    const long n = t / Ho;
    const Lerp ly = ge_lerp(y, Hi, sy, align), lx = ge_lerp(x, Wi, sx, align);
    const T* base = in + n * (long)Hi * Wi * C + c0;
    float v00[VN], v01[VN], v10[VN], v11[VN], o[VN];
    V8<T>::ld(base + ((long)ly.i0 * Wi + lx.i0) * C, v00);
    V8<T>::ld(base + ((long)ly.i0 * Wi + lx.i1) * C, v01);
    V8<T>::ld(base + ((long)ly.i1 * Wi + lx.i0) * C, v10);
    V8<T>::ld(base + ((long)ly.i1 * Wi + lx.i1) * C, v11);
#pragma unroll
    for (int k = 0; k < VN; ++k) o[k] = ly.w0 * (lx.w0 * v00[k] + lx.w1 * v01[k]) + ly.w1 * (lx.w0 * v10[k] + lx.w1 * v11[k]);
    V8<T>::st(out + ((n * Ho + y) * (long)Wo + x) * C + c0, o);
  }
}
// candidate output indices whose taps can touch input index X (same rule as the NCHW kernel, ground.hip)
__device__ __forceinline__ void nh_cand_range(int X, int in, int out, float scale, bool align, int& lo, int& hi) {
  if (scale <= 0.f) { lo = 0; hi = out - 1; return; }
  float a, b;
  if (align) { a = ((float)X - 1.f) / scale; b = ((float)X + 1.f) / scale; }
  else { a = ((float)X - 0.5f) / scale - 0.5f; b = ((float)X + 1.5f) / scale - 0.5f; }
  lo = (int)floorf(a) - 1;
  hi = (int)ceilf(b) + 1;
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}
// deterministic gather: each input element sums the contributions of the outputs whose taps touch it (no atomics)
template <typename T>
__global__ void __launch_bounds__(256) bilinear_nhwc_bwd_k(const T* __restrict__ gout, T* __restrict__ gin, int N, int Hi, int Wi, int Ho, int Wo,
                                                           int C, int lpr, int align) {
  constexpr int VN = V8<T>::N;
  const float sy = ge_scale(Hi, Ho, align), sx = ge_scale(Wi, Wo, align);
  const long total = (long)N * Hi * Wi * lpr;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c0 = (int)(idx % lpr) * VN;
    long t = idx / lpr;
    const int X = (int)(t % Wi); t /= Wi;
    const int Y = (int)(t % Hi);
    const long n = t / Hi;
    int ylo, yhi, xlo, xhi;
    nh_cand_range(Y, Hi, Ho, sy, align, ylo, yhi);
    nh_cand_range(X, Wi, Wo, sx, align, xlo, xhi);
    const T* g = gout + n * (long)Ho * Wo * C + c0;
    float acc[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) acc[k] = 0.f;
    for (int oy = ylo; oy <= yhi; ++oy) {
      const Lerp ly = ge_lerp(oy, Hi, sy, align);
      const float wy = (ly.i0 == Y ? ly.w0 : 0.f) + (ly.i1 == Y ? ly.w1 : 0.f);
      if (wy == 0.f) continue;
      float row[VN];
#pragma unroll
      for (int k = 0; k < VN; ++k) row[k] = 0.f;
      for (int ox = xlo; ox <= xhi; ++ox) {
        const Lerp lx = ge_lerp(ox, Wi, sx, align);
        const float wx = (lx.i0 == X ? lx.w0 : 0.f) + (lx.i1 == X ? lx.w1 : 0.f);
        if (wx == 0.f) continue;
        float v[VN];
        V8<T>::ld(g + ((long)oy * Wo + ox) * C, v);
#pragma unroll
        for (int k = 0; k < VN; ++k) row[k] += wx * v[k];
      }
#pragma unroll
      for (int k = 0; k < VN; ++k) acc[k] += wy * row[k];
    }
    V8<T>::st(gin + ((n * Hi + Y) * (long)Wi + X) * C + c0, acc);
  }
}
// Large up-sampling factors (the PE necks bring 11x35 maps to 176x560): a gather over (2f)^2 candidate outputs per input
// element leaves too few, too long threads.  Bilinear interpolation is separable, so is its transpose:
//   horizontal: tmp[n, oy, X, c] = sum_ox wx(ox -> X) g[n, oy, ox, c]     (N * Ho * Wi * C/VN threads, ~2f taps each, fp32 tmp)
//   vertical  : gin[n, Y, X, c]  = sum_oy wy(oy -> Y) tmp[n, oy, X, c]
template <typename T>
__global__ void __launch_bounds__(256) bilinear_nhwc_bwd_h_k(const T* __restrict__ gout, float* __restrict__ tmp, int N, int Wi, int Ho, int Wo,
                                                             int C, int lpr, int align) {
  constexpr int VN = V8<T>::N;
  const float sx = ge_scale(Wi, Wo, align);
  const long total = (long)N * Ho * Wi * lpr;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c0 = (int)(idx % lpr) * VN;
    long t = idx / lpr;
    const int X = (int)(t % Wi);
    const long row = t / Wi;                                  // n * Ho + oy
    int xlo, xhi;
    nh_cand_range(X, Wi, Wo, sx, align, xlo, xhi);
    const T* g = gout + row * (long)Wo * C + c0;
    float acc[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) acc[k] = 0.f;
    for (int ox = xlo; ox <= xhi; ++ox) {
      const Lerp lx = ge_lerp(ox, Wi, sx, align);
      const float wx = (lx.i0 == X ? lx.w0 : 0.f) + (lx.i1 == X ? lx.w1 : 0.f);
      if (wx == 0.f) continue;
      float v[VN];
      V8<T>::ld(g + (long)ox * C, v);
#pragma unroll
      for (int k = 0; k < VN; ++k) acc[k] += wx * v[k];
    }
    float* o = tmp + (row * Wi + X) * (long)C + c0;
#pragma unroll
    for (int k = 0; k < VN; k += 4) *(float4*)(o + k) = make_float4(acc[k], acc[k + 1], acc[k + 2], acc[k + 3]);
  }
}
template <typename T>
__global__ void __launch_bounds__(256) bilinear_nhwc_bwd_v_k(const float* __restrict__ tmp, T* __restrict__ gin, int N, int Hi, int Wi, int Ho,
                                                             int C, int lpr, int align) {
  constexpr int VN = V8<T>::N;
  const float sy = ge_scale(Hi, Ho, align);
  const long total = (long)N * Hi * Wi * lpr;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c0 = (int)(idx % lpr) * VN;
    long t = idx / lpr;
    const int X = (int)(t % Wi); t /= Wi;
    const int Y = (int)(t % Hi);
    const long n = t / Hi;
    int ylo, yhi;
    nh_cand_range(Y, Hi, Ho, sy, align, ylo, yhi);
    float acc[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) acc[k] = 0.f;
    for (int oy = ylo; oy <= yhi; ++oy) {
      const Lerp ly = ge_lerp(oy, Hi, sy, align);
      const float wy = (ly.i0 == Y ? ly.w0 : 0.f) + (ly.i1 == Y ? ly.w1 : 0.f);
      if (wy == 0.f) continue;
      const float* p = tmp + ((n * Ho + oy) * (long)Wi + X) * C + c0;
#pragma unroll
      for (int k = 0; k < VN; k += 4) {
        const float4 q = *(const float4*)(p + k);
        acc[k] += wy * q.x; acc[k + 1] += wy * q.y; acc[k + 2] += wy * q.z; acc[k + 3] += wy * q.w;
      }
    }
    V8<T>::st(gin + ((n * Hi + Y) * (long)Wi + X) * C + c0, acc);
  }
}

template <typename T>
static int bilinear_nhwc_launch(bool fwd, const void* src, void* dst, int N, int C, int Hi, int Wi, int Ho, int Wo, int align, hipStream_t s,
                                void* workspace = nullptr, size_t workspace_bytes = 0) {
  int lpr, rpi;
  if (!nh_geom<T>(C, lpr, rpi) || !nh_aligned(src, dst)) return GE_ERR_UNSUPPORTED;
  if (!fwd && workspace && (Ho > 3 * Hi || Wo > 3 * Wi) && workspace_bytes >= (size_t)N * Ho * Wi * C * sizeof(float) && nh_aligned(workspace)) {
    const long th = (long)N * Ho * Wi * lpr, tv = (long)N * Hi * Wi * lpr;
    if (th == 0 || tv == 0) return GE_OK;
    bilinear_nhwc_bwd_h_k<T><<<ge_blocks(th, 256, 1 << 18), 256, 0, s>>>((const T*)src, (float*)workspace, N, Wi, Ho, Wo, C, lpr, align);
    GE_LAUNCH_CHECK();
    bilinear_nhwc_bwd_v_k<T><<<ge_blocks(tv, 256, 1 << 18), 256, 0, s>>>((const float*)workspace, (T*)dst, N, Hi, Wi, Ho, C, lpr, align);
    GE_LAUNCH_CHECK();
    return GE_OK;
  }
  if (fwd) {
    const long total = (long)N * Ho * Wo * lpr;
    if (total == 0) return GE_OK;
    bilinear_nhwc_fwd_k<T><<<ge_blocks(total, 256, 1 << 18), 256, 0, s>>>((const T*)src, (T*)dst, N, Hi, Wi, Ho, Wo, C, lpr, align);
  } else {
    const long total = (long)N * Hi * Wi * lpr;
    if (total == 0) return GE_OK;
    bilinear_nhwc_bwd_k<T><<<ge_blocks(total, 256, 1 << 18), 256, 0, s>>>((const T*)src, (T*)dst, N, Hi, Wi, Ho, Wo, C, lpr, align);
  }
  GE_LAUNCH_CHECK();
  return GE_OK;
}
// in (N, Hi, Wi, C) -> out (N, Ho, Wo, C)
extern "C" int ge_bilinear_nhwc_fwd(const void* in, void* out, int N, int C, int Hi, int Wi, int Ho, int Wo, int align_corners, int dtype,
                                    void* stream) {
  if (!in || !out || N < 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return GE_ERR_BAD_ARG;
  if (dtype == GE_F32) return bilinear_nhwc_launch<float>(true, in, out, N, C, Hi, Wi, Ho, Wo, align_corners, ge_stream(stream));
  if (dtype == GE_BF16) return bilinear_nhwc_launch<bf16_t>(true, in, out, N, C, Hi, Wi, Ho, Wo, align_corners, ge_stream(stream));
  return GE_ERR_UNSUPPORTED;
}
// d_out (N, Ho, Wo, C) -> d_in (N, Hi, Wi, C), fully written.  workspace (optional, N * Ho * Wi * C floats): enables the separable
// two-pass form for up-sampling factors > 3.
extern "C" int ge_bilinear_nhwc_bwd(const void* d_out, void* d_in, void* workspace, size_t workspace_bytes, int N, int C, int Hi, int Wi, int Ho,
                                    int Wo, int align_corners, int dtype, void* stream) {
  if (!d_out || !d_in || N < 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return GE_ERR_BAD_ARG;
  if (dtype == GE_F32) return bilinear_nhwc_launch<float>(false, d_out, d_in, N, C, Hi, Wi, Ho, Wo, align_corners, ge_stream(stream), workspace, workspace_bytes);
  if (dtype == GE_BF16) return bilinear_nhwc_launch<bf16_t>(false, d_out, d_in, N, C, Hi, Wi, Ho, Wo, align_corners, ge_stream(stream), workspace, workspace_bytes);
  return GE_ERR_UNSUPPORTED;
}

// ================================================================================================= row concat / glue
// out[r, off_a : off_a + Ca] = a[r, :] * drop(r, c) + res[r, :]   (res may be NULL, p may be 0; `a` = B batches of rpb rows, batch
//                                                                     stride a_bs elements: a token range of a longer sequence)
// out[r, off_b : off_b + Cb] = b[r, :]                              (b may be NULL: only the first part is written)
// rows of `out` have Ca + Cb elements; a, res, b are dense (R, Ca) / (R, Cb).
template <typename T>
__global__ void __launch_bounds__(256) concat_rows_k(const T* __restrict__ a, long rpb, long a_bs, const T* __restrict__ res, const T* __restrict__ b,
                                                     T* __restrict__ out, long R, int Ca, int Cb, int off_a, int off_b, float p, float inv_keep,
                                                     uint64_t seed0, const unsigned long long* __restrict__ salt) {
  const uint64_t seed = ge_salted(seed0, salt);
  const uint32_t thr = ge_drop_threshold(p);
  constexpr int VN = V8<T>::N;
  const int la = Ca / VN, lb = b ? Cb / VN : 0, lt = la + lb, Co = Ca + Cb;
  const long total = R * lt;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / lt;
    const int j = (int)(i - r * lt);
    float v[VN];
    if (j < la) {
      const int c0 = j * VN;
      const long bi = r / rpb;
      V8<T>::ld(a + bi * a_bs + (r - bi * rpb) * Ca + c0, v);                 // `a`: rows packed per batch, free batch stride
      if (p > 0.f) {
#pragma unroll
        for (int k0 = 0; k0 < VN; k0 += 4) {                                  // Ca and c0 are multiples of VN: whole groups of four indices
          float sc[4];
          ge_drop_scale4(seed, (uint64_t)(r * Ca + c0 + k0) >> 2, thr, inv_keep, sc);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[k0 + e] *= sc[e];
        }
      }
      if (res) {
        float q[VN];
        V8<T>::ld(res + r * Ca + c0, q);
#pragma unroll
        for (int k = 0; k < VN; ++k) v[k] = Io<T>::rt(v[k]) + q[k];          // dropout(out) is materialised in T by the reference
      }
      V8<T>::st(out + r * Co + off_a + c0, v);
    } else {
      const int c0 = (j - la) * VN;
      V8<T>::ld(b + r * Cb + c0, v);
      V8<T>::st(out + r * Co + off_b + c0, v);
    }
  }
}
// backward of the first part: d_a[r, :] = d_out[r, off_a : off_a + Ca] * drop(r, c)  (dense (R, Ca) out of rows of Co)
template <typename T>
__global__ void __launch_bounds__(256) slice_rows_drop_k(const T* __restrict__ d_out, T* __restrict__ d_a, long R, int Ca, int Co, int off_a,
                                                         float p, float inv_keep, uint64_t seed0, const unsigned long long* __restrict__ salt) {
  const uint64_t seed = ge_salted(seed0, salt);
  const uint32_t thr = ge_drop_threshold(p);
  constexpr int VN = V8<T>::N;
  const int la = Ca / VN;
  const long total = R * la;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / la;
    const int c0 = (int)(i - r * la) * VN;
    float v[VN];
    V8<T>::ld(d_out + r * Co + off_a + c0, v);
    if (p > 0.f) {
#pragma unroll
      for (int k0 = 0; k0 < VN; k0 += 4) {
        float sc[4];
        ge_drop_scale4(seed, (uint64_t)(r * Ca + c0 + k0) >> 2, thr, inv_keep, sc);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[k0 + e] *= sc[e];
      }
    }
    V8<T>::st(d_a + r * Ca + c0, v);
  }
}
extern "C" int ge_concat_rows_fwd(const void* a, long rows_per_batch, long a_batch_stride, const void* res, const void* b, void* out, long rows,
                                  int Ca, int Cb, int a_first, float p_drop, unsigned long long seed, int dtype, void* stream) {
  if (!a || !out || rows <= 0 || Ca <= 0 || Cb < 0 || (Cb > 0 && !b) || p_drop < 0.f || p_drop >= 1.f) return GE_ERR_BAD_ARG;
  if (rows_per_batch <= 0 || rows % rows_per_batch || a_batch_stride < rows_per_batch * Ca) return GE_ERR_BAD_ARG;
  const int vn = dtype == GE_F32 ? 4 : 8;
  if ((dtype != GE_F32 && dtype != GE_BF16) || Ca % vn || Cb % vn || a_batch_stride % vn || !nh_aligned(a, res, b, out)) return GE_ERR_UNSUPPORTED;
  const int off_a = a_first ? 0 : Cb, off_b = a_first ? Ca : 0;
  const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  const long total = rows * ((Ca + Cb) / vn);
  hipStream_t s = ge_stream(stream);
  if (dtype == GE_F32)
    concat_rows_k<float><<<nh_grid_vec(total), 256, 0, s>>>((const float*)a, rows_per_batch, a_batch_stride, (const float*)res, Cb ? (const float*)b : nullptr, (float*)out, rows, Ca, Cb, off_a, off_b, p_drop, inv_keep, seed, ge_rng_salt_get());
  else
    concat_rows_k<bf16_t><<<nh_grid_vec(total), 256, 0, s>>>((const bf16_t*)a, rows_per_batch, a_batch_stride, (const bf16_t*)res, Cb ? (const bf16_t*)b : nullptr, (bf16_t*)out, rows, Ca, Cb, off_a, off_b, p_drop, inv_keep, seed, ge_rng_salt_get());
  GE_LAUNCH_CHECK();
  return GE_OK;
}
extern "C" int ge_slice_rows_drop(const void* d_out, void* d_a, long rows, int Ca, int Co, int off_a, float p_drop, unsigned long long seed,
                                  int dtype, void* stream) {
  if (!d_out || !d_a || rows <= 0 || Ca <= 0 || Co < Ca || off_a < 0 || off_a + Ca > Co || p_drop < 0.f || p_drop >= 1.f) return GE_ERR_BAD_ARG;
  const int vn = dtype == GE_F32 ? 4 : 8;
  if ((dtype != GE_F32 && dtype != GE_BF16) || Ca % vn || Co % vn || off_a % vn || !nh_aligned(d_out, d_a)) return GE_ERR_UNSUPPORTED;
  const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  hipStream_t s = ge_stream(stream);
  if (dtype == GE_F32)
    slice_rows_drop_k<float><<<nh_grid_vec(rows * (Ca / vn)), 256, 0, s>>>((const float*)d_out, (float*)d_a, rows, Ca, Co, off_a, p_drop, inv_keep, seed, ge_rng_salt_get());
  else
    slice_rows_drop_k<bf16_t><<<nh_grid_vec(rows * (Ca / vn)), 256, 0, s>>>((const bf16_t*)d_out, (bf16_t*)d_a, rows, Ca, Co, off_a, p_drop, inv_keep, seed, ge_rng_salt_get());
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// tokens + positional embedding: out[b, n, :] = x[b, n, :] + pos[n, :] (pos fp32 (N, C), one rounding into T)
template <typename T>
__global__ void __launch_bounds__(256) add_rows_k(const T* __restrict__ x, const float* __restrict__ pos, T* __restrict__ out, long per_batch_vec,
                                                  long total_vec) {
  constexpr int VN = V8<T>::N;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_vec; i += (long)gridDim.x * 256) {
    const long j = i % per_batch_vec;
    float v[VN];
    V8<T>::ld(x + i * VN, v);
#pragma unroll
    for (int k = 0; k < VN; k += 4) {
      const float4 q = *(const float4*)(pos + j * VN + k);
      v[k] += q.x; v[k + 1] += q.y; v[k + 2] += q.z; v[k + 3] += q.w;
    }
    V8<T>::st(out + i * VN, v);
  }
}
extern "C" int ge_add_rows(const void* x, const float* pos, void* out, int B, long N, int C, int dtype, void* stream) {
  if (!x || !pos || !out || B <= 0 || N <= 0 || C <= 0) return GE_ERR_BAD_ARG;
  const int vn = dtype == GE_F32 ? 4 : 8;
  if ((dtype != GE_F32 && dtype != GE_BF16) || C % vn || !nh_aligned(x, pos, out)) return GE_ERR_UNSUPPORTED;
  const long per = N * C / vn, total = per * B;
  hipStream_t s = ge_stream(stream);
  if (dtype == GE_F32) add_rows_k<float><<<nh_grid_vec(total), 256, 0, s>>>((const float*)x, pos, (float*)out, per, total);
  else add_rows_k<bf16_t><<<nh_grid_vec(total), 256, 0, s>>>((const bf16_t*)x, pos, (bf16_t*)out, per, total);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// ============================================================================ column sums of a token matrix
// out[c] (+)= sum_r x[r, c]: the bias gradient of every token Linear (dY is (tokens, C) with 5e4 - 8e5 rows), which ATen runs as a
// generic reduce_kernel.  Columns are split into chunks of <= 256 16-byte lanes (blockIdx.y) so that any C % VN == 0 is served.
template <typename T>
__global__ void __launch_bounds__(256) colsum_k(const T* __restrict__ x, float* __restrict__ ws, int C, long R, int lpr, int rpi) {
  constexpr int VN = V8<T>::N;
  const int t = threadIdx.x, lane = t % lpr, rs = t / lpr;
  const int c0 = (blockIdx.y * lpr + lane) * VN;
  float acc[1][VN];
#pragma unroll
  for (int v = 0; v < VN; ++v) acc[0][v] = 0.f;
  if (rs < rpi)
    for (long r = (long)blockIdx.x * rpi + rs; r < R; r += (long)gridDim.x * rpi) {
      float v[VN];
      V8<T>::ld(x + r * C + c0, v);
#pragma unroll
      for (int k = 0; k < VN; ++k) acc[0][k] += v[k];
    }
  nh_col_commit<1, VN>(acc, ws, C, lpr, rpi, c0);
}
template <typename T>
static int colsum_launch(const void* x, long R, int C, float* out, void* ws, int accumulate, hipStream_t s) {
  constexpr int VN = V8<T>::N;
  if (C % VN || !nh_aligned(x)) return GE_ERR_UNSUPPORTED;
  const int lanes = C / VN;
  int lpr = lanes < NH_MAXLANES ? lanes : NH_MAXLANES;
  while (lanes % lpr) --lpr;                                           // largest divisor of the lane count that fits a workgroup
  const int rpi = NH_MAXLANES / lpr, chunks = lanes / lpr;
  unsigned gx = nh_grid_rows(R, rpi);
  if (chunks > 1) gx = (gx + chunks - 1) / chunks;
  if (!gx) gx = 1;
  colsum_k<T><<<dim3(gx, chunks), 256, 0, s>>>((const T*)x, nh_partials(ws, C, 1), C, R, lpr, rpi);      // R == 0: zero partials
  GE_LAUNCH_CHECK();
  nh_reduce_launch_f32(ws, C, gx, out, accumulate, s);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
// x (R, C) row-major f32 / bf16 -> out (C) f32; workspace: ge_nhwc_workspace(C, 1) bytes; accumulate != 0: out += sums
extern "C" int ge_colsum(const void* x, long R, int C, float* out, void* workspace, int accumulate, int dtype, void* stream) {
  if (!x || !out || !workspace || R < 0 || C <= 0) return GE_ERR_BAD_ARG;
  if (dtype == GE_F32) return colsum_launch<float>(x, R, C, out, workspace, accumulate, ge_stream(stream));
  if (dtype == GE_BF16) return colsum_launch<bf16_t>(x, R, C, out, workspace, accumulate, ge_stream(stream));
  return GE_ERR_UNSUPPORTED;
}

// ============================================================================ bias + GELU epilogue of the FFN's first Linear
// mmcv FFN (depthformer_swin.py:451-459): Linear(C, 4C) -> GELU (exact, erf) -> Linear(4C, C).  The first GEMM runs bias-free; this
// pass adds the bias and applies GELU (one read, one write, like ATen's gelu kernel alone), and the backward pass recomputes the
// pre-activation from the same two inputs, multiplies by GELU' AND accumulates the bias gradient (column sums of d_y) in the same
// sweep — ATen + the Linear's own bias gradient: gelu_backward (2 reads + 1 write) and a column-sum pass (1 more read of d_y).
// fp32 storage (the parity path): libm erff / expf.  bf16 storage: the library erff is ~80 VALU instructions with branches and made this
// pass COMPUTE-bound (197120 x 384: 160 us = 1.9 TB/s); Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below bf16's 2^-9) needs one
// v_rcp, one v_exp and a degree-5 Horner chain, and shares the exponential with the density term of the derivative.
template <typename T> struct Gelu;
template <> struct Gelu<float> {
  static __device__ __forceinline__ float f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
  static __device__ __forceinline__ float grad(float x) {
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
    return cdf + x * 0.3989422804014327f * expf(-0.5f * x * x);
  }
};
template <> struct Gelu<bf16_t> {
  static __device__ __forceinline__ void parts(float x, float& cdf, float& e) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
    e = __expf(-z * z);                                                  // = exp(-x^2 / 2)
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f); p = fmaf(p, t, -0.284496736f); p = fmaf(p, t, 0.254829592f);
    const float erfc_half = 0.5f * p * t * e;                            // (1 - erf(z)) / 2
    cdf = x >= 0.f ? 1.f - erfc_half : erfc_half;
  }
  static __device__ __forceinline__ float f(float x) { float c, e; parts(x, c, e); return x * c; }
  static __device__ __forceinline__ float grad(float x) { float c, e; parts(x, c, e); return fmaf(x * 0.3989422804014327f, e, c); }
};
template <typename T>
__global__ void __launch_bounds__(256) bias_gelu_fwd_k(const T* __restrict__ x, const float* __restrict__ bias, T* __restrict__ out, int C,
                                                       long R, int lpr, int rpi) {
  constexpr int VN = V8<T>::N;                                  // thread = (row slot, channel vector): the bias vector stays in registers
  const int t = threadIdx.x, lane = t % lpr, rs = t / lpr;
  if (rs >= rpi) return;
  const int c0 = (blockIdx.y * lpr + lane) * VN;
  float b[VN];
#pragma unroll
  for (int k = 0; k < VN; ++k) b[k] = bias ? bias[c0 + k] : 0.f;
  for (long r = (long)blockIdx.x * rpi + rs; r < R; r += (long)gridDim.x * rpi) {
    float v[VN];
    V8<T>::ld(x + r * C + c0, v);
#pragma unroll
    for (int k = 0; k < VN; ++k) v[k] = Gelu<T>::f(v[k] + b[k]);
    V8<T>::st(out + r * C + c0, v);
  }
}
template <typename T>
__global__ void __launch_bounds__(256) bias_gelu_bwd_k(const T* __restrict__ dg, const T* __restrict__ x, const float* __restrict__ bias,
                                                       T* __restrict__ dy, float* __restrict__ ws, int C, long R, int lpr, int rpi) {
  constexpr int VN = V8<T>::N;
  const int t = threadIdx.x, lane = t % lpr, rs = t / lpr;
  const int c0 = (blockIdx.y * lpr + lane) * VN;
  float acc[1][VN], b[VN];
#pragma unroll
  for (int v = 0; v < VN; ++v) { acc[0][v] = 0.f; b[v] = bias ? bias[c0 + v] : 0.f; }
  if (rs < rpi)
    for (long r = (long)blockIdx.x * rpi + rs; r < R; r += (long)gridDim.x * rpi) {
      float g[VN], v[VN];
      V8<T>::ld(dg + r * C + c0, g);
      V8<T>::ld(x + r * C + c0, v);
#pragma unroll
      for (int k = 0; k < VN; ++k) {
        g[k] = Io<T>::rt(g[k] * Gelu<T>::grad(v[k] + b[k]));          // the bias gradient sums what the GEMMs downstream read
        acc[0][k] += g[k];
      }
      V8<T>::st(dy + r * C + c0, g);
    }
  nh_col_commit<1, VN>(acc, ws, C, lpr, rpi, c0);
}
// out = gelu(x + bias) over a (R, C) row matrix (f32 / bf16, C a multiple of the 16-byte vector); bias f32 (C) or NULL
template <typename T>
static int bias_gelu_fwd_launch(const void* x, const float* bias, void* out, long R, int C, hipStream_t s) {
  constexpr int VN = V8<T>::N;
  if (C % VN || !nh_aligned(x, out)) return GE_ERR_UNSUPPORTED;
  const int lanes = C / VN;
  int lpr = lanes < NH_MAXLANES ? lanes : NH_MAXLANES;
  while (lanes % lpr) --lpr;
  const int rpi = NH_MAXLANES / lpr, chunks = lanes / lpr;
  long gx = (R + (long)rpi * 4 - 1) / ((long)rpi * 4);           // ~4 rows per thread
  if (gx < 1) gx = 1;
  if (gx > 16384) gx = 16384;
  bias_gelu_fwd_k<T><<<dim3((unsigned)gx, chunks), 256, 0, s>>>((const T*)x, bias, (T*)out, C, R, lpr, rpi);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
extern "C" int ge_bias_gelu_fwd(const void* x, const float* bias, void* out, long R, int C, int dtype, void* stream) {
  if (!x || !out || R < 0 || C <= 0) return GE_ERR_BAD_ARG;
  if (R == 0) return GE_OK;
  if (dtype == GE_F32) return bias_gelu_fwd_launch<float>(x, bias, out, R, C, ge_stream(stream));
  if (dtype == GE_BF16) return bias_gelu_fwd_launch<bf16_t>(x, bias, out, R, C, ge_stream(stream));
  return GE_ERR_UNSUPPORTED;
}
// dy = dg * gelu'(x + bias), d_bias (C) f32 = column sums of dy; workspace: ge_nhwc_workspace(C, 1) bytes
template <typename T>
static int bias_gelu_bwd_launch(const void* dg, const void* x, const float* bias, void* dy, float* d_bias, void* ws, long R, int C, hipStream_t s) {
  constexpr int VN = V8<T>::N;
  if (C % VN || !nh_aligned(dg, x, dy)) return GE_ERR_UNSUPPORTED;
  const int lanes = C / VN;
  int lpr = lanes < NH_MAXLANES ? lanes : NH_MAXLANES;
  while (lanes % lpr) --lpr;
  const int rpi = NH_MAXLANES / lpr, chunks = lanes / lpr;
  unsigned gx = nh_grid_rows(R, rpi);
  if (chunks > 1) gx = (gx + chunks - 1) / chunks;
  if (!gx) gx = 1;
  bias_gelu_bwd_k<T><<<dim3(gx, chunks), 256, 0, s>>>((const T*)dg, (const T*)x, bias, (T*)dy, nh_partials(ws, C, 1), C, R, lpr, rpi);
  GE_LAUNCH_CHECK();
  nh_reduce_launch_f32(ws, C, gx, d_bias, 0, s);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
extern "C" int ge_bias_gelu_bwd(const void* dg, const void* x, const float* bias, void* dy, float* d_bias, void* workspace, long R, int C,
                                int dtype, void* stream) {
  if (!dg || !x || !dy || !d_bias || !workspace || R < 0 || C <= 0) return GE_ERR_BAD_ARG;
  if (dtype == GE_F32) return bias_gelu_bwd_launch<float>(dg, x, bias, dy, d_bias, workspace, R, C, ge_stream(stream));
  if (dtype == GE_BF16) return bias_gelu_bwd_launch<bf16_t>(dg, x, bias, dy, d_bias, workspace, R, C, ge_stream(stream));
  return GE_ERR_UNSUPPORTED;
}
