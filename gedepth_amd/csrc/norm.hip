// LayerNorm over the last dimension of a (rows, C) token matrix, mixed precision: x is f32 or bf16, statistics and
// affine parameters f32, y is f32 or bf16.  Replaces, on the Swin token path (depthformer_swin.py:461-472 norm1 / norm2,
// :98-122 PatchMerging.norm, :1166-1172 per-stage output norms; models/utils/embed.py:282-302 patch-embed norm), what
// autocast makes of `F.layer_norm`: a bf16 -> f32 copy of the input, the f32 LayerNorm, and an f32 -> bf16 copy of the
// output in front of the next Linear (and three kernels plus the same copies in the backward pass).
//
// One row per lane group of GS lanes (16 / 32 / 64 by C), 4 elements per lane and step kept in registers, two-pass
// mean / variance in f32, shuffles only (no LDS); HBM-bound: x read once, y written once.
// Backward: dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma, per row in the same layout;
// dgamma / dbeta are column sums over rows: per-lane partial sums over the rows a workgroup visits, combined across the
// workgroup's row groups through LDS, then one f32 atomic per column and workgroup.
#include "common.h"

#define LN_MAX_CH 12            // 4-element chunks per lane: C <= 4 * GS * 12 (3072 at GS = 64)

template <typename T> struct Q4;      // 4 consecutive elements <-> 4 floats
template <> struct Q4<float> {
  static __device__ __forceinline__ void ld(const float* p, float v[4]) { const float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  static __device__ __forceinline__ void st(float* p, const float v[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Q4<bf16_t> {
  static __device__ __forceinline__ void ld(const bf16_t* p, float v[4]) {
    const uint2 t = *(const uint2*)p;
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
  }
  static __device__ __forceinline__ void st(bf16_t* p, const float v[4]) {
    uint2 t;
    t.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    t.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    *(uint2*)p = t;
  }
};

template <int GS>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = GS / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// grid-stride over rows; C % 4 == 0; NCH = ceil(C / (4 * GS)) <= LN_MAX_CH
template <typename TX, typename TY, int GS, int NCH>
__global__ void __launch_bounds__(256) layernorm_fwd_k(const TX* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, TY* __restrict__ y, float* __restrict__ mean,
                                                       float* __restrict__ rstd, long rows, int C, float eps) {
  constexpr int RPB = 256 / GS;
  const int sub = threadIdx.x % GS;
  const float inv_c = 1.f / (float)C;
  for (long row = (long)blockIdx.x * RPB + threadIdx.x / GS; row < rows; row += (long)gridDim.x * RPB) {
    const TX* xp = x + row * C;
    float v[NCH][4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = (j * GS + sub) * 4;
      if (c < C) { Q4<TX>::ld(xp + c, v[j]); s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]); }
      else { v[j][0] = v[j][1] = v[j][2] = v[j][3] = 0.f; }
    }
    const float mu = group_sum<GS>(s) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = (j * GS + sub) * 4;
      if (c < C) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float d = v[j][k] - mu; q += d * d; }
      }
    }
    const float rs = rsqrtf(group_sum<GS>(q) * inv_c + eps);
    TY* yp = y + row * C;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = (j * GS + sub) * 4;
      if (c < C) {
        const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
        float o[4] = {(v[j][0] - mu) * rs * g.x + b.x, (v[j][1] - mu) * rs * g.y + b.y,
                      (v[j][2] - mu) * rs * g.z + b.z, (v[j][3] - mu) * rs * g.w + b.w};
        Q4<TY>::st(yp + c, o);
      }
    }
    if (sub == 0) { mean[row] = mu; rstd[row] = rs; }
  }
}

template <typename TX, typename TY, int GS, int NCH>
__global__ void __launch_bounds__(256) layernorm_bwd_k(const TY* __restrict__ dy, const TX* __restrict__ x,
                                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, TX* __restrict__ dx,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta, long rows, int C,
                                                       int use_lds, const TX* __restrict__ dres, int copies) {
  constexpr int RPB = 256 / GS;
  // copies > 1: dgamma points at [copies][2][C] floats (dbeta = dgamma + C): workgroup w adds into copy w % copies — same-address fp32
  // atomics serialise in L2 (1024 workgroups on one copy: ~22 us per launch); the CALLER sums the copies.  (Folding them in the last
  // workgroup to finish was tried: the device-scope fence it needs writes the XCD's dirty L2 lines back — the kernel's own dx stream —
  // and doubled the kernel time: 141 vs 68 us at 197120 x 96.)
  const long copy_off = copies > 1 ? (long)(blockIdx.x % copies) * 2 * C : 0;
  dgamma += copy_off; dbeta += copy_off;
  extern __shared__ float red[];                       // [RPB][2][C] when use_lds (column sums across the row groups)
  const int sub = threadIdx.x % GS, rg = threadIdx.x / GS;
  const float inv_c = 1.f / (float)C;
  float ag[NCH][4], ab[NCH][4];                        // this lane's columns: sum dy * xhat, sum dy
#pragma unroll
  for (int j = 0; j < NCH; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) { ag[j][k] = 0.f; ab[j][k] = 0.f; }
  for (long row = (long)blockIdx.x * RPB + rg; row < rows; row += (long)gridDim.x * RPB) {
    const TX* xp = x + row * C;
    const TY* gp = dy + row * C;
    const float mu = mean[row], rs = rstd[row];
    float xh[NCH][4], g[NCH][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = (j * GS + sub) * 4;
      if (c < C) {
        float xv[4], gv[4];
        Q4<TX>::ld(xp + c, xv);
        Q4<TY>::ld(gp + c, gv);
        const float4 gm = *(const float4*)(gamma + c);
        const float gmv[4] = {gm.x, gm.y, gm.z, gm.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          xh[j][k] = (xv[k] - mu) * rs;
          ag[j][k] += gv[k] * xh[j][k];
          ab[j][k] += gv[k];
          g[j][k] = gv[k] * gmv[k];
          s1 += g[j][k];
          s2 += g[j][k] * xh[j][k];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { xh[j][k] = 0.f; g[j][k] = 0.f; }
      }
    }
    const float m1 = group_sum<GS>(s1) * inv_c, m2 = group_sum<GS>(s2) * inv_c;
    TX* dp = dx + row * C;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = (j * GS + sub) * 4;
      if (c < C) {
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = rs * (g[j][k] - m1 - xh[j][k] * m2);
        if (dres) {                                      // the gradient that reached x through the residual branch: summed here, not by autograd
          float q[4];
          Q4<TX>::ld(dres + row * C + c, q);
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] += q[k];
        }
        Q4<TX>::st(dp + c, o);
      }
    }
  }
  // column sums: combine the RPB row groups of the workgroup in LDS, then one atomic per column
  if (use_lds) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = (j * GS + sub) * 4;
      if (c < C) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[(rg * 2 + 0) * C + c + k] = ag[j][k]; red[(rg * 2 + 1) * C + c + k] = ab[j][k]; }
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += 256) {
      const int which = c / C, col = c - which * C;
      float t = 0.f;
      for (int r = 0; r < RPB; ++r) t += red[(r * 2 + which) * C + col];
      atomicAdd((which ? dbeta : dgamma) + col, t);
    }
  } else {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = (j * GS + sub) * 4;
      if (c < C) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { atomicAdd(dgamma + c + k, ag[j][k]); atomicAdd(dbeta + c + k, ab[j][k]); }
      }
    }
  }
}

struct LnPlan { int gs, nch; };
static bool ln_plan(int C, LnPlan& p) {
  if (C <= 0 || C % 4) return false;
  p.gs = C <= 128 ? 16 : (C <= 512 ? 32 : 64);
  p.nch = (C + 4 * p.gs - 1) / (4 * p.gs);
  return p.nch <= LN_MAX_CH;
}

// NCH is a template parameter (register arrays): instantiate the bucket that covers it
#define LN_DISPATCH_NCH(CALL)                         \
  if (pl.nch <= 2) { CALL(2); }                       \
  else if (pl.nch <= 4) { CALL(4); }                  \
  else if (pl.nch <= 6) { CALL(6); }                  \
  else { CALL(LN_MAX_CH); }

template <typename TX, typename TY>
static int ln_fwd_launch(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, long rows, int C,
                         float eps, hipStream_t s) {
  LnPlan pl;
  if (!ln_plan(C, pl)) return GE_ERR_UNSUPPORTED;
  const int rpb = 256 / pl.gs;
  const unsigned blocks = ge_blocks(rows, rpb * 2, 256 * 64);
#define LN_F(GS_, N_) layernorm_fwd_k<TX, TY, GS_, N_><<<blocks, 256, 0, s>>>((const TX*)x, gamma, beta, (TY*)y, mean, rstd, rows, C, eps)
#define LN_F16(N_) LN_F(16, N_)
#define LN_F32(N_) LN_F(32, N_)
#define LN_F64(N_) LN_F(64, N_)
  if (pl.gs == 16) { LN_DISPATCH_NCH(LN_F16) }
  else if (pl.gs == 32) { LN_DISPATCH_NCH(LN_F32) }
  else { LN_DISPATCH_NCH(LN_F64) }
#undef LN_F
#undef LN_F16
#undef LN_F32
#undef LN_F64
  GE_LAUNCH_CHECK();
  return GE_OK;
}
template <typename TX, typename TY>
static int ln_bwd_launch(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                         float* dgamma, float* dbeta, long rows, int C, hipStream_t s, const void* dres = nullptr, int copies = 1) {
  LnPlan pl;
  if (!ln_plan(C, pl)) return GE_ERR_UNSUPPORTED;
  const int rpb = 256 / pl.gs;
  // few, long-lived workgroups keep the column atomics down (1024 workgroups: ~22 us of serialised adds per launch) — but a workgroup walks
  // its rows rpb at a time with a full memory round trip per step, so small inputs must not be dealt 8 steps to 100 workgroups
  // (3080 x 768: 35 us for 14 MB): aim at ~768 workgroups, between 1 and 8 steps each
  long per = (rows + 767) / 768;
  per = (per + rpb - 1) / rpb * rpb;
  if (per < rpb) per = rpb;
  if (per > rpb * 8) per = rpb * 8;
  const unsigned blocks = ge_blocks(rows, (int)per, 256 * 4);
  size_t smem = (size_t)rpb * 2 * C * sizeof(float);
  const int use_lds = smem <= 60 * 1024;                               // else: per-lane column atomics (only LN(3072), few rows)
  if (!use_lds) smem = 0;
#define LN_B(GS_, N_) layernorm_bwd_k<TX, TY, GS_, N_><<<blocks, 256, smem, s>>>((const TY*)dy, (const TX*)x, gamma, mean, rstd, (TX*)dx, dgamma, dbeta, rows, C, use_lds, (const TX*)dres, copies)
#define LN_B16(N_) LN_B(16, N_)
#define LN_B32(N_) LN_B(32, N_)
#define LN_B64(N_) LN_B(64, N_)
  if (pl.gs == 16) { LN_DISPATCH_NCH(LN_B16) }
  else if (pl.gs == 32) { LN_DISPATCH_NCH(LN_B32) }
  else { LN_DISPATCH_NCH(LN_B64) }
#undef LN_B
#undef LN_B16
#undef LN_B32
#undef LN_B64
  GE_LAUNCH_CHECK();
  return GE_OK;
}

extern "C" int ge_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                                float* mean, float* rstd, long rows, int C, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows < 0 || C <= 0) return GE_ERR_BAD_ARG;
  if (rows == 0) return GE_OK;
  hipStream_t s = ge_stream(stream);
  if (x_dtype == GE_F32 && y_dtype == GE_F32) return ln_fwd_launch<float, float>(x, gamma, beta, y, mean, rstd, rows, C, eps, s);
  if (x_dtype == GE_F32 && y_dtype == GE_BF16) return ln_fwd_launch<float, bf16_t>(x, gamma, beta, y, mean, rstd, rows, C, eps, s);
  if (x_dtype == GE_BF16 && y_dtype == GE_BF16) return ln_fwd_launch<bf16_t, bf16_t>(x, gamma, beta, y, mean, rstd, rows, C, eps, s);
  if (x_dtype == GE_BF16 && y_dtype == GE_F32) return ln_fwd_launch<bf16_t, float>(x, gamma, beta, y, mean, rstd, rows, C, eps, s);
  return GE_ERR_UNSUPPORTED;
}

extern "C" int ge_layernorm_bwd(const void* dy, int y_dtype, const void* x, int x_dtype, const float* gamma, const float* mean,
                                const float* rstd, void* dx, float* dgamma, float* dbeta, long rows, int C, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || rows < 0 || C <= 0) return GE_ERR_BAD_ARG;
  if (rows == 0) return GE_OK;
  hipStream_t s = ge_stream(stream);
  if (x_dtype == GE_F32 && y_dtype == GE_F32) return ln_bwd_launch<float, float>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, s);
  if (x_dtype == GE_F32 && y_dtype == GE_BF16) return ln_bwd_launch<float, bf16_t>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, s);
  if (x_dtype == GE_BF16 && y_dtype == GE_BF16) return ln_bwd_launch<bf16_t, bf16_t>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, s);
  if (x_dtype == GE_BF16 && y_dtype == GE_F32) return ln_bwd_launch<bf16_t, float>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, s);
  return GE_ERR_UNSUPPORTED;
}

// ge_layernorm_bwd with the residual gradient folded in: dx = LN'(dy) + dres (dres in x's dtype, may be NULL).  In a pre-norm block
// x feeds both the LayerNorm and the skip connection; autograd would add the two gradients in a separate pass over the token tensor.
extern "C" int ge_layernorm_bwd_res(const void* dy, int y_dtype, const void* x, int x_dtype, const float* gamma, const float* mean,
                                    const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta, long rows, int C, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || rows < 0 || C <= 0) return GE_ERR_BAD_ARG;
  if (rows == 0) return GE_OK;
  hipStream_t s = ge_stream(stream);
  if (x_dtype == GE_F32 && y_dtype == GE_F32) return ln_bwd_launch<float, float>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, s, dres);
  if (x_dtype == GE_F32 && y_dtype == GE_BF16) return ln_bwd_launch<float, bf16_t>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, s, dres);
  if (x_dtype == GE_BF16 && y_dtype == GE_BF16) return ln_bwd_launch<bf16_t, bf16_t>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, s, dres);
  if (x_dtype == GE_BF16 && y_dtype == GE_F32) return ln_bwd_launch<bf16_t, float>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, s, dres);
  return GE_ERR_UNSUPPORTED;
}

// ge_layernorm_bwd_res with the column sums spread over `copies` accumulators: dgb = [copies][2][C] floats (gamma row, beta row), zero-filled by
// the caller; d_gamma = sum over copies of dgb[k][0][:], d_beta = sum of dgb[k][1][:] (the caller's one small reduction).
extern "C" int ge_layernorm_bwd_multi(const void* dy, int y_dtype, const void* x, int x_dtype, const float* gamma, const float* mean,
                                      const float* rstd, const void* dres, void* dx, float* dgb, int copies, long rows, int C, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgb || rows < 0 || C <= 0 || copies < 1 || copies > 64) return GE_ERR_BAD_ARG;
  if (rows == 0) return GE_OK;
  hipStream_t s = ge_stream(stream);
  float* dg = dgb;
  float* db = dgb + C;
  if (x_dtype == GE_F32 && y_dtype == GE_F32) return ln_bwd_launch<float, float>(dy, x, gamma, mean, rstd, dx, dg, db, rows, C, s, dres, copies);
  if (x_dtype == GE_F32 && y_dtype == GE_BF16) return ln_bwd_launch<float, bf16_t>(dy, x, gamma, mean, rstd, dx, dg, db, rows, C, s, dres, copies);
  if (x_dtype == GE_BF16 && y_dtype == GE_BF16) return ln_bwd_launch<bf16_t, bf16_t>(dy, x, gamma, mean, rstd, dx, dg, db, rows, C, s, dres, copies);
  if (x_dtype == GE_BF16 && y_dtype == GE_F32) return ln_bwd_launch<bf16_t, float>(dy, x, gamma, mean, rstd, dx, dg, db, rows, C, s, dres, copies);
  return GE_ERR_UNSUPPORTED;
}

// The caller's reduction of ge_layernorm_bwd_multi's accumulator copies, leaving them ZERO again: out[i] = sum_k dgb[k][i], dgb[k][i] = 0 for
// i < 2 C.  With a persistent, initially zeroed accumulator buffer per width a LayerNorm backward is two launches (kernel + fold) instead of
// zero-fill + kernel + sum.
__global__ void __launch_bounds__(256) layernorm_fold_k(float* __restrict__ dgb, float* __restrict__ out, int copies, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float a = 0.f;
  for (int k = 0; k < copies; ++k) { a += dgb[(long)k * n + i]; dgb[(long)k * n + i] = 0.f; }
  out[i] = a;
}
extern "C" int ge_layernorm_fold(float* dgb, float* out, int copies, int C, void* stream) {
  if (!dgb || !out || copies < 1 || copies > 64 || C <= 0) return GE_ERR_BAD_ARG;
  layernorm_fold_k<<<(2 * C + 255) / 256, 256, 0, ge_stream(stream)>>>(dgb, out, copies, 2 * C);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// ============================================================================ BatchNorm2d (training) + ReLU, NCHW
// conv -> BN -> ReLU of mmcv ConvModule (necks/hahi.py:150-166 lateral / proj / fusion convs, depthformer_swin.py:1127-1139
// stem).  MIOpen's spatial BN plus ATen's clamp / threshold_backward make four kernels and ~10 passes over the map; here:
//   forward : bn_stats_k (one read: per-channel sum, sum of squares; fp32 per thread, fp64 across threads) ->
//             bn_finalize_k (mean, rstd, running statistics, per-channel scale / shift) -> bn_apply_k (read x, write y)
//   backward: bn_bwd_stats_k (read dy, y, x: sum g, sum g * xhat with g = dy * relu'(y)) -> bn_bwd_finalize_k ->
//             bn_bwd_apply_k (read dy, y, x, write dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)))
// HBM-bound streaming kernels, 16-byte vectors, one (n, c) plane per blockIdx.y like bias_act.
template <typename T, bool VEC>
__global__ void __launch_bounds__(256) bn_stats_k(const T* __restrict__ x, double* __restrict__ ws, int C, long HW) {
  const long plane = blockIdx.y;
  const T* p = x + plane * HW;
  constexpr int VN = V8<T>::N;
  float s = 0.f, ss = 0.f;
  if (VEC) {
    const long nv = HW / VN;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
      float v[VN];
      V8<T>::ld(p + i * VN, v);
#pragma unroll
      for (int k = 0; k < VN; ++k) { s += v[k]; ss += v[k] * v[k]; }
    }
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) {
      const float v = Io<T>::ld(p + i);
      s += v; ss += v * v;
    }
  }
  double ds = s, dss = ss;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { ds += __shfl_xor(ds, o, 64); dss += __shfl_xor(dss, o, 64); }
  __shared__ double sm[8];
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = ds; sm[4 + (threadIdx.x >> 6)] = dss; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int c = (int)(plane % C);
    atomicAdd(&ws[c], sm[0] + sm[1] + sm[2] + sm[3]);
    atomicAdd(&ws[C + c], sm[4] + sm[5] + sm[6] + sm[7]);
  }
}

// ws: [0,C) sum, [C,2C) sum of squares (in) ; coef: [0,C) scale = gamma * rstd, [C,2C) shift = beta - mean * scale (out)
__global__ void __launch_bounds__(256) bn_finalize_k(const double* __restrict__ ws, const float* __restrict__ gamma,
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
  coef[C + c] = beta[c] - (float)mean * a;
}

template <typename T, bool VEC>
__global__ void __launch_bounds__(256) bn_apply_k(const T* __restrict__ x, const float* __restrict__ coef, T* __restrict__ y,
                                                  int C, long HW, float slope) {
  const long plane = blockIdx.y;
  const int c = (int)(plane % C);
  const float a = coef[c], b = coef[C + c];
  const T* p = x + plane * HW;
  T* q = y + plane * HW;
  constexpr int VN = V8<T>::N;
  if (VEC) {
    const long nv = HW / VN;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
      float v[VN];
      V8<T>::ld(p + i * VN, v);
#pragma unroll
      for (int k = 0; k < VN; ++k) { const float t = v[k] * a + b; v[k] = t > 0.f ? t : t * slope; }
      V8<T>::st(q + i * VN, v);
    }
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) {
      const float t = Io<T>::ld(p + i) * a + b;
      Io<T>::st(q + i, t > 0.f ? t : t * slope);
    }
  }
}

template <typename T, bool VEC>
__global__ void __launch_bounds__(256) bn_bwd_stats_k(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ x,
                                                      const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
                                                      double* __restrict__ ws, int C, long HW, float slope) {
  const long plane = blockIdx.y;
  const int c = (int)(plane % C);
  const float mu = save_mean[c], rs = save_rstd[c];
  const T* gp = dy + plane * HW;
  const T* yp = y + plane * HW;
  const T* xp = x + plane * HW;
  constexpr int VN = V8<T>::N;
  float s1 = 0.f, s2 = 0.f;
  if (VEC) {
    const long nv = HW / VN;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
      float g[VN], yv[VN], xv[VN];
      V8<T>::ld(gp + i * VN, g);
      V8<T>::ld(yp + i * VN, yv);
      V8<T>::ld(xp + i * VN, xv);
#pragma unroll
      for (int k = 0; k < VN; ++k) {
        const float gg = yv[k] > 0.f ? g[k] : g[k] * slope;
        s1 += gg;
        s2 += gg * ((xv[k] - mu) * rs);
      }
    }
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) {
      float gg = Io<T>::ld(gp + i);
      gg = Io<T>::ld(yp + i) > 0.f ? gg : gg * slope;
      s1 += gg;
      s2 += gg * ((Io<T>::ld(xp + i) - mu) * rs);
    }
  }
  double d1 = s1, d2 = s2;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { d1 += __shfl_xor(d1, o, 64); d2 += __shfl_xor(d2, o, 64); }
  __shared__ double sm[8];
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = d1; sm[4 + (threadIdx.x >> 6)] = d2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&ws[c], sm[0] + sm[1] + sm[2] + sm[3]);
    atomicAdd(&ws[C + c], sm[4] + sm[5] + sm[6] + sm[7]);
  }
}

// ws: sum g, sum g * xhat -> dbeta, dgamma and the per-channel terms of dx: coef = {gamma * rstd, mean(g), mean(g * xhat)}
__global__ void __launch_bounds__(256) bn_bwd_finalize_k(const double* __restrict__ ws, const float* __restrict__ gamma,
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

template <typename T, bool VEC>
__global__ void __launch_bounds__(256) bn_bwd_apply_k(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ x,
                                                      const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
                                                      const float* __restrict__ coef, T* __restrict__ dx, int C, long HW,
                                                      float slope) {
  const long plane = blockIdx.y;
  const int c = (int)(plane % C);
  const float mu = save_mean[c], rs = save_rstd[c], a = coef[c], m1 = coef[C + c], m2 = coef[2 * C + c];
  const T* gp = dy + plane * HW;
  const T* yp = y + plane * HW;
  const T* xp = x + plane * HW;
  T* dp = dx + plane * HW;
  constexpr int VN = V8<T>::N;
  if (VEC) {
    const long nv = HW / VN;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
      float g[VN], yv[VN], xv[VN];
      V8<T>::ld(gp + i * VN, g);
      V8<T>::ld(yp + i * VN, yv);
      V8<T>::ld(xp + i * VN, xv);
#pragma unroll
      for (int k = 0; k < VN; ++k) {
        const float gg = yv[k] > 0.f ? g[k] : g[k] * slope;
        g[k] = a * (gg - m1 - ((xv[k] - mu) * rs) * m2);
      }
      V8<T>::st(dp + i * VN, g);
    }
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) {
      float gg = Io<T>::ld(gp + i);
      gg = Io<T>::ld(yp + i) > 0.f ? gg : gg * slope;
      Io<T>::st(dp + i, a * (gg - m1 - ((Io<T>::ld(xp + i) - mu) * rs) * m2));
    }
  }
}

static inline dim3 bn_grid(int N, int C, long HW, int vn, bool vec) {
  const long per_plane = vec ? HW / vn : HW;
  long gx = (per_plane + 256 * 4 - 1) / (256 * 4);
  if (gx < 1) gx = 1;
  if (gx > 64) gx = 64;
  return dim3((unsigned)gx, (unsigned)(N * C));
}
static inline bool bn_vec(long HW, int vn, const void* a, const void* b, const void* c, const void* d) {
  return HW % vn == 0 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d)) & 15) == 0;
}

extern "C" size_t ge_bn_workspace(int C) { return (size_t)C * (2 * sizeof(double) + 3 * sizeof(float)); }

template <typename T>
static int bn_fwd_launch(const void* x, const float* gamma, const float* beta, void* y, float* save_mean, float* save_rstd,
                         float* running_mean, float* running_var, void* workspace, int N, int C, long HW, float eps,
                         float momentum, float slope, hipStream_t s) {
  double* ws = (double*)workspace;
  float* coef = (float*)(ws + 2 * C);
  hipError_t he = hipMemsetAsync(ws, 0, (size_t)C * 2 * sizeof(double), s);
  if (he != hipSuccess) return (int)he;
  const bool vec = bn_vec(HW, V8<T>::N, x, y, nullptr, nullptr);
  const dim3 grid = bn_grid(N, C, HW, V8<T>::N, vec);
  if (vec) bn_stats_k<T, true><<<grid, 256, 0, s>>>((const T*)x, ws, C, HW);
  else bn_stats_k<T, false><<<grid, 256, 0, s>>>((const T*)x, ws, C, HW);
  GE_LAUNCH_CHECK();
  bn_finalize_k<<<(C + 255) / 256, 256, 0, s>>>(ws, gamma, beta, save_mean, save_rstd, running_mean, running_var, coef, C,
                                               (double)N * (double)HW, eps, momentum);
  GE_LAUNCH_CHECK();
  if (vec) bn_apply_k<T, true><<<grid, 256, 0, s>>>((const T*)x, coef, (T*)y, C, HW, slope);
  else bn_apply_k<T, false><<<grid, 256, 0, s>>>((const T*)x, coef, (T*)y, C, HW, slope);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
template <typename T>
static int bn_bwd_launch(const void* dy, const void* y, const void* x, const float* gamma, const float* save_mean,
                         const float* save_rstd, void* dx, float* dgamma, float* dbeta, void* workspace, int N, int C, long HW,
                         float slope, hipStream_t s) {
  double* ws = (double*)workspace;
  float* coef = (float*)(ws + 2 * C);
  hipError_t he = hipMemsetAsync(ws, 0, (size_t)C * 2 * sizeof(double), s);
  if (he != hipSuccess) return (int)he;
  const bool vec = bn_vec(HW, V8<T>::N, dy, y, x, dx);
  const dim3 grid = bn_grid(N, C, HW, V8<T>::N, vec);
  if (vec) bn_bwd_stats_k<T, true><<<grid, 256, 0, s>>>((const T*)dy, (const T*)y, (const T*)x, save_mean, save_rstd, ws, C, HW, slope);
  else bn_bwd_stats_k<T, false><<<grid, 256, 0, s>>>((const T*)dy, (const T*)y, (const T*)x, save_mean, save_rstd, ws, C, HW, slope);
  GE_LAUNCH_CHECK();
  bn_bwd_finalize_k<<<(C + 255) / 256, 256, 0, s>>>(ws, gamma, save_rstd, dgamma, dbeta, coef, C, (double)N * (double)HW);
  GE_LAUNCH_CHECK();
  if (vec) bn_bwd_apply_k<T, true><<<grid, 256, 0, s>>>((const T*)dy, (const T*)y, (const T*)x, save_mean, save_rstd, coef, (T*)dx, C, HW, slope);
  else bn_bwd_apply_k<T, false><<<grid, 256, 0, s>>>((const T*)dy, (const T*)y, (const T*)x, save_mean, save_rstd, coef, (T*)dx, C, HW, slope);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

extern "C" int ge_bn_act_fwd(const void* x, const float* gamma, const float* beta, void* y, float* save_mean, float* save_rstd,
                             float* running_mean, float* running_var, void* workspace, int N, int C, long HW, float eps,
                             float momentum, float slope, int dtype, void* stream) {
  if (!x || !gamma || !beta || !y || !save_mean || !save_rstd || !workspace || N <= 0 || C <= 0 || HW <= 0) return GE_ERR_BAD_ARG;
  if ((long)N * C > 2147483647L) return GE_ERR_BAD_ARG;
  if (dtype == GE_F32) return bn_fwd_launch<float>(x, gamma, beta, y, save_mean, save_rstd, running_mean, running_var, workspace, N, C, HW, eps, momentum, slope, ge_stream(stream));
  if (dtype == GE_BF16) return bn_fwd_launch<bf16_t>(x, gamma, beta, y, save_mean, save_rstd, running_mean, running_var, workspace, N, C, HW, eps, momentum, slope, ge_stream(stream));
  return GE_ERR_UNSUPPORTED;
}
extern "C" int ge_bn_act_bwd(const void* dy, const void* y, const void* x, const float* gamma, const float* save_mean,
                             const float* save_rstd, void* dx, float* dgamma, float* dbeta, void* workspace, int N, int C, long HW,
                             float slope, int dtype, void* stream) {
  if (!dy || !y || !x || !gamma || !save_mean || !save_rstd || !dx || !dgamma || !dbeta || !workspace || N <= 0 || C <= 0 || HW <= 0)
    return GE_ERR_BAD_ARG;
  if ((long)N * C > 2147483647L) return GE_ERR_BAD_ARG;
  if (dtype == GE_F32) return bn_bwd_launch<float>(dy, y, x, gamma, save_mean, save_rstd, dx, dgamma, dbeta, workspace, N, C, HW, slope, ge_stream(stream));
  if (dtype == GE_BF16) return bn_bwd_launch<bf16_t>(dy, y, x, gamma, save_mean, save_rstd, dx, dgamma, dbeta, workspace, N, C, HW, slope, ge_stream(stream));
  return GE_ERR_UNSUPPORTED;
}
