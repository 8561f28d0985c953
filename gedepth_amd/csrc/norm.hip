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
                                                       int use_lds) {
  constexpr int RPB = 256 / GS;
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
                         float* dgamma, float* dbeta, long rows, int C, hipStream_t s) {
  LnPlan pl;
  if (!ln_plan(C, pl)) return GE_ERR_UNSUPPORTED;
  const int rpb = 256 / pl.gs;
  const unsigned blocks = ge_blocks(rows, rpb * 8, 256 * 4);          // few, long-lived workgroups: fewer column atomics
  size_t smem = (size_t)rpb * 2 * C * sizeof(float);
  const int use_lds = smem <= 60 * 1024;                               // else: per-lane column atomics (only LN(3072), few rows)
  if (!use_lds) smem = 0;
#define LN_B(GS_, N_) layernorm_bwd_k<TX, TY, GS_, N_><<<blocks, 256, smem, s>>>((const TY*)dy, (const TX*)x, gamma, mean, rstd, (TX*)dx, dgamma, dbeta, rows, C, use_lds)
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
