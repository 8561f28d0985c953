// Multi-scale deformable attention sampling core for gfx950 (replaces mmcv's ms_deform_attn CUDA op,
// reference call sites depth/models/necks/hahi.py:279-289,316-325).
//
// Work decomposition (wave64-first): a group of 16 lanes owns one (batch, query, head); each lane owns
// 4 of the head's 64 channels, so every bilinear tap is one 256-byte (fp32) / 128-byte (bf16)
// fully-coalesced read of `value`.  A 256-thread workgroup therefore covers 16 (query, head) pairs =
// two whole queries, whose sampling locations / attention weights are contiguous in memory.
// The op is a gather (no contraction) -> no MFMA; it is bound by L2/MALL gather bandwidth.
// Backward accumulates d_value with fp32 hardware atomics (global_atomic_add_f32) and reduces
// d_loc / d_attw over the 64 channels with 16-lane butterfly shuffles.
#include "common.h"

#define MSDA_MAX_L 8
struct MsdaLevels { int H[MSDA_MAX_L]; int W[MSDA_MAX_L]; int start[MSDA_MAX_L]; };

template <typename T> struct Vec4;
template <> struct Vec4<float> {
  static __device__ __forceinline__ void ld(const float* p, float v[4]) {
    float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void st(float* p, const float v[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Vec4<bf16_t> {
  static __device__ __forceinline__ void ld(const bf16_t* p, float v[4]) {
    uint2 t = *(const uint2*)p;
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

__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 1, 64);
  return v;
}

template <typename T>
__global__ void __launch_bounds__(256) msda_fwd_k(const T* __restrict__ value, MsdaLevels lv, const float* __restrict__ loc,
                                                  const float* __restrict__ attw, T* __restrict__ out,
                                                  long n_groups, int Nv, int Nq, int nH, int L, int P) {
  const int c4 = (threadIdx.x & 15) * 4;
  const long grp0 = (long)blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
  const long gstride = (long)gridDim.x * (blockDim.x >> 4);
  for (long grp = grp0; grp < n_groups; grp += gstride) {   // grp = (b*Nq + q)*nH + head
    const int head = (int)(grp % nH);
    const long bq = grp / nH;
    const int b = (int)(bq / Nq);
    const float* lp = loc + grp * (long)(L * P * 2);
    const float* ap = attw + grp * (long)(L * P);
    const T* vb = value + ((long)b * Nv * nH + head) * 64 + c4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < L; ++l) {
      const int Hl = lv.H[l], Wl = lv.W[l];
      const T* vl = vb + (long)lv.start[l] * nH * 64;
      for (int p = 0; p < P; ++p) {
        const float lx = lp[(l * P + p) * 2], ly = lp[(l * P + p) * 2 + 1];
        const float wgt = ap[l * P + p];
        const float x = lx * (float)Wl - 0.5f, y = ly * (float)Hl - 0.5f;   // grid_sample, align_corners=False
        if (!(y > -1.f && x > -1.f && y < (float)Hl && x < (float)Wl)) continue;
        const float xf = floorf(x), yf = floorf(y);
        const int x0 = (int)xf, y0 = (int)yf, x1 = x0 + 1, y1 = y0 + 1;
        const float ax = x - xf, ay = y - yf;
        const float w00 = (1.f - ay) * (1.f - ax) * wgt, w01 = (1.f - ay) * ax * wgt;
        const float w10 = ay * (1.f - ax) * wgt, w11 = ay * ax * wgt;
        float v[4];
        if (y0 >= 0 && x0 >= 0) { Vec4<T>::ld(vl + ((long)y0 * Wl + x0) * nH * 64, v);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] += w00 * v[i]; }
        if (y0 >= 0 && x1 < Wl) { Vec4<T>::ld(vl + ((long)y0 * Wl + x1) * nH * 64, v);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] += w01 * v[i]; }
        if (y1 < Hl && x0 >= 0) { Vec4<T>::ld(vl + ((long)y1 * Wl + x0) * nH * 64, v);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] += w10 * v[i]; }
        if (y1 < Hl && x1 < Wl) { Vec4<T>::ld(vl + ((long)y1 * Wl + x1) * nH * 64, v);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] += w11 * v[i]; }
      }
    }
    Vec4<T>::st(out + grp * 64 + c4, acc);
  }
}

template <typename T>
__global__ void __launch_bounds__(256) msda_bwd_k(const T* __restrict__ value, MsdaLevels lv, const float* __restrict__ loc,
                                                  const float* __restrict__ attw, const T* __restrict__ gout,
                                                  float* __restrict__ d_value, float* __restrict__ d_loc,
                                                  float* __restrict__ d_attw, long n_groups, int Nv, int Nq, int nH,
                                                  int L, int P) {
  const int sub = threadIdx.x & 15;
  const int c4 = sub * 4;
  const long grp0 = (long)blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
  const long gstride = (long)gridDim.x * (blockDim.x >> 4);
  // all 64 lanes of a wave must stay converged for the shuffles: iterate a wave-uniform trip count
  const long iters = (n_groups + gstride - 1) / gstride;
  for (long it = 0; it < iters; ++it) {
    const long grp = grp0 + it * gstride;
    const bool live = grp < n_groups;
    const long g_ = live ? grp : 0;
    const int head = (int)(g_ % nH);
    const long bq = g_ / nH;
    const int b = (int)(bq / Nq);
    const float* lp = loc + g_ * (long)(L * P * 2);
    const float* ap = attw + g_ * (long)(L * P);
    const long vbase = ((long)b * Nv * nH + head) * 64 + c4;
    float go[4];
    Vec4<T>::ld(gout + g_ * 64 + c4, go);
    for (int l = 0; l < L; ++l) {
      const int Hl = lv.H[l], Wl = lv.W[l];
      const long lbase = vbase + (long)lv.start[l] * nH * 64;
      for (int p = 0; p < P; ++p) {
        const float lx = lp[(l * P + p) * 2], ly = lp[(l * P + p) * 2 + 1];
        const float wgt = ap[l * P + p];
        const float x = lx * (float)Wl - 0.5f, y = ly * (float)Hl - 0.5f;
        const bool inside = live && (y > -1.f && x > -1.f && y < (float)Hl && x < (float)Wl);
        float s_val = 0.f, s_dx = 0.f, s_dy = 0.f;   // per-lane partial sums over its 4 channels
        if (inside) {
          const float xf = floorf(x), yf = floorf(y);
          const int x0 = (int)xf, y0 = (int)yf, x1 = x0 + 1, y1 = y0 + 1;
          const float ax = x - xf, ay = y - yf, bx = 1.f - ax, by = 1.f - ay;
          float v[4];
          if (y0 >= 0 && x0 >= 0) {
            const long o = lbase + ((long)y0 * Wl + x0) * nH * 64;
            Vec4<T>::ld(value + o, v);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float gv = go[i] * v[i];
              s_val += by * bx * gv; s_dx -= by * gv; s_dy -= bx * gv;
              atomicAdd(d_value + o + i, go[i] * wgt * by * bx);
            }
          }
          if (y0 >= 0 && x1 < Wl) {
            const long o = lbase + ((long)y0 * Wl + x1) * nH * 64;
            Vec4<T>::ld(value + o, v);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float gv = go[i] * v[i];
              s_val += by * ax * gv; s_dx += by * gv; s_dy -= ax * gv;
              atomicAdd(d_value + o + i, go[i] * wgt * by * ax);
            }
          }
          if (y1 < Hl && x0 >= 0) {
            const long o = lbase + ((long)y1 * Wl + x0) * nH * 64;
            Vec4<T>::ld(value + o, v);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float gv = go[i] * v[i];
              s_val += ay * bx * gv; s_dx -= ay * gv; s_dy += bx * gv;
              atomicAdd(d_value + o + i, go[i] * wgt * ay * bx);
            }
          }
          if (y1 < Hl && x1 < Wl) {
            const long o = lbase + ((long)y1 * Wl + x1) * nH * 64;
            Vec4<T>::ld(value + o, v);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float gv = go[i] * v[i];
              s_val += ay * ax * gv; s_dx += ay * gv; s_dy += ax * gv;
              atomicAdd(d_value + o + i, go[i] * wgt * ay * ax);
            }
          }
        }
        s_val = group16_sum(s_val);
        s_dx = group16_sum(s_dx);
        s_dy = group16_sum(s_dy);
        if (live && sub == 0) {
          d_attw[g_ * (long)(L * P) + l * P + p] = s_val;
          d_loc[(g_ * (long)(L * P) + l * P + p) * 2] = s_dx * wgt * (float)Wl;
          d_loc[(g_ * (long)(L * P) + l * P + p) * 2 + 1] = s_dy * wgt * (float)Hl;
        }
      }
    }
  }
}

static int msda_levels(const int* spatial_hw, int L, int Nv, MsdaLevels& lv) {
  if (L < 1 || L > MSDA_MAX_L) return GE_ERR_UNSUPPORTED;
  long start = 0;
  for (int l = 0; l < L; ++l) {
    lv.H[l] = spatial_hw[2 * l]; lv.W[l] = spatial_hw[2 * l + 1]; lv.start[l] = (int)start;
    if (lv.H[l] <= 0 || lv.W[l] <= 0) return GE_ERR_BAD_ARG;
    start += (long)lv.H[l] * lv.W[l];
  }
  return start == Nv ? GE_OK : GE_ERR_BAD_ARG;
}

extern "C" int ge_msda_fwd(const void* value, const int* spatial_hw, const float* loc, const float* attw, void* out,
                           int B, int Nv, int Nq, int nH, int L, int P, int dtype, void* stream) {
  if (!value || !spatial_hw || !loc || !attw || !out || B < 0 || Nv <= 0 || Nq < 0 || nH <= 0 || P <= 0) return GE_ERR_BAD_ARG;
  MsdaLevels lv;
  int e = msda_levels(spatial_hw, L, Nv, lv);
  if (e) return e;
  const long n_groups = (long)B * Nq * nH;
  if (n_groups == 0) return GE_OK;
  const unsigned blocks = ge_blocks(n_groups, 16, 1 << 22);
  if (dtype == GE_F32)
    msda_fwd_k<float><<<blocks, 256, 0, ge_stream(stream)>>>((const float*)value, lv, loc, attw, (float*)out, n_groups, Nv, Nq, nH, L, P);
  else if (dtype == GE_BF16)
    msda_fwd_k<bf16_t><<<blocks, 256, 0, ge_stream(stream)>>>((const bf16_t*)value, lv, loc, attw, (bf16_t*)out, n_groups, Nv, Nq, nH, L, P);
  else
    return GE_ERR_UNSUPPORTED;
  GE_LAUNCH_CHECK();
  return GE_OK;
}

extern "C" int ge_msda_bwd(const void* value, const int* spatial_hw, const float* loc, const float* attw, const void* d_out,
                           float* d_value, float* d_loc, float* d_attw, int B, int Nv, int Nq, int nH, int L, int P,
                           int dtype, void* stream) {
  if (!value || !spatial_hw || !loc || !attw || !d_out || !d_value || !d_loc || !d_attw) return GE_ERR_BAD_ARG;
  if (B < 0 || Nv <= 0 || Nq < 0 || nH <= 0 || P <= 0) return GE_ERR_BAD_ARG;
  MsdaLevels lv;
  int e = msda_levels(spatial_hw, L, Nv, lv);
  if (e) return e;
  const long n_groups = (long)B * Nq * nH;
  if (n_groups == 0) return GE_OK;
  const unsigned blocks = ge_blocks(n_groups, 16, 1 << 22);
  if (dtype == GE_F32)
    msda_bwd_k<float><<<blocks, 256, 0, ge_stream(stream)>>>((const float*)value, lv, loc, attw, (const float*)d_out, d_value, d_loc, d_attw, n_groups, Nv, Nq, nH, L, P);
  else if (dtype == GE_BF16)
    msda_bwd_k<bf16_t><<<blocks, 256, 0, ge_stream(stream)>>>((const bf16_t*)value, lv, loc, attw, (const bf16_t*)d_out, d_value, d_loc, d_attw, n_groups, Nv, Nq, nH, L, P);
  else
    return GE_ERR_UNSUPPORTED;
  GE_LAUNCH_CHECK();
  return GE_OK;
}
